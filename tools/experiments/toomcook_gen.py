"""Toom-Cook F(2, K) matrices over time for the K-tap correlation of a TDNN layer (local/tf/models.py:60):
y[i] = sum_k d[i + k] g[k], i < m, k < r  ==  A^T [ (G g) . (B^T d) ]   with n = m + r - 1 products instead of m r.
Transposition principle: the linear convolution s = u * g is  C [ (E_m u) . (E_r g) ]  (evaluate at n points, multiply, interpolate,
C = E_n^-1), hence the correlation is E_m^T [ (E_r g) . (C^T d) ].  Exact rationals; rows scaled so that B^T is integer-ish."""
from fractions import Fraction as Fr
import numpy as np

INF = "inf"


def evalmat(points, deg):
    rows = []
    for p in points:
        if p == INF:
            rows.append([Fr(0)] * (deg - 1) + [Fr(1)])
        else:
            rows.append([Fr(p) ** i for i in range(deg)])
    return rows


def inv(M):
    n = len(M)
    A = [list(r) + [Fr(int(i == j)) for j in range(n)] for i, r in enumerate(M)]
    for c in range(n):
        p = next(r for r in range(c, n) if A[r][c] != 0)
        A[c], A[p] = A[p], A[c]
        A[c] = [v / A[c][c] for v in A[c]]
        for r in range(n):
            if r != c and A[r][c] != 0:
                f = A[r][c]
                A[r] = [a - f * b for a, b in zip(A[r], A[c])]
    return [r[n:] for r in A]


def toomcook(m, r, points):
    n = m + r - 1
    assert len(points) == n
    Em, Er, En = evalmat(points, m), evalmat(points, r), evalmat(points, n)
    C = inv(En)
    AT = [[Em[j][i] for j in range(n)] for i in range(m)]
    G = Er
    BT = [[C[i][j] for i in range(n)] for j in range(n)]
    # scale row j of B^T to integers with gcd 1 (inverse on G)
    from math import lcm, gcd
    for j in range(n):
        den = 1
        for v in BT[j]:
            den = lcm(den, v.denominator)
        num = 0
        for v in BT[j]:
            num = gcd(num, int(v * den))
        s = Fr(den, num)
        if next(v for v in reversed(BT[j]) if v != 0) < 0:
            s = -s
        BT[j] = [v * s for v in BT[j]]
        G[j] = [v / s for v in G[j]]
    return AT, G, BT


def check(m, r, points):
    AT, G, BT = toomcook(m, r, points)
    n = m + r - 1
    import random
    d = [Fr(random.randint(-9, 9)) for _ in range(n)]
    g = [Fr(random.randint(-9, 9)) for _ in range(r)]
    U = [sum(G[j][k] * g[k] for k in range(r)) for j in range(n)]
    V = [sum(BT[j][i] * d[i] for i in range(n)) for j in range(n)]
    y = [sum(AT[i][j] * U[j] * V[j] for j in range(n)) for i in range(m)]
    ref = [sum(d[i + k] * g[k] for k in range(r)) for i in range(m)]
    assert y == ref, (y, ref)
    return AT, G, BT


P5 = [0, 1, -1, 2, -2, INF]
P7 = [0, 1, -1, 2, -2, Fr(1, 2), Fr(-1, 2), INF]
P3 = [0, 1, -1, INF]

if __name__ == "__main__":
    for m, r, pts in ((2, 5, P5), (2, 7, P7), (2, 3, P3)):
        AT, G, BT = check(m, r, pts)
        print("F(%d,%d) points %s" % (m, r, pts))
        for name, M in (("A^T", AT), ("G", G), ("B^T", BT)):
            print(name)
            for row in M:
                print("   ", "  ".join("%7s" % v for v in row))
