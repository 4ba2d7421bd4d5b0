"""CPU check of the Toom-Cook tables compiled into csrc/xv_toom.hip against the exact-rational construction
(tools/experiments/toomcook_gen.py): G, the scales SC, AT's second row A1, the stage ORDER and the row masks NEED are parsed from
the source and must reproduce the K-tap correlation exactly, with the scaled transform rows BT / SC the kernel's xform() writes
out.  (The hand-written transforms themselves are covered on the GPU, tests/test_gpu_toom.py.)"""
import os
import re
import sys
from fractions import Fraction as Fr

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools", "experiments"))


def _tables(K):
    src = open(os.path.join(ROOT, "x-vector-kaldi-tf_amd", "csrc", "xv_toom.hip")).read()
    body = src[src.index("struct Toom<%d>" % K):]
    body = body[:body.index("\n};") + 3]
    J = K + 1

    def arr(name):
        m = re.search(r"%s(?:\[\d+\])+ = (\{.*?\});" % name, body, re.S)
        text = re.sub(r"(\d+)\.(?=[^\d])", r"\1", m.group(1))            # "1. / 6" -> "1 / 6"
        text = text.replace("f", "").replace("{", "[").replace("}", "]")
        return eval(re.sub(r"(-?\d+(?:\.\d+)?)", r"Fr('\1')", text))
    G, SC, A1, ORDER = arr("G"), arr("SC"), arr("A1"), arr("ORDER")
    bt_comment = re.search(r"// BT = (\{\{.*?\}\})", body, re.S).group(1)
    bt_comment = re.sub(r"\n\s*//", " ", bt_comment)
    BT = eval(re.sub(r"(-?\d+)", r"Fr(\1)", bt_comment.replace("{", "[").replace("}", "]")))
    assert len(G) == J and len(BT) == J and all(len(r) == K for r in G) and all(len(r) == J for r in BT)
    need = [int(n, 16) for n in re.search(r"NEED\[\d+\] = \{(.*?)\}", body).group(1).replace(" ", "").split(",")]
    return G, SC, A1, [int(o) for o in ORDER], need, BT


def test_tables_in_the_kernel_source_are_the_exact_construction():
    import random
    from toomcook_gen import toomcook, P5, P7
    random.seed(5)
    for K, pts in ((5, P5), (7, P7)):
        G, SC, A1, ORDER, NEED, BT = _tables(K)
        AT, Gref, BTref = toomcook(2, K, pts)
        J = K + 1
        assert BT == BTref and G == Gref                                  # the generator's integer-row form, row for row
        assert A1 == AT[1] and AT[0] == [Fr(1)] * (J - 1) + [Fr(0)]       # row 2P: every product but infinity's, coefficient 1
        assert sorted(ORDER) == list(range(J)) and ORDER[0] == 0 and ORDER[1] == J - 1      # the two one-row products first
        assert all(ORDER[q + 1] == ORDER[q] + 1 for q in range(2, J - 1, 2))                # +- pairs adjacent, first of a pair on an even stage
        for j in range(J):
            assert NEED[j] == sum(1 << i for i in range(J) if BT[j][i] != 0), j
            assert SC[j] != 0
        for _ in range(20):                                               # out = AT [ (SC G g) . (BT d / SC) ]  ==  the correlation
            d = [Fr(random.randint(-9, 9)) for _ in range(J)]
            g = [Fr(random.randint(-9, 9)) for _ in range(K)]
            U = [SC[j] * sum(G[j][k] * g[k] for k in range(K)) for j in range(J)]
            V = [sum(BT[j][i] * d[i] for i in range(J)) / SC[j] for j in range(J)]
            out = [sum(AT[q][j] * U[j] * V[j] for j in range(J)) for q in range(2)]
            assert out == [sum(d[q + k] * g[k] for k in range(K)) for q in range(2)]
        # the scaled rows the kernel's xform() writes out: a unit coefficient to start every chain from
        for j in range(J):
            row = [v / SC[j] for v in BT[j]]
            assert any(abs(v) == 1 for v in row if v != 0) or K == 5 and j in (1, 2), (K, j, row)
