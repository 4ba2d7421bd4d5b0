"""CPU simulation of candidate split-precision schemes for the TDNN GEMMs against the fp64 forward (EXPERIMENT).

  bf16x3      : x = xh + xl (bf16 each), products xh*wh + xh*wl + xl*wh           (what ships; 3 bf16 MFMAs per product)
  f16+fp8x2   : xh, wh in fp16; cross terms xl*wh + xh*wl with BOTH factors rounded to fp8 e4m3 (scaled by a power of two)
                -> 1 fp16 MFMA + 2 fp8 MFMAs (= 2 bf16-MFMA times per product)
  bf16+fp8x2  : the same with a bf16 main term

Prints the relative L2 error of the embedding for a few utterances.  Products are accumulated in fp64 here, so the figures are
the scheme's representation error alone (the fp32 accumulation adds ~1e-6).
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "x-vector-kaldi-tf_amd"))
from oracle import oracle                                   # noqa: E402
from xvector_amd import synthetic                           # noqa: E402


def rnd(a, dt):
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dt).to(torch.float64).numpy()


def fp8(a, scale):
    t = torch.from_numpy(np.ascontiguousarray(a * scale, np.float32)).clamp(-448, 448)
    return t.to(torch.float8_e4m3fn).to(torch.float64).numpy() / scale


def make_mm(scheme):
    if scheme == "exact":
        return lambda x, w: x @ w
    main = torch.float16 if scheme.startswith("f16") else torch.bfloat16
    lo_scale = 2.0 ** (15 if main == torch.float16 else 12)

    def mm(x, w):
        x32, w32 = x.astype(np.float32).astype(np.float64), w.astype(np.float32).astype(np.float64)
        xh, wh = rnd(x32, main), rnd(w32, main)
        xl, wl = x32 - xh, w32 - wh
        if scheme == "bf16x3":
            xl, wl = rnd(xl, torch.bfloat16), rnd(wl, torch.bfloat16)
            return xh @ wh + xh @ wl + xl @ wh
        sx = 2.0 ** np.round(-np.log2(np.abs(x32).max() + 1e-30) + 7)      # largest |x| -> ~128..256
        sw = 2.0 ** np.round(-np.log2(np.abs(w32).max() + 1e-30) + 7)
        return xh @ wh + fp8(xl, sx * lo_scale) @ fp8(wh, sw) + fp8(xh, sx) @ fp8(wl, sw * lo_scale)
    return mm


def forward(x, weights, topo, mm, first_exact=True):
    h = np.asarray(x, np.float64)

    def bn(r, scope):
        g, be, m, v = (np.asarray(a, np.float64) for a in oracle._bn(weights, scope))
        s = g / np.sqrt(v + oracle.BN_EPSILON)
        return r * s + (be - m * s)

    for i, (K, d) in enumerate(zip(topo["kernel_sizes"], topo["dilations"])):
        sc = "frame_level_info_layer-%d" % i
        w = np.asarray(weights[sc + "/w:0"], np.float64)
        T = h.shape[0]
        left = (K - 1) * d // 2
        hp = np.zeros((T + (K - 1) * d, h.shape[1]))
        hp[left:left + T] = h
        cols = np.concatenate([hp[k * d:k * d + T] for k in range(K)], axis=1)
        f = make_mm("bf16x3") if (i == 0 and first_exact) else mm
        z = f(cols, w.reshape(-1, w.shape[2])) + np.asarray(weights[sc + "/b:0"], np.float64)
        h = bn(np.maximum(z, 0.0), sc).astype(np.float32).astype(np.float64)
    mu = h.mean(axis=0)
    var = ((h - mu) ** 2).mean(axis=0)
    pooled = np.concatenate([mu, np.sqrt(var + oracle.VAR2STD_EPSILON)])
    return pooled @ np.asarray(weights["embed_layer-0/w:0"], np.float64) + weights["embed_layer-0/b:0"]


def main():
    topo = dict(oracle.DEFAULT_TOPOLOGY)
    for name, weights in (("reference_init", synthetic.reference_init(topo, 23, 64, seed=1)),
                          ("trained_like", synthetic.trained_like(topo, 23, seed=1))):
        rng = np.random.default_rng(0)
        xs = [rng.standard_normal((T, 23)).astype(np.float32) * 3 for T in (200, 317, 400)]
        ref = [forward(x, weights, topo, make_mm("exact"), first_exact=False) for x in xs]
        for scheme in ("bf16x3", "f16+fp8x2", "bf16+fp8x2"):
            err = [oracle.rel_l2(forward(x, weights, topo, make_mm(scheme)), r) for x, r in zip(xs, ref)]
            print("%-15s %-12s rel L2 %s" % (name, scheme, " ".join("%.2e" % e for e in err)))


if __name__ == "__main__":
    main()
