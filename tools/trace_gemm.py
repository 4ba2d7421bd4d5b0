"""Per-workgroup timeline of the bf16x3 GEMM (needs build/ablate/lib_trace.so from tools/ablate_gemm.py)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["XVECTOR_HIP_LIB"] = os.path.join(ROOT, "build", "ablate", "lib_%s.so" % (sys.argv[1] if len(sys.argv) > 1 else "trace"))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "x-vector-kaldi-tf_amd"))
import numpy as np, torch
from xvector_amd import hiplib
dev = torch.device("cuda:0"); R = 130889
for (cin, cout, K) in ((512, 512, 1), (512, 1536, 1), (512, 512, 7)):
    w = torch.randn((K, cin, cout), device=dev) / (K * cin) ** 0.5
    wp = hiplib.pack_weights_bf16x3(w)
    x = torch.randn((R, cin), device=dev); xs = hiplib.SplitBuf(R, cin, dev); hiplib.split_encode(x, xs)
    bias = torch.zeros(cout, device=dev); rv = torch.ones(R, dtype=torch.uint8, device=dev)
    ys = hiplib.SplitBuf(R, cout, dev)
    trc = torch.zeros((R, cout), device=dev)
    for _ in range(3):
        hiplib.tdnn_layer3(xs, R, wp, bias, None, None, 1, None, 1, rv, ys, trc)
    torch.cuda.synchronize()
    nwg = ((R + 127) // 128) * ((cout + 127) // 128)
    t = trc.view(-1)[: nwg * 16].cpu().numpy().view(np.int64).reshape(nwg, 8)
    t0 = t[:, 0].min()
    us = (t[:, :4] - t0) / 100.0
    P, M, E = us[:, 1] - us[:, 0], us[:, 2] - us[:, 1], us[:, 3] - us[:, 2]
    print("cin %d cout %d K %d: %d WGs, kernel span %.1f us" % (cin, cout, K, nwg, us[:, 3].max()))
    for name, v in (("prologue", P), ("mainloop", M), ("epilogue", E), ("total", us[:, 3] - us[:, 0])):
        print("   %-9s mean %.2f  p10 %.2f  p50 %.2f  p90 %.2f us" % (name, v.mean(), *np.percentile(v, [10, 50, 90])))
    # CU identity: XCC_ID (reg 20) + HW_ID se/sh/cu
    hw, xcc = t[:, 4] & 0xffffff, t[:, 6] & 0xf
    cu = (xcc << 16) | (hw & 0xff00) | ((hw >> 13) & 7) << 20
    cus = np.unique(cu)
    e_lds = (t[:, 7] - t0) / 100.0 - us[:, 2]
    tl = ((t[:, 6] >> 8) - t0) / 100.0
    e_par = 0 * e_lds
    e_loop = tl - (t[:, 7] - t0) / 100.0
    e_drain = us[:, 3] - tl
    print("   epilogue split: acc->LDS+barrier %.2f | param loads %.2f | row loop %.2f | store drain %.2f us (means)" %
          (e_lds.mean(), e_par.mean(), e_loop.mean(), e_drain.mean()))
    cyc = ((t[:, 4] >> 24) & 0xffffffffff) - ((t[:, 5] >> 24) & 0xffffffffff)
    ok = cyc > 0
    print("   raw:", [hex(int(v)) for v in t[5]], int(ok.sum()))
    print("   shader clock while the WG ran: %.3f GHz (median of s_memtime delta / wall delta)" %
          np.median(cyc[ok] / ((us[ok, 3] - us[ok, 0]) * 1e3)))
    print("   distinct CUs seen: %d" % len(cus))
    # timeline of one CU
    sel = np.where(cu == cus[len(cus) // 2])[0]
    sel = sel[np.argsort(us[sel, 0])]
    for i in sel[:10]:
        print("     wg %5d  start %7.2f  P %5.2f  M %6.2f  E %5.2f  end %7.2f" % (t[i, 5] & 0xffffff, us[i, 0], P[i], M[i], E[i], us[i, 3]))
    # busy fraction: union of mainloop intervals vs span, per CU
    gaps = []
    for c in cus[:64]:
        s = np.where(cu == c)[0]
        iv = sorted((us[i, 1], us[i, 2]) for i in s)
        cov, end = 0.0, -1.0
        for a, b in iv:
            if b > end:
                cov += b - max(a, end); end = b
        gaps.append(cov / (max(us[s, 3]) - min(us[s, 0])))
    print("   fraction of CU time with >=1 WG in its main loop: mean %.3f" % np.mean(gaps))
