"""Per-layer timing of the exact-fp32 GEMM (xv_tdnn_layer_f32) on random data: every layer shape of the default topology against
the 157.3 TF fp32-MFMA peak.  argv[1] = rows per batch (default 262144), argv[2] = one layer only (0..4; for counter runs).
XV_BENCH_PERSIST=0 / 1: one workgroup per tile / the persistent launch form (XV_TUNE_FP32_PERSIST; bit-identical: see `bits`)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "x-vector-kaldi-tf_amd"))
import torch
from xvector_amd import hiplib
dev = torch.device("cuda:0"); R = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
if os.environ.get("XV_BENCH_FORM"):                  # XV_TUNE_FP32_GEMM: 1 register-staged, 2 DMA-fed, 3 DMA-fed + K = 1 layers on 16-channel slabs
    hiplib.set_tuning(hiplib.TUNE_FP32_GEMM, int(os.environ["XV_BENCH_FORM"]))
if os.environ.get("XV_BENCH_PERSIST") and hasattr(hiplib, "TUNE_FP32_PERSIST"):     # (only with tools/experiments/fp32_persistent.patch applied)
    hiplib.set_tuning(hiplib.TUNE_FP32_PERSIST, int(os.environ["XV_BENCH_PERSIST"]))
tot_ms = tot_fl = 0.0
SHAPES = ((24, 512, 5), (512, 512, 5), (512, 512, 7), (512, 512, 1), (512, 1536, 1))
for (cin, cout, K) in (SHAPES if len(sys.argv) < 3 else SHAPES[int(sys.argv[2]):int(sys.argv[2]) + 1]):
    w = torch.randn((K * cin, cout), device=dev) / (K * cin) ** 0.5
    wp = hiplib.pack_weights(w)
    torch.manual_seed(cin * 7 + cout + K)
    w = torch.randn((K * cin, cout), device=dev) / (K * cin) ** 0.5
    wp = hiplib.pack_weights(w)
    x = torch.relu(torch.randn((R, cin), device=dev))
    bias = torch.zeros(cout, device=dev); rv = torch.ones(R, dtype=torch.uint8, device=dev)
    scale = torch.ones(cout, device=dev); shift = torch.zeros(cout, device=dev)
    y = torch.empty((R, cout), device=dev)
    fn = lambda: hiplib.tdnn_layer(x, wp, bias, scale, shift, 1, None, K, 1, rv, y)
    ts = []
    for rnd in range(6):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(4): fn()
        b.record(); torch.cuda.synchronize()
        if rnd: ts.append(a.elapsed_time(b) / 4)
    ts.sort(); ms = ts[len(ts) // 2]
    fl = 2.0 * R * cin * cout * K
    tot_ms += ms; tot_fl += fl
    bits = int(y.view(torch.int32).to(torch.int64).sum().item())      # (a checksum of the bit patterns: XV_FP32_DMA=0 / 1 must agree)
    print("%4d -> %4d K=%d: %.3f ms  %.1f TF = %.3f of 157.3   (output %.2f GB at %.2f TB/s)  bits %d" % (cin, cout, K, ms, fl / ms / 1e9, fl / ms / 1e9 / 157.3, R * cout * 4 / 1e9, R * cout * 4 / ms / 1e9, bits))
print("all five: %.3f ms, %.1f TF = %.3f" % (tot_ms, tot_fl / tot_ms / 1e9, tot_fl / tot_ms / 1e9 / 157.3))
