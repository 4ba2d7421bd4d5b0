"""profiles/r06_toom_pmc.txt from the output of tools/toom_pmc.sh (+ tools/toom_layer_bench.py): python tools/toom_pmc_table.py
gpurun_out/toom_pmc.txt gpurun_out/toom_layer_bench.txt > profiles/r06_toom_pmc.txt"""
import re, sys
raw = open(sys.argv[1]).read().splitlines()
bench = [l for l in open(sys.argv[2]).read().splitlines() if l.startswith("K=")]
C = {}
for l in raw:
    m = re.match(r"(tdnn_gemm_\w+<\d>)\(\w+\)\s+(\w+)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+(\d+)", l)
    if m:
        C.setdefault(m.group(1), {})[m.group(2)] = (float(m.group(4)), int(m.group(6)))
print("Toom-Cook F(2, K) fp32 kernel (csrc/xv_toom.hip) next to the direct DMA-fed fp32 kernel on the K = 5 / K = 7 layer shapes of the default")
print("topology and the K = 3 shapes of the dilated class (d = 2, 3 and 1; the <3> rows average the three) (512 -> 512, 262144 rows);")
print("tools/toom_pmc.sh (three separate --pmc passes) + tools/toom_layer_bench.py, one box, round 6, final kernel sources")
print("(table: tools/toom_pmc_table.py).\n")
print("layer bench without the profiler (direct ms / TF / of 157.3 | toom ms, algorithmic and EXECUTED TF, error of both against a float64 matmul):")
for l in bench:
    print("  " + l)
print("\n%-28s %9s %9s %10s %12s %12s %10s %12s %10s" % ("kernel", "avg us", "clock GHz", "MFMA busy", "MFMA insts", "other VALU", "VALU/MFMA", "LDS insts", "LDS busy"))
for k in ("tdnn_gemm_dma_kernel<3>", "tdnn_gemm_toom_kernel<3>", "tdnn_gemm_dma_kernel<5>", "tdnn_gemm_toom_kernel<5>", "tdnn_gemm_dma_kernel<7>", "tdnn_gemm_toom_kernel<7>"):
    if k not in C:
        continue
    c = C[k]
    gui, dur = c["GRBM_GUI_ACTIVE"]
    mfma = c["SQ_INSTS_MFMA"][0]; valu = c["SQ_INSTS_VALU"][0] - mfma
    busy = c["SQ_VALU_MFMA_BUSY_CYCLES"][0] / (1024 * gui / 8)
    lds_busy = c["SQ_LDS_IDX_ACTIVE"][0] / (256 * gui / 8)                 # per CU
    print("%-28s %9.1f %9.2f %10.3f %12d %12d %10.2f %12d %10.3f" % (k, dur / 1e3, gui / 8 / dur, busy, mfma, valu, valu / mfma, c["SQ_INSTS_LDS"][0], lds_busy))
print("""
SQ_LDS_BANK_CONFLICT = %s for all of them.  MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs); 'other VALU' = SQ_INSTS_VALU - SQ_INSTS_MFMA
(whole kernel: main loop + prologue + epilogue).  The Toom-Cook kernel executes 0.667 / 0.60 / 0.571 of the direct kernel's MFMAs (4 of 6, 6 of 10, 8 of 14 products per row pair)
and ~1.0 / 1.4 / 1.7 other VALU instructions per MFMA where the direct kernel has ~0.25; beside v_mfma_f32_32x32x2_f32 a VALU instruction is not hidden
(tools/experiments/f32_mfma_valu_probe.hip: ~4 cycles each, +~10 when alone between two MFMAs), which is what separates ~0.85 busy from ~0.95.
""" % ("0" if all(C[k]["SQ_LDS_BANK_CONFLICT"][0] == 0 for k in C) else "NOT 0"))
print("raw counter lines:")
print("\n".join(raw))
