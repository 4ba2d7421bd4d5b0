"""Turn the outputs of tools/profile_round.sh (gpurun_out/) into the committed profiles/<tag>_kernel_stats.txt,
profiles/<tag>_pmc.txt and profiles/traffic.json (stamped with the kernel sources and batch size it was measured at:
bench.py quotes it only while both still match).   python tools/write_profiles.py r02c"""
import hashlib, json, os, re, subprocess, sys
tag = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
commit = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"]).decode().strip()
h = hashlib.sha256()
d = os.path.join(ROOT, "x-vector-kaldi-tf_amd", "csrc")
for f in sorted(os.listdir(d)):
    if f.endswith((".hip", ".h")):                       # device code (the host library, xv_host.cpp, moves no HBM traffic)
        h.update(open(os.path.join(d, f), "rb").read())
sha = h.hexdigest()[:16]
stats = open("gpurun_out/round_stats.txt").read()
rd = lambda n: open("gpurun_out/%s.json" % n).read().strip()
GEMM = ("tdnn_gemm_bf16x3_kernel", "tdnn_gemm_f16bf8_wide16_kernel", "tdnn_gemm_f16bf8_wide_kernel", "tdnn_gemm_f16bf8_kernel", "tdnn_pair_pool_kernel", "tdnn_pair_pool_f16bf8_kernel", "tdnn_first_kernel")
tot = cnt = 0
for l in stats.splitlines():
    if l.startswith(GEMM):
        f = l.split(); cnt += int(f[-4]); tot += float(f[-3])
under = json.loads(rd("bench_under_prof"))
hdr = ("# rocprofv3 --kernel-trace --stats -- python bench.py --cpu-budget 0 --e2e-utts 0 --no-fp32-leg --no-extra-legs   (MI355X, default precision %s,\n"
       "#   commit %s, kernel sources %s; tools/profile_round.sh).  6 steps (1 warm-up + 5 timed) x 12 batches of <= 262144 rows + the 6 passes over the largest batch behind roofline.by_launch.\n"
       "#   tdnn_first_kernel<2, true>           = layer 0 (K=5, 23 MFCC dims in 24 columns -> 512; bf16x3 arithmetic, split8 output from the accumulators)\n"
       "#   tdnn_gemm_f16bf8_wide16_kernel<5|7, false> = layers 1 / 2 (K = 5 / 7, 512 -> 512): fp16 MFMA + scaled bf8 MFMA per product on the 16 x 16 shapes, 256 x 256 workgroup tiles\n"
       "#   tdnn_pair_pool_f16bf8_kernel<2>      = layers 3 + 4 (K=1, 512 -> 512 -> 1536) chained in registers (f16bf8, a pair of waves per 32 frames) + 8-row block statistics of the pooling\n"
       "#   tdnn_gemm_bf16x3_kernel<false,0,false,2> = the per-step segment FC (embed_layer-0)\n"
       "# All GEMM kernels together: %d launches, total %.1f us, average %.2f us per launch  (bench.py roofline.avg_launch_ms %.4f under the profiler)\n"
       "# stats_pool_kernel is bench.py's separate \"roofline_pool\" measurement of the standalone pooling kernel, outside the timed region.\n"
       % (under["config"]["precision"], commit, sha, cnt, tot, tot / max(cnt, 1), under["roofline"]["avg_launch_ms"]))
open("profiles/%s_kernel_stats.txt" % tag, "w").write(
    hdr + stats + "\n# bench.py line printed under the profiler:\n" + rd("bench_under_prof") +
    "\n\n# bench.py line of the same build, same box, without the profiler (default flags: cpu_baseline, fp32_exact, e2e_ark_to_ark):\n" + rd("bench_plain") + "\n")
pmc = open("gpurun_out/round_pmc.txt").read().splitlines()
keep = [l for l in pmc if l.startswith("kernel ") or any(g in l for g in GEMM) or "stats_pool" in l]
vals = {}
for l in keep:
    m = re.match(r"(\S+<[^>]*>|\S+)\(.*?\s+(\S+)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+(\d+)\s*$", l)
    if m:
        vals.setdefault(m.group(1), {})[m.group(2)] = (int(m.group(3)), float(m.group(4)), float(m.group(5)), int(m.group(6)))
lines = []
for k, v in vals.items():
    if "GRBM_GUI_ACTIVE" in v and "SQ_VALU_MFMA_BUSY_CYCLES" in v and v["SQ_VALU_MFMA_BUSY_CYCLES"][1] > 0:
        cyc = v["GRBM_GUI_ACTIVE"][1] / 8
        lines.append("#   %-48s %4d launches  avg %8.1f us  MFMA busy %5.1f %% of cycles at %.2f GHz  waves parked (s_waitcnt/barrier) %4.1f %%" %
                     (k, v["GRBM_GUI_ACTIVE"][0], v["GRBM_GUI_ACTIVE"][3] / 1e3, 100 * v["SQ_VALU_MFMA_BUSY_CYCLES"][1] / (cyc * 1024),
                      cyc / v["GRBM_GUI_ACTIVE"][3], 100 * v["SQ_WAIT_ANY"][1] / v["SQ_WAVE_CYCLES"][1] if "SQ_WAIT_ANY" in v else float("nan")))
hdr2 = ("# rocprofv3 --pmc <set> --kernel-trace -- python bench.py --steps 1 --warmup 0 --cpu-budget 0 --e2e-utts 0 --no-fp32-leg --no-extra-legs   (separate passes: FETCH_SIZE | WRITE_SIZE |\n"
        "#   SQ/GRBM set; tools/profile_round.sh, commit %s, kernel sources %s).  12 batches of <= 262144 rows per pass (+ 6 by_launch passes over the largest batch); last column = average kernel duration in ns.\n"
        "# FETCH_SIZE / WRITE_SIZE in KiB per launch.  Fabric-side bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md, verified on\n"
        "#   stats_pool_kernel: its corrected reads equal its algorithmic bytes within 0.3 %%).  FETCH_SIZE counts L2-miss requests: Infinity-Cache hits are included.\n"
        "# MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 XCDs x 1024 SIMDs); effective shader clock = (GRBM_GUI_ACTIVE/8) / duration (nominal 2.4 GHz):\n"
        % (commit, sha) + "\n".join(lines) + "\n")
open("profiles/%s_pmc.txt" % tag, "w").write(hdr2 + "\n".join(keep) + "\n")
fs = sum(v["FETCH_SIZE"][2] for k, v in vals.items() if k.startswith(GEMM) and "FETCH_SIZE" in v)
ws = sum(v["WRITE_SIZE"][2] for k, v in vals.items() if k.startswith(GEMM) and "WRITE_SIZE" in v)
n = sum(v["FETCH_SIZE"][0] for k, v in vals.items() if k.startswith(GEMM) and "FETCH_SIZE" in v)
json.dump({"round": 6, "kernel": "GEMM kernels of a step (tdnn_first_kernel, tdnn_gemm_f16bf8_wide16_kernel, tdnn_pair_pool_f16bf8_kernel, tdnn_gemm_bf16x3_kernel; launch-weighted mean)",
           "source": "profiles/%s_pmc.txt (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, default bench.py workload)" % tag,
           "kernel_sha": sha, "commit": commit, "batch_rows": under["config"]["batch_rows"], "launches": n,
           "gemm_fetch_kib_raw": round(fs / n, 1), "gemm_write_kib": round(ws / n, 1),
           "gemm_hbm_bytes_per_launch": int((2 * fs / n + ws / n) * 1024),
           "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024; x2 on FETCH_SIZE per MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B), verified on stats_pool_kernel",
           "note": "fabric-side traffic incl. Infinity-Cache hits"},
          open("profiles/traffic.json", "w"), indent=1)
print("\n".join(lines)); print(open("profiles/traffic.json").read())
