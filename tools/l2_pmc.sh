# L2 (TCC) hit / miss counters of the GEMM kernels of a step -- the counter behind the 1.3 x read amplification the fabric-side
# FETCH_SIZE shows for layers 1 / 2: how many of the L2's requests miss, and how many read requests go out to
# the fabric (Infinity Cache / HBM).  Run on the GPU box from the repo root: bash tools/l2_pmc.sh > gpurun_out/l2_pmc.txt
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pf_l2
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --kernel-trace -d gpurun_out/pf_l2 -- python bench.py --steps 1 --warmup 0 --cpu-budget 0 --e2e-utts 0 --no-fp32-leg --no-extra-legs > /dev/null 2> gpurun_out/l2_pmc.err
python tools/prof_summary.py pmc $(find gpurun_out/pf_l2 -name "*.db" | head -1) > gpurun_out/l2_pmc_raw.txt
rm -rf gpurun_out/pf_l2
python - <<'PY'
import re, collections
rows = collections.defaultdict(dict)
for l in open('gpurun_out/l2_pmc_raw.txt'):
    m = re.match(r'(\S.*?\))\s+(\S+)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+(\d+)\s*$', l)
    if m: rows[m.group(1)][m.group(2)] = (int(m.group(3)), float(m.group(4)), float(m.group(6)))
print("%-60s %8s %14s %14s %10s %16s" % ("kernel", "launches", "L2 hits", "L2 misses", "hit rate", "fabric read reqs"))
for k, c in rows.items():
    if 'TCC_HIT_sum' not in c: continue
    h, ms = c['TCC_HIT_sum'][1], c['TCC_MISS_sum'][1]
    print("%-60s %8d %14.0f %14.0f %9.1f%% %16.0f" % (k[:60], c['TCC_HIT_sum'][0], h, ms, 100 * h / max(h + ms, 1), c.get('TCC_EA0_RDREQ_sum', (0, 0, 0))[1]))
PY
