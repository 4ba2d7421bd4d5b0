# PMC comparison of the two 256 x 256 f16bf8 kernels (32 x 32 and 16 x 16 MFMA shapes) on the K = 5 / 7 layers: MFMA-busy cycles, clock,
# waiting, LDS activity and bank conflicts.  Run on the GPU box from the repo root: bash tools/wide16_pmc.sh [random|zeros] > gpurun_out/x.txt
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
if [ "${1:-random}" = zeros ]; then export WIDE_BENCH_ZERO=1; fi
rm -rf gpurun_out/lp16
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d gpurun_out/lp16 -- python tools/wide_bench.py > /dev/null 2>&1
python tools/prof_summary.py pmc $(find gpurun_out/lp16 -name "*.db" | head -1) wide > gpurun_out/w16_pmc_raw.txt
rm -rf gpurun_out/lp16
python - <<'PY'
import re,collections
rows=collections.defaultdict(dict)
for l in open('gpurun_out/w16_pmc_raw.txt'):
    m=re.match(r'(\S.*?\))\s+(\S+)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+(\d+)\s*$', l)
    if m: rows[m.group(1)][m.group(2)]=(float(m.group(4)), float(m.group(6)))
for k,c in rows.items():
    gui=c['GRBM_GUI_ACTIVE'][0]; dur=c['GRBM_GUI_ACTIVE'][1]
    clk=gui/8/dur
    print("%-52s dur %7.1f us clk %.2f GHz MFMA busy %.1f%% insts_mfma %.1f M wait_any %.1f%% wait_inst %.1f%% lds_active %.1f%% conflicts/lds %.3f" % (k[:52], dur/1e3, clk, 100*c['SQ_VALU_MFMA_BUSY_CYCLES'][0]/(gui/8*1024), c['SQ_INSTS_MFMA'][0]/1e6, 100*c['SQ_WAIT_ANY'][0]/c['SQ_WAVE_CYCLES'][0], 100*c['SQ_WAIT_INST_ANY'][0]/c['SQ_WAVE_CYCLES'][0], 100*c['SQ_LDS_IDX_ACTIVE'][0]/(gui/8*256), c['SQ_LDS_BANK_CONFLICT'][0]/max(c['SQ_LDS_IDX_ACTIVE'][0],1)))
PY
