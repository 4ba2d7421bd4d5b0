cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/lp8
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d gpurun_out/lp8 -- python ${1:-tools/layer8_bench.py} 262144 > gpurun_out/l8_under_pmc.txt 2>&1
python tools/prof_summary.py pmc $(find gpurun_out/lp8 -name "*.db" | head -1) ${2:-f16bf8} > gpurun_out/l8_pmc.txt
rm -rf gpurun_out/lp8
python - <<'PY'
import re,collections
rows=collections.defaultdict(dict)
for l in open('gpurun_out/l8_pmc.txt'):
    m=re.match(r'(\S.*?\))\s+(\S+)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+(\d+)\s*$', l)
    if m: rows[m.group(1)][m.group(2)]=(float(m.group(4)), float(m.group(6)))
for k,c in rows.items():
    gui=c['GRBM_GUI_ACTIVE'][0]; dur=c['GRBM_GUI_ACTIVE'][1]
    clk=gui/8/dur
    print("%-60s dur %7.1f us clk %.2f GHz MFMA busy %.1f%% wait_any %.1f%% wait_inst %.1f%% lds_active %.1f%% conflicts/lds %.2f" % (k[:60], dur/1e3, clk, 100*c['SQ_VALU_MFMA_BUSY_CYCLES'][0]/(gui/8*1024), 100*c['SQ_WAIT_ANY'][0]/c['SQ_WAVE_CYCLES'][0], 100*c['SQ_WAIT_INST_ANY'][0]/c['SQ_WAVE_CYCLES'][0], 100*c['SQ_LDS_IDX_ACTIVE'][0]/(gui/8*256), c['SQ_LDS_BANK_CONFLICT'][0]/max(c['SQ_LDS_IDX_ACTIVE'][0],1)))
PY
