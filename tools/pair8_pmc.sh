# PMC of the pair kernel (tools/pair8_bench.py): duration, clock, MFMA-busy, waiting, LDS activity, bank conflicts -- on random data and
# on zeros, for the library in the tree and (optionally) for another build of it:  bash tools/pair8_pmc.sh [other_lib.so] > gpurun_out/x.txt
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for lib in "" "$1"; do
  for zero in "" wx; do
    [ -z "$lib" ] && [ -n "$1" ] || true
    if [ -n "$lib" ]; then export XVECTOR_HIP_LIB=$lib; name="other build ($lib)"; else unset XVECTOR_HIP_LIB; name="this tree"; fi
    [ -z "$lib" ] || [ -f "$lib" ] || continue
    rm -rf gpurun_out/lp8
    PAIR8_BENCH_ZERO=$zero rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d gpurun_out/lp8 -- python tools/pair8_bench.py > /dev/null 2>&1
    python tools/prof_summary.py pmc $(find gpurun_out/lp8 -name "*.db" | head -1) pair_pool_f16bf8 > gpurun_out/p8_pmc_raw.txt
    rm -rf gpurun_out/lp8
    echo "== $name, ${zero:-random data}${zero:+ = all-zero frames and weights}"
    python - <<'PY'
import re,collections
rows=collections.defaultdict(dict)
for l in open('gpurun_out/p8_pmc_raw.txt'):
    m=re.match(r'(\S.*?\))\s+(\S+)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+(\d+)\s*$', l)
    if m: rows[m.group(1)][m.group(2)]=(float(m.group(4)), float(m.group(6)))
for k,c in rows.items():
    gui=c['GRBM_GUI_ACTIVE'][0]; dur=c['GRBM_GUI_ACTIVE'][1]
    clk=gui/8/dur
    print("%-44s dur %7.1f us clk %.2f GHz MFMA busy %.1f%% insts_mfma %.1f M wait_any %.1f%% wait_inst %.1f%% lds_active %.1f%% (%.1f M cycles) conflicts/lds %.3f" % (k[:44], dur/1e3, clk, 100*c['SQ_VALU_MFMA_BUSY_CYCLES'][0]/(gui/8*1024), c['SQ_INSTS_MFMA'][0]/1e6, 100*c['SQ_WAIT_ANY'][0]/c['SQ_WAVE_CYCLES'][0], 100*c['SQ_WAIT_INST_ANY'][0]/c['SQ_WAVE_CYCLES'][0], 100*c['SQ_LDS_IDX_ACTIVE'][0]/(gui/8*256), c['SQ_LDS_IDX_ACTIVE'][0]/1e6, c['SQ_LDS_BANK_CONFLICT'][0]/max(c['SQ_LDS_IDX_ACTIVE'][0],1)))
PY
  done
  [ -n "$1" ] || break
done
