# L2 hit rate / fabric read requests / clock / time of the wide f16bf8 kernels with and without XCD-aware column placement
# (VERDICT r4 item 7).  Run on the GPU box from the repo root: bash tools/xcd_columns_pmc.sh > gpurun_out/xcd_columns.txt
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python tools/xcd_columns_ab.py 2>&1 | grep "K="
for mode in 0 1; do
  for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
    rm -rf gpurun_out/pf_xc
    XV_XCD_COLUMNS=$mode rocprofv3 --pmc $set --kernel-trace -d gpurun_out/pf_xc -- python bench.py --steps 1 --warmup 0 --cpu-budget 0 --e2e-utts 0 --no-fp32-leg --no-extra-legs > /dev/null 2> gpurun_out/xcd_columns.err
    echo "== XV_XCD_COLUMNS=$mode"
    python tools/prof_summary.py pmc $(find gpurun_out/pf_xc -name "*.db" | head -1) | grep "kernel  \|wide16"
    rm -rf gpurun_out/pf_xc
  done
done
for mode in 0 1; do
  echo "== step, XV_XCD_COLUMNS=$mode"
  XV_XCD_COLUMNS=$mode python bench.py --steps 10 --warmup 2 --cpu-budget 0 --e2e-utts 0 --no-fp32-leg --no-extra-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"
done
