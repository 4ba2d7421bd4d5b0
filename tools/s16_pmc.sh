# PMC evidence for the 16 x 16 MFMA form of the bf16x3 GEMM: the K = 5 / 7 layers with XV_BF16X3_S16=1 (default) and =0,
# MFMA-busy cycles and the effective clock.  Run on the GPU box from the repo root: bash tools/s16_pmc.sh > gpurun_out/s16_pmc.txt
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for v in 1 0; do
  export XV_BF16X3_S16=$v
  rm -rf gpurun_out/pf_s16
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_IDX_ACTIVE -d gpurun_out/pf_s16 --kernel-trace -- python tools/layer_bench.py > /dev/null 2>&1
  echo "== XV_BF16X3_S16=$v"
  python tools/prof_summary.py pmc $(find gpurun_out/pf_s16 -name "*.db" | head -1) | grep "kernel  \|bf16x3_kernel<true, [57], false" 
  rm -rf gpurun_out/pf_s16
done
