# Counters of the Toom-Cook fp32 kernel next to the direct DMA-fed kernel on the K = 5 / K = 7 layer shapes.
# Run on the GPU box from the repo root: bash tools/toom_pmc.sh > gpurun_out/toom_pmc.txt
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for set in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_LDS" "SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS"; do
  rm -rf gpurun_out/pf_toom
  rocprofv3 --pmc $set -d gpurun_out/pf_toom --kernel-trace -- python tools/toom_layer_bench.py 262144 > /dev/null 2>&1
  python tools/prof_summary.py pmc $(find gpurun_out/pf_toom -name "*.db" | head -1) | grep "kernel  \|tdnn_gemm_toom\|tdnn_gemm_dma"
  rm -rf gpurun_out/pf_toom
done
