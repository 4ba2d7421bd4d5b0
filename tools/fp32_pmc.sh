# Counters of the exact-fp32 GEMM per layer shape: the K = 7 layer against the two K = 1 layers.
# Run on the GPU box from the repo root: bash tools/fp32_pmc.sh > gpurun_out/fp32_pmc.txt
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for layer in 2 3 4; do
  for set in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_LDS"; do
    rm -rf gpurun_out/pf_f32
    rocprofv3 --pmc $set -d gpurun_out/pf_f32 --kernel-trace -- python tools/fp32_layer_bench.py 262144 $layer > /dev/null 2>&1
    echo "== layer $layer"
    python tools/prof_summary.py pmc $(find gpurun_out/pf_f32 -name "*.db" | head -1) | grep "kernel  \|tdnn_gemm_kernel"
    rm -rf gpurun_out/pf_f32
  done
done
