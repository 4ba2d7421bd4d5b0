# PMC passes over tools/layer_bench.py (every GEMM shape of the default topology + the pair kernel); run on the GPU box:
#   bash tools/profile_layers.sh <tag>      -> gpurun_out/<tag>_layers_pmc.txt
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
tag=${1:-r02}
rm -rf gpurun_out/lp_sq gpurun_out/lp_fetch gpurun_out/lp_write
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d gpurun_out/lp_sq -- python tools/layer_bench.py 262144 > gpurun_out/${tag}_layers_under_pmc.txt 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/lp_fetch -- python tools/layer_bench.py 262144 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/lp_write -- python tools/layer_bench.py 262144 > /dev/null 2>&1
: > gpurun_out/${tag}_layers_pmc.txt
for d in lp_sq lp_fetch lp_write; do python tools/prof_summary.py pmc $(find gpurun_out/$d -name "*.db" | head -1) >> gpurun_out/${tag}_layers_pmc.txt; done
rm -rf gpurun_out/lp_sq gpurun_out/lp_fetch gpurun_out/lp_write
