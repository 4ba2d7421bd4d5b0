# The rocprofv3 passes behind profiles/<tag>_* and profiles/traffic.json.  Run on the GPU box:  bash tools/profile_round.sh
# (kernel trace + stats of the default bench command, the plain bench line, and three separate PMC passes -- FETCH_SIZE,
# WRITE_SIZE, SQ/GRBM -- at the default batch size; PMC and tracing domains are never combined).
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/prof_sq
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_stats -- python bench.py --cpu-budget 0 --e2e-utts 0 --no-fp32-leg --no-extra-legs > gpurun_out/bench_under_prof.json 2> gpurun_out/prof_stats.err
python bench.py > gpurun_out/bench_plain.json 2> gpurun_out/bench_plain.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/prof_fetch -- python bench.py --steps 1 --warmup 0 --cpu-budget 0 --e2e-utts 0 --no-fp32-leg --no-extra-legs > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/prof_write -- python bench.py --steps 1 --warmup 0 --cpu-budget 0 --e2e-utts 0 --no-fp32-leg --no-extra-legs > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d gpurun_out/prof_sq --kernel-trace -- python bench.py --steps 1 --warmup 0 --cpu-budget 0 --e2e-utts 0 --no-fp32-leg --no-extra-legs > /dev/null 2>&1
python tools/prof_summary.py stats $(find gpurun_out/prof_stats -name "*.db" | head -1) > gpurun_out/round_stats.txt
: > gpurun_out/round_pmc.txt
for d in prof_fetch prof_write prof_sq; do python tools/prof_summary.py pmc $(find gpurun_out/$d -name "*.db" | head -1) >> gpurun_out/round_pmc.txt; done
rm -rf gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/prof_sq
