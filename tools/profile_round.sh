cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/prof_sq
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_stats -- python bench.py --cpu-budget 0 > gpurun_out/bench_under_prof.json 2> gpurun_out/prof_stats.err
python bench.py > gpurun_out/bench_plain.json 2> gpurun_out/bench_plain.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/prof_fetch -- python bench.py --steps 1 --warmup 0 --cpu-budget 0 --utts 2000 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/prof_write -- python bench.py --steps 1 --warmup 0 --cpu-budget 0 --utts 2000 > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_ANY -d gpurun_out/prof_sq --kernel-trace -- python bench.py --steps 1 --warmup 0 --cpu-budget 0 --utts 2000 > /dev/null 2>&1
for d in prof_stats prof_fetch prof_write prof_sq; do f=$(find gpurun_out/$d -name "*.db" | head -1); echo "== $d $f"; done
python tools/prof_summary.py stats $(find gpurun_out/prof_stats -name "*.db" | head -1) > gpurun_out/r01d_stats.txt
for d in prof_fetch prof_write prof_sq; do python tools/prof_summary.py pmc $(find gpurun_out/$d -name "*.db" | head -1) >> gpurun_out/r01d_pmc.txt; done
python bench.py --precision fp32 --cpu-budget 0 > gpurun_out/bench_fp32.json 2>/dev/null
