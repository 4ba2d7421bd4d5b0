# The pair kernel's parity tests + tools/pair8_bench.py (random data and zeros), twice; with OLD=<another libxvector_hip.so> (e.g. one
# built from an earlier xv_pair8.hip and this tree's other objects) the same bench on that library in alternation: A/B in one call.
mkdir -p gpurun_out/p8a
(python -m pytest tests/test_gpu_f16bf8.py -x -q -k "pair or block_stat or rows_past" 2>&1 | tail -5) > gpurun_out/p8a/tests.log; cat gpurun_out/p8a/tests.log
for i in 1 2; do
  echo "== this tree"; python tools/pair8_bench.py 2>&1 | grep "f16bf8"; PAIR8_BENCH_ZERO=wx python tools/pair8_bench.py 2>&1 | grep "f16bf8, wave"
  if [ -n "$OLD" ] && [ -f "$OLD" ]; then
    echo "== $OLD"; XVECTOR_HIP_LIB=$OLD python tools/pair8_bench.py 2>&1 | grep "f16bf8, wave"; XVECTOR_HIP_LIB=$OLD PAIR8_BENCH_ZERO=wx python tools/pair8_bench.py 2>&1 | grep "f16bf8, wave"
  fi
done > gpurun_out/p8a/bench.log 2>&1; cat gpurun_out/p8a/bench.log
