# timing-only variants of the pair kernel against the kernel itself (tools/experiments/pair8_variants.py builds them)
for v in base noexch nopool noswap nobar nodma noconv base; do
  echo "== $v"
  if [ $v = base ]; then L=""; else L="build/variants/libxv_p8_$v.so"; fi
  XVECTOR_HIP_LIB=$L python tools/pair8_bench.py 2>&1 | grep "f16bf8, wave"
  XVECTOR_HIP_LIB=$L PAIR8_BENCH_ZERO=wx python tools/pair8_bench.py 2>&1 | grep "f16bf8, wave" | sed 's/^/   zeros: /'
done
