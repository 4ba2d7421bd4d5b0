# three builds of the pair kernel alternating in one call: build/variants/libxv_p8_head.so (another build of the library), the tree's
# library with one tile per workgroup and with XV_PAIR8_PERSISTENT=1 (only the experiment of xv_pair8_persistent.patch reads that)
for i in 1 2; do for m in head loop persistent; do echo "== $m"; if [ $m = head ]; then L=build/variants/libxv_p8_head.so; P=0; elif [ $m = loop ]; then L=""; P=0; else L=""; P=1; fi
XV_PAIR8_PERSISTENT=$P XVECTOR_HIP_LIB=$L python tools/pair8_bench.py 2>&1 | grep "f16bf8, wave"; XV_PAIR8_PERSISTENT=$P XVECTOR_HIP_LIB=$L PAIR8_BENCH_ZERO=wx python tools/pair8_bench.py 2>&1 | grep "f16bf8, wave"; done; done
