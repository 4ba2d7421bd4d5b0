# A/B: s_setprio 2 around the MFMA loops of the exact-fp32 kernels (tdnn_gemm_k1_kernel, tdnn_gemm_dma_kernel, tdnn_gemm_toom_kernel), 0 in their epilogues
# (XV_FP32_PRIO=1): does the other workgroups' MFMA wave win the issue port against an epilogue wave on the same SIMD?   bash tools/experiments/fp32_prio_ab.sh
for rep in 1 2; do for m in 0 2; do echo "== XV_FP32_PRIO=$m"; XV_FP32_PRIO=$m python tools/fp32_layer_bench.py 2>/dev/null | grep -v "24 ->"; XV_FP32_PRIO=$m python tools/toom_layer_bench.py 2>/dev/null | cut -c1-60,90-170; done; done
