# A/B of the K = 1 layers of the exact-fp32 paths on 16-channel slabs (tdnn_gemm_k1_kernel, three workgroups per CU) against the 32-channel DMA-fed kernel:
# bench.py's fp32_exact / fp32_toomcook legs (configs[1], same box, alternating) + the layer bench.   bash tools/experiments/fp32_k1_ab.sh
for rep in 1 2 3; do for m in 0 1; do
XV_FP32_K1=$m python bench.py --cpu-budget 0 --e2e-utts 0 --no-extra-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['roofline']['same_arithmetic']
print('XV_FP32_K1=$m  fp32_toomcook %.1f utt/s %.2f ms executed %.4f | fp32_exact %.1f utt/s %.2f ms %.4f' % (s['fp32_toomcook']['utt_s'], s['fp32_toomcook']['ms_per_step'], s['fp32_toomcook']['frac_of_157.3TF_executed'], s['fp32_exact']['utt_s'], s['fp32_exact']['ms_per_step'], s['fp32_exact']['frac_of_157.3TF_executed']))"
done; done
for m in 0 1; do echo "== XV_FP32_K1=$m"; XV_FP32_K1=$m python tools/fp32_layer_bench.py 2>/dev/null | grep "K=1\|all five"; done
