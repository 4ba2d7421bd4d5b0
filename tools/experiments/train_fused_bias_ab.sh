timeout 600 python -m pytest tests/test_gpu_training.py tests/test_gpu_fuzz.py -q -x -p no:cacheprovider -k "wgrad or gradients or training or step or refgraph or three_steps" 2>&1 | tail -3
for rep in 1 2 3; do for m in 0 1; do
XVECTOR_TRAIN_FUSED_BIAS=$m python bench.py --mode train --train-precision bf16x3 --steps 300 --warmup 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('XVECTOR_TRAIN_FUSED_BIAS=$m: %.4f ms/step  last_loss %.6f' % (d['ms_per_step'], d['last_loss']))"
done; done
