# A/B of the training step's two-stream arrangement (bf16x3, configs[4] shapes, 300 steps after 30): the weight side queued in front of /
# behind the input-gradient GEMM it runs beside (XVECTOR_TRAIN_WGRAD_AFTER), the step on a high-priority stream (the device offers
# priorities 0 and -1: the side stream cannot be LOWER than the default, the step can be higher).   bash tools/experiments/train_stream_ab.sh
mkdir -p gpurun_out
: > gpurun_out/train_ab.txt
for rep in 1 2 3; do
for cfg in "none 0" "none 1" "high 0" "high 1"; do
  set -- $cfg
  XVECTOR_TRAIN_MAIN_PRIORITY=$1 XVECTOR_TRAIN_WGRAD_AFTER=$2 timeout 120 python bench.py --mode train --train-precision bf16x3 --steps 300 --warmup 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('main priority $1 wgrad_after $2: %.4f ms/step  last_loss %.5f' % (d['ms_per_step'], d['last_loss']))" >> gpurun_out/train_ab.txt
done; done
cat gpurun_out/train_ab.txt
