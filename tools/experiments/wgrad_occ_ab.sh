# A/B: the bf16x3 weight-gradient kernel with its register budget cut for two / three workgroups per CU (XV_WGRAD_OCC), training step of configs[4]
# shapes (bench.py --mode train --train-precision bf16x3, 300 steps after 30), three alternating rounds.   bash tools/experiments/wgrad_occ_ab.sh
for rep in 1 2 3; do for m in 2 3; do
XV_WGRAD_OCC=$m python bench.py --mode train --train-precision bf16x3 --steps 300 --warmup 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('XV_WGRAD_OCC=$m: %.4f ms/step  last_loss %.6f' % (d['ms_per_step'], d['last_loss']))"
done; done
