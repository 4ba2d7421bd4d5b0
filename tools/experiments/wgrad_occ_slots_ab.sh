for rep in 1 2 3; do for cfg in "2 512" "3 768" "2 768"; do set -- $cfg
XV_WGRAD_OCC=$1 XV_WGRAD_SLOTS=$2 python bench.py --mode train --train-precision bf16x3 --steps 300 --warmup 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('XV_WGRAD_OCC=$1 XV_WGRAD_SLOTS=$2: %.4f ms/step  last_loss %.6f' % (d['ms_per_step'], d['last_loss']))"
done; done
