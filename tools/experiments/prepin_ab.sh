mkdir -p gpurun_out
for rep in 1 2; do for m in 0 1; do
XVECTOR_PREPIN=$m timeout 300 python bench.py --no-fp32-leg --cpu-budget 0 --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); cj=d['cli_job']; rh=cj.get('rehearsal_8x125k',{})
b=cj['breakdown_s']
print('PREPIN=$m cli_job wall %.3f (gloo %.3f files %.3f) | weights+probe %.3f first window %.3f extraction %.3f | rehearsal wall %.2f rank0: first window %.2f pg wait %.2f gather %.2f' % (
 cj['wall_s'], cj['gloo_gather']['wall_s'], cj['shard_files']['wall_s'], b.get('weights packed on the device + accuracy probe',0), b.get('first window launched',0), [v for k,v in b.items() if k.startswith('extraction')][0],
 rh.get('wall_s',0), rh.get('breakdown_s_rank0',{}).get('first window launched',0), rh.get('breakdown_s_rank0',{}).get('wait for the process group',0), rh.get('breakdown_s_rank0',{}).get('gather',0)))
" | tee -a gpurun_out/prepin_ab.txt
done; done
