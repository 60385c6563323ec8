for cfg in "" "KVQ_TAIL_MAXC=192"; do
  echo "=== C2 $cfg"
  env $cfg python bench.py --no-cpu-baseline --legs c2 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('C2', round(d['value'],1), round(d['ms_per_step'],4), 'step_gpu_ms', d['roofline']['step_gpu_ms'])
for k,v in d['roofline']['by_kernel_ms_per_step'].items(): print('   %-60s %.4f'%(k,v))
"
  echo "=== C5 $cfg"
  env $cfg python tools/swinb_probe.py 4 table 2>&1 | tail -25
done
