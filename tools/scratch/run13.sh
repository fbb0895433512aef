for v in default 0 1; do
  if [ $v = default ]; then unset HIP_FORCE_DEV_KERNARG; else export HIP_FORCE_DEV_KERNARG=$v; fi
  for rep in 1 2; do
  python bench.py --only-main --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('KERNARG=$v', d['value'], d['ms_per_step'], {k:v['avg_launch_ms'] for k,v in d['kernels'].items()}, d['roofline']['avg_launch_ms'])"
  done
done
