#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 300 python tools/stream_probe.py 2>&1 | grep "75 MB\|100 MB" > gpurun_out/r45.log
for rep in 1 2; do
timeout 300 python bench.py --only-main --steps 100 --warmup 20 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('main', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], {k:v['avg_launch_ms'] for k,v in d['kernels'].items()})" >> gpurun_out/r45.log
done
timeout 300 python tools/opt_scale_bench.py 2>&1 | grep -v amdgpu | tail -8 >> gpurun_out/r45.log
