#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3 > gpurun_out/r25_parity.log
for B in 8192 4096 2048; do
timeout 300 python bench.py --only-main --batch-size $B --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('B$B', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], {k:v['avg_launch_ms'] for k,v in d['kernels'].items()})" >> gpurun_out/r25_bench.log
done
timeout 300 python tools/kbench.py --durations 4096 2>&1 | grep -v amdgpu | tail -11 > gpurun_out/r25_dur4096.log
