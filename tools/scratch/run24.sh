#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3 > gpurun_out/r24_parity.log
timeout 300 python bench.py --only-main --workload reddit-synth --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('reddit', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], {k:v['avg_launch_ms'] for k,v in d['kernels'].items()})" > gpurun_out/r24_bench.log
timeout 300 python bench.py --only-main --batch-size 8192 --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('B8192', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], {k:v['avg_launch_ms'] for k,v in d['kernels'].items()})" >> gpurun_out/r24_bench.log
timeout 300 python bench.py --only-main --steps 100 --warmup 20 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('main', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], {k:v['avg_launch_ms'] for k,v in d['kernels'].items()})" >> gpurun_out/r24_bench.log
timeout 600 python tools/kbench.py --durations 512 --workload reddit-synth --dim 256 2>&1 | grep -v amdgpu | tail -10 > gpurun_out/r24_dur_reddit.log
