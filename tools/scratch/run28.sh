#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "256 or reddit or config5" 2>&1 | tail -2 > gpurun_out/r28_parity.log
for rep in 1 2; do
timeout 300 python bench.py --only-main --workload reddit-synth --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('reddit', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], {k:v['avg_launch_ms'] for k,v in d['kernels'].items()})" >> gpurun_out/r28_bench.log
done
timeout 600 python tools/kbench.py --durations 512 --workload reddit-synth --dim 256 2>&1 | grep -v amdgpu | tail -10 > gpurun_out/r28_dur_reddit.log
