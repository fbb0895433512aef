#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -2 > gpurun_out/r27_parity.log
for args in "--batch-size 8192" "--workload reddit-synth" ""; do
timeout 300 python bench.py --only-main $args --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$args', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], {k:v['avg_launch_ms'] for k,v in d['kernels'].items()})" >> gpurun_out/r27_bench.log
done
