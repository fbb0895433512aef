#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "eight_wave or baseline_configs" 2>&1 | tail -2 > gpurun_out/r43_parity.log
for B in 8192 4096 2048 512; do
timeout 300 python bench.py --only-main --batch-size $B --steps 20 --warmup 5 --min-seconds 0.2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('B$B', d['value'], d['ms_per_step'], {k:v['avg_launch_ms'] for k,v in d['kernels'].items()}, d['kernels']['pair_gemm']['mfma_TFs'])" >> gpurun_out/r43_bench.log
done
timeout 300 python bench.py --only-main --workload reddit-synth --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('reddit', d['value'], d['ms_per_step'], {k:v['avg_launch_ms'] for k,v in d['kernels'].items()}, d['kernels']['pair_gemm']['mfma_TFs'])" >> gpurun_out/r43_bench.log
