#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "baseline_configs or bio_synth or random_schema" 2>&1 | tail -8 > gpurun_out/r37_parity.log
for rep in 1 2 3; do
timeout 120 python bench.py --only-main --steps 100 --warmup 20 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('ink main', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], {k:v['avg_launch_ms'] for k,v in d['kernels'].items()}, d['final_loss'])" >> gpurun_out/r37_bench.log
GQE_DEBUG_NO_INK=1 timeout 120 python bench.py --only-main --steps 100 --warmup 20 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('sep main', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], {k:v['avg_launch_ms'] for k,v in d['kernels'].items()}, d['final_loss'])" >> gpurun_out/r37_bench.log
done
