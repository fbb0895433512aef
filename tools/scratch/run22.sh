#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "lazy" 2>&1 | tail -3 > gpurun_out/r22_lazy_tests.log
timeout 600 python -m pytest tests/test_gpu_shard.py tests/test_gpu_exchange.py tests/test_gpu_limits.py -x -q 2>&1 | grep -v "amdgpu.ids\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -3 >> gpurun_out/r22_lazy_tests.log
for lib in default build/rows4/libgqe.so; do
  if [ $lib = default ]; then unset GQE_LIB; else export GQE_LIB=$GRAFT_REPO_ROOT/$lib; fi
  for rep in 1 2; do
    timeout 300 python bench.py --only-main --lazy-adam --steps 100 --warmup 20 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], {k:v['avg_launch_ms'] for k,v in d['kernels'].items()})" >> gpurun_out/r22_lazy_bench.log
  done
done
unset GQE_LIB
timeout 300 python bench.py --only-main --workload reddit-synth --lazy-adam --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('reddit lazy U2', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], {k:v['avg_launch_ms'] for k,v in d['kernels'].items()})" >> gpurun_out/r22_lazy_bench.log
