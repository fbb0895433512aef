#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
GQE_SHARD_PROFILE=1 timeout 600 python tools/shard_overhead_bench.py > gpurun_out/r21_shard_overhead.log 2>&1
GQE_SHARD_PROFILE=1 timeout 900 python bench.py --gpus 2 --backend gloo --steps 20 --warmup 5 --no-reddit > gpurun_out/r21_bench2.json 2> gpurun_out/r21_bench2.err
timeout 900 python -m pytest tests/test_gpu_shard.py tests/test_gpu_rccl.py -x -q 2>&1 | grep -v "amdgpu.ids\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -4 > gpurun_out/r21_tests.log
