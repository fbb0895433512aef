#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_shard.py tests/test_gpu_rccl.py -x -q 2>&1 | grep -v "amdgpu.ids\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -8 > gpurun_out/r20_shard_tests.log
GQE_SHARD_PROFILE=1 timeout 600 python tools/shard_overhead_bench.py > gpurun_out/r20_shard_overhead.log 2>&1
