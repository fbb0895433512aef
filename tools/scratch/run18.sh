#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_shard.py -x -q 2>&1 | tail -15 > gpurun_out/r18_shard.log
timeout 600 python -m pytest tests/test_gpu_api.py tests/test_gpu_limits.py -x -q 2>&1 | tail -5 > gpurun_out/r18_api.log
