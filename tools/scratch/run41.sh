#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_limits.py -x -q -k "tuning" 2>&1 | grep -v amdgpu | tail -5 > gpurun_out/r41.log
