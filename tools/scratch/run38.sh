#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
GQE_LIB=$GRAFT_REPO_ROOT/build/oldlib/libgqe.so timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "random_schema_vs_oracle and 16" 2>&1 | grep -v amdgpu | tail -3 > gpurun_out/r38.log
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "random_schema_vs_oracle and 16" 2>&1 | grep -v amdgpu | tail -3 >> gpurun_out/r38.log
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "random_schema_vs_oracle" 2>&1 | grep -v amdgpu | tail -3 >> gpurun_out/r38.log
