#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; rm -f gpurun_out/r42.log
for k in 1 2 3 4 5; do
  timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -v "amdgpu.ids\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -1 >> gpurun_out/r42.log
done
