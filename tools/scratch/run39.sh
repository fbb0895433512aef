#!/bin/bash
# every GPU test file in its own process (order independence), then the suite in reverse file order
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; rm -f gpurun_out/r39.log
for f in tests/test_gpu_*.py; do
  r=$(timeout 900 python -m pytest $f -q -m gpu -p no:cacheprovider 2>&1 | grep -v "amdgpu.ids\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -1)
  echo "$f: $r" >> gpurun_out/r39.log
done
