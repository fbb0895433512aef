#!/bin/bash
# flakiness check: the whole GPU suite three times, then smoke
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for k in 1 2 3; do
  timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -v "amdgpu.ids\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -12 > gpurun_out/r19_suite$k.log
done
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r19_smoke.log 2>&1
