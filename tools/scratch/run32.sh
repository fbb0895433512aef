#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for pad in 0 20000; do
GQE_DEBUG_LDS_PAD=$pad timeout 300 python tools/kbench.py --durations 4096 2>&1 | grep -v amdgpu | tail -11 | head -4 >> gpurun_out/r32_pad.log
done
