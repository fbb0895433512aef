#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 300 python tools/kbench.py --durations 4096 > gpurun_out/r16_dur4096.log 2>&1
timeout 300 python tools/kbench.py --durations 512 > gpurun_out/r16_dur512.log 2>&1
