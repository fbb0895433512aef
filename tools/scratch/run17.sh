#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python tools/kbench.py --durations 512 --workload reddit-synth --dim 256 > gpurun_out/r17_dur_reddit.log 2>&1
