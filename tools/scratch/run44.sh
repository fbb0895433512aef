#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 300 python tools/scratch/mall_probe.py > gpurun_out/r44_mall.log 2>&1
