#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "eight_wave" 2>&1 | tail -3 ) > gpurun_out/r30_parity.log 2>&1
