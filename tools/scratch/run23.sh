#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r23_bench_default.json 2> gpurun_out/r23_bench_default.err ) 2> gpurun_out/r23_time.log
