#!/bin/bash
# two-queue step: parity test, then the bench with and without it
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "two_queue" -s 2>&1 | tail -15 > gpurun_out/r14_test.log
timeout 300 python bench.py --only-main --steps 100 --warmup 20 > gpurun_out/r14_bench_overlap.json 2> gpurun_out/r14_bench_overlap.err
timeout 300 python bench.py --only-main --steps 100 --warmup 20 --no-overlap > gpurun_out/r14_bench_one.json 2> gpurun_out/r14_bench_one.err
timeout 300 python bench.py --only-main --steps 100 --warmup 20 > gpurun_out/r14_bench_overlap2.json 2>> gpurun_out/r14_bench_overlap.err
