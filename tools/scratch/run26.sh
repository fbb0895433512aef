#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for B in 512 640 768 1024 1536 2048; do
for T in 0 100000; do
GQE_DEBUG_FW8_MIN_TILES=$T timeout 300 python bench.py --only-main --batch-size $B --steps 20 --warmup 5 --min-seconds 0.2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('B$B thr$T', d['value'], d['ms_per_step'], {k:v['avg_launch_ms'] for k,v in d['kernels'].items()})" >> gpurun_out/r26_sweep.log
done
done
for dec in transe bilinear; do
for T in 0 100000; do
GQE_DEBUG_FW8_MIN_TILES=$T timeout 300 python bench.py --only-main --decoder $dec --batch-size 2048 --steps 20 --warmup 5 --min-seconds 0.2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$dec B2048 thr$T', d['value'], d['ms_per_step'], {k:v['avg_launch_ms'] for k,v in d['kernels'].items()})" >> gpurun_out/r26_sweep.log
done
done
