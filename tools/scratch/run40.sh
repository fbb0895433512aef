#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for rep in 1 2 3; do
for v in base mv1 mv2 mv3; do
GQE_LIB=$GRAFT_REPO_ROOT/build/ab/libgqe_$v.so timeout 300 python bench.py --only-main --steps 100 --warmup 20 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v main', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], {k:v['avg_launch_ms'] for k,v in d['kernels'].items()})" >> gpurun_out/r40_ab.log
done
done
