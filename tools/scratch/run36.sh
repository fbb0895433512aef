#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for rep in 1 2 3; do
for v in base scratch both; do
GQE_LIB=$GRAFT_REPO_ROOT/build/ab/libgqe_$v.so timeout 300 python bench.py --only-main --steps 100 --warmup 20 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v main', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], {k:v['avg_launch_ms'] for k,v in d['kernels'].items()})" >> gpurun_out/r36_ab.log
done
done
for v in base scratch both; do
GQE_LIB=$GRAFT_REPO_ROOT/build/ab/libgqe_$v.so timeout 300 python bench.py --only-main --workload reddit-synth --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v reddit', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], {k:v['avg_launch_ms'] for k,v in d['kernels'].items()})" >> gpurun_out/r36_ab.log
GQE_LIB=$GRAFT_REPO_ROOT/build/ab/libgqe_$v.so timeout 300 python bench.py --only-main --batch-size 8192 --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v B8192', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], {k:v['avg_launch_ms'] for k,v in d['kernels'].items()})" >> gpurun_out/r36_ab.log
GQE_LIB=$GRAFT_REPO_ROOT/build/ab/libgqe_$v.so timeout 300 python bench.py --only-main --lazy-adam --steps 100 --warmup 20 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v lazy', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], {k:v['avg_launch_ms'] for k,v in d['kernels'].items()})" >> gpurun_out/r36_ab.log
done
