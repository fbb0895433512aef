#!/bin/bash
# SQ instruction-mix / wait counters of the fused kernel for one bench configuration (run through gpurun from the repo root):
#   tools/sq_counters.sh <tag> <bench.py args...>       e.g.  tools/sq_counters.sh r04_B8192 --batch-size 8192
# one rocprofv3 --pmc pass per counter group (with --kernel-trace only), summaries in gpurun_out/<tag>_sq_<group>.txt
set -u
TAG=$1; shift
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out; mkdir -p "$OUT"; export TMPDIR=/tmp
ARGS="--only-main --steps 5 --warmup 2 --min-seconds 0 $*"
cd /tmp
g=0
for GROUP in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
             "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64" \
             "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" \
             "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_VALU_CVT SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_ANY"; do
  g=$((g+1))
  rocprofv3 --kernel-trace --pmc $GROUP -d "$OUT/sq_${TAG}_$g" -o bench -- python "$ROOTDIR/bench.py" $ARGS > /dev/null 2>> "$OUT/${TAG}_sq.err"
  DB=$(find "$OUT/sq_${TAG}_$g" -name '*.db' | head -1)
  python "$ROOTDIR/tools/rocpd_summary.py" "$DB" 2>> "$OUT/${TAG}_sq.err" | sed -n '/PMC counters/,$p' | grep -E "gqe_|counter" > "$OUT/${TAG}_sq_$g.txt"
done
cd "$ROOTDIR"; cat "$OUT"/${TAG}_sq_*.txt; tail -2 "$OUT/${TAG}_sq.err"
