#!/bin/bash
# HBM traffic of the split step's second launch by sub-phase: FETCH_SIZE / WRITE_SIZE passes (rocprofv3 --pmc, separate runs) of the
# headline bench under GQE_SPLIT_DEBUG_B = 0 (everything) / 1 (no pair-GEMM units) / 2 (no named rows) / 3 (neither) — the debug
# settings give WRONG training results by design; only the traffic of gqe_split_rows_kernel is read.   (run through gpurun)
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out; mkdir -p "$OUT"; export TMPDIR=/tmp
ARGS="--only-main --steps 20 --warmup 5 --min-seconds 0.02"
cd /tmp
for DBG in 0 1 2 3; do
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf "$OUT/pmc_b_${DBG}_$C"
    GQE_SPLIT_DEBUG_B=$DBG rocprofv3 --kernel-trace --pmc $C -d "$OUT/pmc_b_${DBG}_$C" -o bench -- python "$ROOTDIR/bench.py" $ARGS > /dev/null 2>> "$OUT/split_b_traffic.err"
    DB=$(find "$OUT/pmc_b_${DBG}_$C" -name '*.db' | head -1)
    echo "== GQE_SPLIT_DEBUG_B=$DBG $C"
    python "$ROOTDIR/tools/rocpd_summary.py" "$DB" 2>> "$OUT/split_b_traffic.err" | grep "gqe_split_rows_kernel\|gqe_fused_kernel\|gqe_prestep" | cut -c1-60,90-200
  done
done
