#!/bin/bash
# Collect the rocprofv3 evidence of a round on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh r02 bio-synth   |   tools/profile_round.sh r02 reddit-synth
# kernel trace (+ stats) of the main measurement, then FETCH_SIZE and WRITE_SIZE in separate --pmc passes
# (MI355X_MICROARCH.md: HBM bytes = 2 * FETCH_SIZE KiB * 1024 on gfx950 reads + WRITE_SIZE KiB * 1024).
# Summaries land in gpurun_out/<tag>_<workload>_*.txt; copy the ones to keep into profiles/.
set -u
TAG=${1:-r02}
WL=${2:-bio-synth}
ROOTDIR=$(pwd)
OUT=$ROOTDIR/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
ARGS="--workload $WL --only-main --steps 20 --warmup 5 --min-seconds 0.02"
cd /tmp
rocprofv3 --kernel-trace --stats -d "$OUT/prof_${TAG}_${WL}" -o bench -- python "$ROOTDIR/bench.py" $ARGS > "$OUT/${TAG}_${WL}_stdout.json" 2> "$OUT/${TAG}_${WL}_prof.err"
DB=$(find "$OUT/prof_${TAG}_${WL}" -name '*.db' | head -1)
python "$ROOTDIR/tools/rocpd_summary.py" "$DB" > "$OUT/${TAG}_${WL}_kernel_stats.txt" 2>> "$OUT/${TAG}_${WL}_prof.err"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C -d "$OUT/pmc_${TAG}_${WL}_$C" -o bench -- python "$ROOTDIR/bench.py" $ARGS > /dev/null 2>> "$OUT/${TAG}_${WL}_prof.err"
  DB=$(find "$OUT/pmc_${TAG}_${WL}_$C" -name '*.db' | head -1)
  python "$ROOTDIR/tools/rocpd_summary.py" "$DB" > "$OUT/${TAG}_${WL}_pmc_$C.txt" 2>> "$OUT/${TAG}_${WL}_prof.err"
done
cd "$ROOTDIR"
tail -3 "$OUT/${TAG}_${WL}_prof.err"
