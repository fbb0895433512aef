#!/bin/bash
# kernel trace of tools/probes/step_probe.py under the given switches: tools/probes/trace_probe.sh <tag> [step_probe args...]
# (environment variables of the caller reach the probe) -> gpurun_out/<tag>_kernel_stats.txt
set -u
TAG=$1; shift
ROOTDIR=$(pwd)
OUT=$ROOTDIR/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -d "$OUT/prof_$TAG" -o probe -- python "$ROOTDIR/tools/probes/step_probe.py" "$@" > "$OUT/${TAG}_stdout.log" 2> "$OUT/${TAG}_prof.err"
DB=$(find "$OUT/prof_$TAG" -name '*.db' | head -1)
python "$ROOTDIR/tools/rocpd_summary.py" "$DB" > "$OUT/${TAG}_kernel_stats.txt" 2>> "$OUT/${TAG}_prof.err"
rm -rf "$OUT/prof_$TAG"
cd "$ROOTDIR"
tail -1 "$OUT/${TAG}_stdout.log"
head -8 "$OUT/${TAG}_kernel_stats.txt" | cut -c1-150
grep -A6 "steady state" "$OUT/${TAG}_kernel_stats.txt" | cut -c1-110
