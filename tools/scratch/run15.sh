#!/bin/bash
# timeline of the two-queue step: kernel begin/end per queue
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r15
export GQE_BENCH_EVENT_STRIDE=0
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r15/trace -- python $GRAFT_REPO_ROOT/bench.py --only-main --steps 30 --warmup 5 --min-seconds 0 > $GRAFT_REPO_ROOT/gpurun_out/r15/bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/r15/err.log
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/r15/trace -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = len(rows)
sel = rows[n // 2: n // 2 + 40]
t0 = int(sel[0]["Start_Timestamp"])
out = open("gpurun_out/r15/timeline.txt", "w")
for r in sel:
    out.write("%-40s q=%s  start %8.2f  end %8.2f  dur %7.2f us\n" % (r["Kernel_Name"][:40], r.get("Queue_Id", "?"), (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
out.close()
PY
rm -rf gpurun_out/r15/trace
