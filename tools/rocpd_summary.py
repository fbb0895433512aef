#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.x rocpd sqlite) result: per-kernel calls / total / average
duration, and per-kernel PMC counter averages when a --pmc pass was recorded.
    python tools/rocpd_summary.py gpurun_out/prof_r1/bench_results.db > profiles/r01_kernel_stats.txt
"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    print("# source: %s" % path)
    print("# kernel-trace summary (rocpd view top_kernels; durations in microseconds)")
    print("%-90s %8s %14s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    for name, calls, total, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels order by total_duration desc"):
        print("%-90s %8d %14d %12.1f %7.2f" % (name[:90], calls, total, avg, pct))
    print()
    print("# per-kernel launch geometry / registers (first dispatch of each kernel)")
    q = ("select name,grid_x,workgroup_x,lds_size,scratch_size,vgpr_count,accum_vgpr_count,sgpr_count,"
         "min(duration),avg(duration),max(duration),count(*) from kernels group by name order by sum(duration) desc")
    print("%-60s %9s %5s %7s %7s %5s %5s %5s %10s %10s %10s %6s" % ("kernel", "grid", "wg", "lds", "scratch", "vgpr", "agpr", "sgpr", "min_ns", "avg_ns", "max_ns", "n"))
    for r in cur.execute(q):
        print("%-60s %9d %5d %7d %7d %5d %5d %5d %10d %10.0f %10d %6d" % ((r[0][:60],) + tuple(r[1:])))
    # the stream's timeline in steady state: per kernel, the median duration and the median gap between the previous dispatch's
    # end and this dispatch's start (the last 60 % of the dispatches: warm-up and initialisation kernels stay out of the medians)
    try:
        disp = list(cur.execute("select name, start, end from kernels order by start"))
    except sqlite3.Error:
        disp = []
    if len(disp) > 20:
        import statistics
        tail = disp[int(len(disp) * 0.4):]
        per = {}
        for prev, d in zip(tail[:-1], tail[1:]):
            per.setdefault(d[0], []).append((d[2] - d[1], d[1] - prev[2]))
        print()
        print("# steady state (last 60 %% of %d dispatches): median duration and median idle gap in front of each kernel, ns" % len(disp))
        print("%-60s %8s %12s %12s" % ("kernel", "n", "duration", "gap_before"))
        for name, v in sorted(per.items(), key=lambda kv: -sum(x[0] for x in kv[1])):
            if len(v) >= 10:
                print("%-60s %8d %12.0f %12.0f" % (name[:60], len(v), statistics.median(x[0] for x in v), statistics.median(x[1] for x in v)))
    try:
        rows = list(cur.execute("select * from counters_collection limit 1"))
        cols = [d[0] for d in cur.description]
    except sqlite3.Error:
        rows, cols = [], []
    if rows and "counter_name" in cols and "kernel_name" in cols:
        print()
        print("# PMC counters: per-kernel average of the per-dispatch counter value")
        vcol = "value"
        print("# (FETCH_SIZE / WRITE_SIZE are in KiB per dispatch; on gfx950 FETCH_SIZE counts a wide coalesced read at half")
        print("#  its bytes - MI355X_MICROARCH.md, HBM section - so HBM read bytes ~= 2 * FETCH_SIZE * 1024)")
        print("%-60s %-16s %16s %8s" % ("kernel", "counter", "avg_value", "n"))
        for r in cur.execute("select kernel_name,counter_name,avg(%s),count(*) from counters_collection group by kernel_name,counter_name order by kernel_name" % vcol):
            print("%-60s %-16s %16.1f %8d" % (r[0][:60], r[1], r[2], r[3]))


if __name__ == "__main__":
    main(sys.argv[1])
