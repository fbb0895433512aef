#!/usr/bin/env python3
"""Copy what tools/profile_round.sh left under gpurun_out/ into profiles/ (tracked) and write the PMC traffic summary.
    python tools/collect_profiles.py r02c"""
import json, os, re, shutil, sys
tag = sys.argv[1]
out = {}
for wl in ("bio-synth", "reddit-synth"):
    vals = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for l in open("gpurun_out/%s_%s_pmc_%s.txt" % (tag, wl, c)):
            m = re.match(r"(?:void )?(gqe_\w+).*\s%s\s+([\d.]+)\s+(\d+)\s*$" % c, l)
            if m:
                vals.setdefault(m.group(1), {})[c] = float(m.group(2))
    out[wl] = {k: {"fetch_kib": v["FETCH_SIZE"], "write_kib": v["WRITE_SIZE"],
                   "hbm_bytes_per_launch": int((2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024)} for k, v in vals.items()}
    # the kernels' own average durations in the kernel trace (bench.py prints them as rocprof_avg_launch_ms next to its hipEvent brackets)
    for l in open("gpurun_out/%s_%s_kernel_stats.txt" % (tag, wl)):
        m = re.match(r"(?:void )?(gqe_\w+)\S*.*?\s+(\d+)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s*$", l)
        if m and m.group(1) in out[wl] and "rocprof_avg_us" not in out[wl][m.group(1)]:   # first = largest total of that kernel
            out[wl][m.group(1)]["rocprof_avg_us"] = float(m.group(4))
            out[wl][m.group(1)]["rocprof_calls"] = int(m.group(2))
    for suffix in ("kernel_stats", "pmc_FETCH_SIZE", "pmc_WRITE_SIZE"):
        src, dst = "gpurun_out/%s_%s_%s.txt" % (tag, wl, suffix), "profiles/%s_%s_%s.txt" % (tag, wl, suffix)
        text = open(src).read().replace("/tmp/code/williamleif__graphqembed/repo/", "").replace("/root/repo/", "")
        open(dst, "w").write(text)
    line = [l for l in open("gpurun_out/%s_%s_stdout.json" % (tag, wl)) if l.startswith("{")][0]
    open("profiles/%s_%s_bench_line.json" % (tag, wl), "w").write(line)
out["source"] = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, each with --kernel-trace) of `python bench.py --workload <w> "
                 "--only-main --steps 20 --warmup 5 --min-seconds 0.02` (tools/profile_round.sh %s); HBM bytes per launch = 2 * FETCH_SIZE KiB * 1024 "
                 "(gfx950 correction for wide coalesced reads, MI355X_MICROARCH.md HBM section) + WRITE_SIZE KiB * 1024" % tag)
json.dump(out, open("profiles/%s_pmc_traffic.json" % tag, "w"), indent=1)
for name in ("bench_default", "bench_2rank_gloo"):
    p = "gpurun_out/%s_%s.json" % (tag, name)
    if os.path.exists(p):
        lines = [l for l in open(p) if l.startswith("{")]
        if lines:
            open("profiles/%s_%s_line.json" % (tag, name), "w").write(lines[0])
print(json.dumps({w: {k: (v["hbm_bytes_per_launch"], v.get("rocprof_avg_us")) for k, v in out[w].items()} for w in out if w != "source"}))
