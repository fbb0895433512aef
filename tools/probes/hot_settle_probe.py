#!/usr/bin/env python3
"""How long does the hot-row set of a heavy-tailed workload take to settle, and what does the fused launch cost on the way?
python tools/probes/hot_settle_probe.py [--workload reddit-synth] [--dim 256] [--zipf 1.0] [--blocks 12]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from graphqembed_amd import synth
ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="reddit-synth")
ap.add_argument("--dim", type=int, default=256)
ap.add_argument("--zipf", type=float, default=1.0)
ap.add_argument("--blocks", type=int, default=12)
ap.add_argument("--distinct", type=int, default=16)
a = ap.parse_args()
wl = bench.Workload(a.workload, a.dim, "bilinear-diag", "min", synth.FULL_MIX, 512, n_distinct=a.distinct, zipf=a.zipf)
eng = wl.engine()
prep = wl.prepare(eng)
eng.timing_enable(1)
i = 0
for blk in range(a.blocks):
    for k in range(7):
        eng.timing_read(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        ps = prep[i % wl.n_distinct]
        eng.run_margin(ps)
        eng.run_adam(ps["adam"])
        i += 1
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / 50
    ms, n = eng.timing_read(0)
    print("steps %4d..%4d: step %.1f us, fused bracket %.1f us (%d launches), hot rows %d" % (i - 50, i, el * 1e6, ms * 1e3, n, eng.hot_rows()), flush=True)
