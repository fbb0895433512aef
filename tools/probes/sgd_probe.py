#!/usr/bin/env python3
"""Step time with --opt sgd (bio/train.py:59-60: torch.optim.SGD, momentum 0): gqe_margin_fwd_bwd + gqe_sgd_step on the headline workload.
python tools/probes/sgd_probe.py [--defer]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from graphqembed_amd import synth
wl = bench.Workload("bio-synth", 128, "bilinear-diag", "min", synth.FULL_MIX, 512)
eng = wl.engine()
prep = wl.prepare(eng)
n = wl.n_distinct
if "--defer" in sys.argv:
    eng.set_deferred_gemm(True)


def run(k0, k):
    for i in range(k0, k0 + k):
        ps = prep[i % n]
        eng.run_margin(ps)
        eng._check(eng.lib.gqe_sgd_step(eng.ctx, ps["adam"]["arr"], ps["adam"]["n"], 0.01, eng._stream()))


run(0, 50)
torch.cuda.synchronize()
if os.environ.get("STEP_PROBE_TIMING"):
    eng.timing_enable(8)
    run(50, 400)
    torch.cuda.synchronize()
    print("event brackets (us): " + ", ".join("%d: %.1f x%d" % ((k,) + (lambda r: (r[0] * 1e3, r[1]))(eng.timing_read(k))) for k in range(5)), flush=True)
    eng.timing_enable(0)
ts = []
for rep in range(20):
    t0 = time.perf_counter()
    run(50 + 100 * rep, 100)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) / 100)
print("sgd step: %.1f us/step (median of 20 x 100) = %.1f M queries/s" % (np.median(ts) * 1e6, wl.qpi / np.median(ts) / 1e6))
