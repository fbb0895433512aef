#!/usr/bin/env python3
"""Step time of one engine (run_margin + run_adam on the default stream, no library events), for A / B runs of library
switches or builds (GQE_LIB=...): python tools/probes/step_probe.py [--dim 128] [--decoder bilinear-diag] [--batch 512]
[--workload bio-synth]"""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from graphqembed_amd import synth
ap = argparse.ArgumentParser()
ap.add_argument("--dim", type=int, default=128)
ap.add_argument("--decoder", default="bilinear-diag")
ap.add_argument("--batch", type=int, default=512)
ap.add_argument("--workload", default="bio-synth")
ap.add_argument("--mix", default="full", help="full | 1-chain (the edge-only burn-in step, SURVEY C1)")
ap.add_argument("--defer", action="store_true", help="gqe_set_deferred_gemm: the pair GEMM rides in the Adam pass's launch")
ap.add_argument("--lazy", action="store_true", help="gqe_set_lazy_adam (with the next feed declared every step: gqe_lazy_prefetch)")
ap.add_argument("--train-step", action="store_true", help="gqe_train_step: one call per iteration (the split step where it applies)")
ap.add_argument("--zipf", type=float, default=None, help="heavy-tailed degrees / word frequencies: 1 / rank^zipf")
a = ap.parse_args()
wl = bench.Workload(a.workload, a.dim, a.decoder, "min", synth.FULL_MIX if a.mix == "full" else (synth.FULL_MIX[0],), a.batch, zipf=a.zipf)
eng = wl.engine(lazy=a.lazy)
prep = wl.prepare(eng)
if a.defer:
    eng.set_deferred_gemm(True)
n = wl.n_distinct


def run(k0, k):
    for i in range(k0, k0 + k):
        ps = prep[i % n]
        if a.train_step:
            eng.run_train_step(ps, ps["adam"])
            continue
        eng.run_margin(ps)
        if a.lazy and os.environ.get("STEP_PROBE_NO_PREFETCH") is None:
            eng.lazy_prefetch(prep[(i + 1) % n])
        eng.run_adam(ps["adam"])


run(0, 50)
torch.cuda.synchronize()
if os.environ.get("STEP_PROBE_TIMING"):      # the library's hipEvent brackets per launch kind (include/gqe.h, gqe_timing_read)
    eng.timing_enable(8)
    run(50, 400)
    torch.cuda.synchronize()
    print("event brackets (us): " + ", ".join("%d: %.1f x%d" % ((k,) + (lambda r: (r[0] * 1e3, r[1]))(eng.timing_read(k))) for k in range(7)), flush=True)
    eng.timing_enable(0)
ts = []
for rep in range(20):
    t0 = time.perf_counter()
    run(50 + 100 * rep, 100)
    eng.sync()
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) / 100)
loss = float(prep[(50 + 2000 - 1) % n]["losses"][-1].item())
print(("lazy (%d rides) " % eng.gemm_rides() if a.lazy else "") + ("train_step (%d split) " % eng.split_steps() if a.train_step else "") + ("deferred GEMM " if a.defer else "") + "%s d=%d %s B=%d: %.1f us/step (median of 20 x 100), final loss %.6f, params checksum %.6f"
      % (a.workload, a.dim, a.decoder, a.batch, np.median(ts) * 1e6, loss, float(eng._params.double().abs().sum())), flush=True)
if a.zipf:
    print("hot rows %d, sub-list heads %d (fused launches link onto them: %s)" % ((eng.hot_rows(),) + eng.hot_sub_lists()), flush=True)
