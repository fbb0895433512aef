#!/usr/bin/env python3
"""Split step (gqe_train_step) against the two-call step on the bench workload: same parameters after K steps?
python tools/probes/split_check.py [--steps 5] [--dim 128]"""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from graphqembed_amd import synth
ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--dim", type=int, default=128)
ap.add_argument("--decoder", default="bilinear-diag")
ap.add_argument("--batch", type=int, default=512)
a = ap.parse_args()
wl = bench.Workload("bio-synth", a.dim, a.decoder, "min", synth.FULL_MIX, a.batch, n_distinct=8)
e1, e2 = wl.engine(), wl.engine()
p1, p2 = wl.prepare(e1), wl.prepare(e2)
for i in range(a.steps):
    e1.run_margin(p1[i % 8]); e1.run_adam(p1[i % 8]["adam"])
    e2.run_train_step(p2[i % 8], p2[i % 8]["adam"])
    torch.cuda.synchronize()
    l1, l2 = p1[i % 8]["losses"].cpu().numpy(), p2[i % 8]["losses"].cpu().numpy()
    print("step", i, "loss", l1[-1], l2[-1], "max |dloss|", np.abs(l1 - l2).max(), flush=True)
print("split steps:", e2.split_steps())
a1, a2 = e1.params.cpu().numpy(), e2.params.cpu().numpy()
for name, x, y in (("params", a1, a2), ("exp_avg", e1.exp_avg.cpu().numpy(), e2.exp_avg.cpu().numpy()),
                   ("exp_avg_sq", e1.exp_avg_sq.cpu().numpy(), e2.exp_avg_sq.cpu().numpy())):
    diff = np.abs(x - y)
    print(name, "max abs diff", diff.max(), "elements differing", int((x != y).sum()), "of", x.size, "max |x|", np.abs(x).max())
    for k, (off, shape) in e1.layout.entries.items():
        n = int(np.prod(shape))
        dd = diff[off:off + n]
        if dd.max() > 0:
            print("   ", k, shape, "max diff", dd.max(), "n differing", int((dd > 0).sum()), "rel", dd.max() / max(np.abs(x[off:off + n]).max(), 1e-30))
print("grads left:", float(e2.grads.abs().max()))
