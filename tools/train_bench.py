#!/usr/bin/env python3
"""End-to-end training throughput with the HOST feed (sampling + packing on the host cores, index upload on
libgqe's side stream): graphqembed_amd.trainer.TensorizedTrainer on bio-synth, full mix, d=128.
Complements bench.py (which times the device path with pre-staged index feeds)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import build_layout, init_params
from graphqembed_amd import synth
from graphqembed_amd.engine import Engine
from graphqembed_amd.tensorize import FormulaPlan, pack_margin_batches
from graphqembed_amd.trainer import TensorizedTrainer

d, dec, inter, B = 128, "bilinear-diag", "min", 512
g = synth.bio_synth(seed=0)
layout = build_layout(g, d, dec, inter)
eng = Engine(d, dec, inter, layout, max_queries=9 * B, max_batches=9)
init_params(eng, d, 0)
types = ["1-chain", "2-chain", "3-chain", "2-inter", "3-inter", "3-inter_chain"]
pools = synth.make_pools(g, types, formulas_per_type=6, pool_size=32768, seed=0)


class Shim(object):                       # margin_step / optimiser on the bare engine
    def __init__(self):
        self.plans, self.touched = {}, set()

    def margin_step(self, items):
        packed = []
        for (f, t, ng, a, w, m) in items:
            if f not in self.plans:
                self.plans[f] = FormulaPlan(f, layout, inter)
            packed.append((self.plans[f], t, ng, a, w, m))
            self.touched |= self.plans[f].touched
        descs, idx, n = pack_margin_batches(packed)
        return eng.margin_fwd_bwd(descs, idx, n)

    def step(self):
        eng.adam_step(self.touched)
        self.touched = set()


shim = Shim()
all_rows = {m: np.arange(1, g.mode_sizes[m] + 1, dtype=np.int32) for m in g.modes}
tr = TensorizedTrainer(shim, shim, pools, all_rows, batch_size=B, seed=0)
tr.run(50, log_every=0)
torch.cuda.synchronize()
n0, t0 = tr.queries_seen, time.perf_counter()
last = tr.run(500, log_every=0)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("host-fed training: %d iterations, %.1f us/iteration, %.2f M queries/s, final loss %.5f"
      % (500, dt / 500 * 1e6, (tr.queries_seen - n0) / dt / 1e6, float(last[-1].item())))
# host share: sampling + packing only
t0 = time.perf_counter()
for it in range(200):
    items = tr.items(it)
    packed = [(shim.plans[f], t, ng, a, w, m) for (f, t, ng, a, w, m) in items]
    pack_margin_batches(packed)
print("host sampling + packing alone: %.1f us/iteration" % ((time.perf_counter() - t0) / 200 * 1e6))

# ---- the same loop with the native feeder (C++ sampling + packing + launches, one call for 500 iterations) ----
eng2 = Engine(d, dec, inter, layout, max_queries=9 * B, max_batches=9)
init_params(eng2, d, 0)
plans2 = {}
plist = []
for t in types:
    for p in pools[t]:
        plist.append((FormulaPlan(p.formula, layout, inter), p))
from graphqembed_amd.tensorize import table_key
feeder = eng2.make_feeder(plist, {table_key(m): all_rows[m] for m in g.modes}, batch_size=B, seed=0)
eng2.feeder_run(feeder, 0, 50)
torch.cuda.synchronize()
t0 = time.perf_counter()
losses = eng2.feeder_run(feeder, 50, 500)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("native feeder (gqe_feeder_run): 500 iterations, %.1f us/iteration, %.2f M queries/s, final loss %.5f"
      % (dt / 500 * 1e6, 500 * 9 * B / dt / 1e6, float(losses[9].item())))
