#!/usr/bin/env python3
"""How the optimiser pass reacts to hub rows: a fraction of every batch's targets is replaced by one node, so
that node's gradient list grows to hundreds of entries.  python tools/hub_bench.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import build_layout, init_params
from graphqembed_amd import synth
from graphqembed_amd.engine import Engine
from graphqembed_amd.tensorize import FormulaPlan, pack_margin_batches

d, dec, inter, B = 128, "bilinear-diag", "min", 512
g = synth.bio_synth(seed=0)
layout = build_layout(g, d, dec, inter)
mix = synth.FULL_MIX
eng = Engine(d, dec, inter, layout, max_queries=B * len(mix), max_batches=len(mix))
init_params(eng, d, 0)
pools = synth.make_pools(g, sorted(set(m[0] for m in mix)), formulas_per_type=6, pool_size=8192, seed=0)
plans = {}
for frac in (0.0, 0.02, 0.1, 0.5):
    prepared = []
    for s in range(8):
        packed = []
        for (f, t, ng, a, w, m) in synth.mix_iteration(pools, mix, s, B):
            if f not in plans:
                plans[f] = FormulaPlan(f, layout, inter)
            t = t.copy()
            k = int(frac * len(t))
            if k:
                t[:k] = t[0]                       # one hub node per batch
            packed.append((plans[f], t, ng, a, w, m))
        descs, idx, _ = pack_margin_batches(packed)
        ps = eng.prepare_margin(descs, torch.from_numpy(idx).to(eng.device))
        ps["adam"] = eng.prepare_adam(set().union(*[p[0].touched for p in packed]))
        prepared.append(ps)
    for i in range(10):
        eng.run_margin(prepared[i % 8]); eng.run_adam(prepared[i % 8]["adam"])
    torch.cuda.synchronize()
    eng.timing_enable(1)
    for i in range(40):
        eng.run_margin(prepared[i % 8]); eng.run_adam(prepared[i % 8]["adam"])
    torch.cuda.synchronize()
    f_ms, _ = eng.timing_read(0); g_ms, _ = eng.timing_read(1); o_ms, _ = eng.timing_read(2)
    eng.timing_enable(0)
    print("hub fraction %.2f (list length ~%d per hub row): fused %.1f us, pair-gemm %.1f us, optimiser %.1f us"
          % (frac, int(frac * B), f_ms * 1e3, g_ms * 1e3, o_ms * 1e3), flush=True)
