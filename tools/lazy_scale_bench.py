#!/usr/bin/env python3
"""Eager vs lazy (deferred, bit-exact) Adam as the tables grow: the eager pass streams every row of every stepped
table each iteration, the lazy mode only the rows an iteration touches — the bio-synth graph at 1x / 4x / 16x its
node counts, same 9 x 512 queries per step.  python tools/lazy_scale_bench.py [factors ...]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import build_layout, init_params
from graphqembed_amd import data_utils, synth
from graphqembed_amd.engine import Engine
from graphqembed_amd.tensorize import FormulaPlan, pack_margin_batches

d, dec, inter, B = 128, "bilinear-diag", "min", 512
factors = [int(x) for x in sys.argv[1:]] or [1, 4, 16]
for fac in factors:
    sizes = {m: n * fac for m, n in data_utils.BIO_SYNTH_SIZES.items()}
    g = synth.bio_synth(seed=0, sizes=sizes, edges_per_kind=data_utils.BIO_SYNTH_EDGES_PER_KIND * fac)
    layout = build_layout(g, d, dec, inter)
    mix = synth.FULL_MIX
    pools = synth.make_pools(g, sorted(set(m[0] for m in mix)), formulas_per_type=6, pool_size=8192, seed=0)
    res = {}
    for lazy in (False, True):
        eng = Engine(d, dec, inter, layout, max_queries=B * len(mix), max_batches=len(mix), lazy_adam=lazy)
        init_params(eng, d, 0)
        plans, prepared = {}, []
        for s in range(16):
            packed = []
            for (f, t, ng, a, w, m) in synth.mix_iteration(pools, mix, s, B):
                if f not in plans:
                    plans[f] = FormulaPlan(f, layout, inter)
                packed.append((plans[f], t, ng, a, w, m))
            descs, idx, _ = pack_margin_batches(packed)
            ps = eng.prepare_margin(descs, torch.from_numpy(idx).to(eng.device))
            ps["adam"] = eng.prepare_adam(set().union(*[p[0].touched for p in packed]))
            prepared.append(ps)
        for i in range(40):
            eng.run_margin(prepared[i % 16]); eng.run_adam(prepared[i % 16]["adam"])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 200
        for i in range(n):
            eng.run_margin(prepared[i % 16]); eng.run_adam(prepared[i % 16]["adam"])
        eng.sync()
        torch.cuda.synchronize()
        res[lazy] = (time.perf_counter() - t0) / n * 1e6
        eng.close()
        del eng
        torch.cuda.empty_cache()
    print("tables x%-3d  P = %6.1f M params   eager %7.1f us/step   lazy %7.1f us/step   (%.2fx)"
          % (fac, layout.total / 1e6, res[False], res[True], res[False] / res[True]), flush=True)
