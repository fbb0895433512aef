#!/usr/bin/env python3
"""The fused optimiser pass against the size of what it streams: bio-synth with 1/4 .. 4x the nodes, same 9 x 512 queries
per step.  Prints the pass's hipEvent time and the p + m + v bytes it moves per second — does a footprint that fits the
32 MB of L2 / the 256 MB Infinity Cache stream faster than HBM?  python tools/opt_scale_bench.py [factors ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import build_layout, init_params
from graphqembed_amd import data_utils, synth
from graphqembed_amd.engine import Engine
from graphqembed_amd.tensorize import FormulaPlan, pack_margin_batches

d, dec, inter, B = 128, "bilinear-diag", "min", 512
factors = [float(x) for x in sys.argv[1:]] or [0.125, 0.25, 0.5, 1, 2, 4]
for fac in factors:
    sizes = {m: max(int(n * fac), 600) for m, n in data_utils.BIO_SYNTH_SIZES.items()}
    g = synth.bio_synth(seed=0, sizes=sizes, edges_per_kind=max(int(data_utils.BIO_SYNTH_EDGES_PER_KIND * fac), 8000))
    layout = build_layout(g, d, dec, inter)
    mix = synth.FULL_MIX
    pools = synth.make_pools(g, sorted(set(m[0] for m in mix)), formulas_per_type=6, pool_size=8192, seed=0)
    eng = Engine(d, dec, inter, layout, max_queries=B * len(mix), max_batches=len(mix))
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
    eng.timing_enable(4)
    for k in range(3):
        eng.timing_read(k)
    for i in range(400):
        eng.run_margin(prepared[i % 16]); eng.run_adam(prepared[i % 16]["adam"])
    torch.cuda.synchronize()
    ms_opt, n_opt = eng.timing_read(2)
    ms_fused, _ = eng.timing_read(0)
    eng.timing_enable(0)
    table_params = sum(n for n in sizes.values()) * d
    byts = 24.0 * table_params
    print("tables x%-5g P = %6.1f M   p+m+v in/out %7.1f MB   K_opt %7.1f us  -> %5.2f TB/s   (fused %5.1f us)"
          % (fac, layout.total / 1e6, byts / 1e6, ms_opt * 1e3, byts / (ms_opt * 1e-3) / 1e12, ms_fused * 1e3), flush=True)
    eng.close()
    del eng
    torch.cuda.empty_cache()
