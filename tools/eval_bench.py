#!/usr/bin/env python3
"""Evaluation micro-benchmark: score Q queries of one formula against K stored negatives each
(eval_perc_queries' inner loop, utils.py:70-91) — expanded (one forward query per candidate, as the reference
does) vs the fused candidate-list launch.  bio-synth, d=128, bilinear-diag + SetIntersection(min)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import build_layout, init_params
from graphqembed_amd import synth
from graphqembed_amd.engine import Engine
from graphqembed_amd.tensorize import FormulaPlan, pack_candidate_batches, pack_forward_batches

d, dec, inter = 128, "bilinear-diag", "min"
g = synth.bio_synth(seed=0)
layout = build_layout(g, d, dec, inter)
eng = Engine(d, dec, inter, layout, max_queries=4096, max_batches=4)
init_params(eng, d, 0)
pools = synth.make_pools(g, ["2-chain", "2-inter", "3-inter"], formulas_per_type=1, pool_size=1024, seed=0)
rng = np.random.RandomState(0)
Q, K = 1000, 1000
for qt in ("2-chain", "2-inter", "3-inter"):
    p = pools[qt][0]
    plan = FormulaPlan(p.formula, layout, inter)
    nt = g.mode_sizes[p.formula.target_mode]
    ptr = (np.arange(Q + 1) * (K + 1)).astype(np.int32)
    rows = rng.randint(1, nt + 1, size=Q * (K + 1)).astype(np.int32)
    rows[ptr[:-1]] = p.target[:Q]
    anchors = p.anchors[:, :Q]
    # fused
    descs, idx, n = pack_candidate_batches([(plan, anchors, ptr, rows)])
    didx = torch.from_numpy(idx).cuda(); out = torch.empty(n, device="cuda")
    eng.forward(descs, didx, n, out=out); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): eng.forward(descs, didx, n, out=out)
    torch.cuda.synchronize(); t_f = (time.perf_counter() - t0) / 5
    fused = out.cpu().numpy()
    # expanded: chunks of 64 queries x (K+1) candidates as ordinary forward batches
    t_e, got = 0.0, []
    for c0 in range(0, Q, 64):
        c1 = min(Q, c0 + 64)
        rep = np.repeat(np.arange(c0, c1), K + 1)
        descs, idx, n2 = pack_forward_batches([(plan, rows[ptr[c0]:ptr[c1]], anchors[:, rep])])
        didx2 = torch.from_numpy(idx).cuda(); out2 = torch.empty(n2, device="cuda")
        eng.forward(descs, didx2, n2, out=out2); torch.cuda.synchronize()
        t0 = time.perf_counter(); eng.forward(descs, didx2, n2, out=out2); torch.cuda.synchronize()
        t_e += time.perf_counter() - t0
        got.append(out2.cpu().numpy())
    err = np.abs(np.concatenate(got) - fused).max()
    pairs = Q * (K + 1)
    print("%-8s %d queries x %d candidates: fused %.3f ms (%.0f M pairs/s, %.1f GB/s of candidate rows) | expanded %.3f ms | x%.1f | max|diff| %.1e"
          % (qt, Q, K + 1, t_f * 1e3, pairs / t_f / 1e6, pairs * d * 4 / t_f / 1e9, t_e * 1e3, t_e / t_f, err))
