#!/usr/bin/env python3
"""Evaluation micro-benchmark: score Q queries of one formula against K stored negatives each
(eval_perc_queries' inner loop, utils.py:70-91) — expanded (one forward query per candidate, as the reference
does) vs the fused candidate-list launch.
    python tools/eval_bench.py                                   bio-synth, d=128, bilinear-diag + SetIntersection(min): the 50 MB
                                                                 target tables sit in the Infinity Cache (a cache-bandwidth figure)
    python tools/eval_bench.py --workload reddit-synth           d=256, candidates drawn from the 500 k-user table (512 MB, far
                                                                 beyond the 256 MB Infinity Cache): the HBM figure
    python tools/eval_bench.py --decoder bilinear                full-Bilinear chains project the CANDIDATE: tiles of 16
                                                                 candidates, [16 x d].[d x d] per hop on the matrix cores"""
import argparse
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import build_layout, init_params
from graphqembed_amd import synth
from graphqembed_amd.engine import Engine
from graphqembed_amd.tensorize import FormulaPlan, pack_candidate_batches, pack_forward_batches

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="bio-synth", choices=["bio-synth", "reddit-synth"])
ap.add_argument("--decoder", default="bilinear-diag")
ap.add_argument("--inter", default="min")
args = ap.parse_args()
reddit = args.workload == "reddit-synth"
d, dec, inter = (256 if reddit else 128), args.decoder, args.inter
g = synth.reddit_synth(seed=0) if reddit else synth.bio_synth(seed=0)
layout = build_layout(g, d, dec, inter)
bags = {}
if reddit:
    from graphqembed_amd.tensorize import table_key
    bags = {table_key(m): csr for m, csr in g.bags.items()}
eng = Engine(d, dec, inter, layout, max_queries=4096, max_batches=4, bags=bags)
init_params(eng, d, 0)
types = ["1-chain", "2-chain", "2-inter", "3-inter"] if dec == "bilinear" else ["2-chain", "2-inter", "3-inter"]
pools = synth.make_pools(g, types, formulas_per_type=6 if reddit else 1, pool_size=1024, seed=0)
rng = np.random.RandomState(0)
Q, K = 1000, 1000
print("%s d=%d %s + %s; target tables: %s" % (args.workload, d, dec, inter, ", ".join("%s %.0f MB" % (m, g.table_rows[m] * d * 4 / 1e6) for m in g.modes)))
for qt in types:
    cand = [p for p in pools[qt] if not (reddit and p.formula.target_mode != "user")]    # reddit-synth: score against the big user table
    if not cand:
        continue
    p = cand[0]
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
    hops = {"1-chain": 1, "2-chain": 2, "3-chain": 3}.get(qt, 0)
    mfma = ", %.1f TF/s fp32 MFMA" % (2.0 * d * d * pairs * hops / t_f / 1e12) if dec == "bilinear" and hops else ""
    print("%-8s (target %s) %d queries x %d candidates: fused %.3f ms (%.0f M pairs/s, %.1f GB/s of candidate rows%s) | expanded %.3f ms | x%.1f | max|diff| %.1e"
          % (qt, p.formula.target_mode, Q, K + 1, t_f * 1e3, pairs / t_f / 1e6, pairs * d * 4 / t_f / 1e9, mfma, t_e * 1e3, t_e / t_f, err))
