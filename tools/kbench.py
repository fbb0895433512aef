#!/usr/bin/env python3
"""Kernel micro-benchmark: per-launch time of the fused kernel for different batch compositions
(bio-synth, d=128 by default).  python tools/kbench.py [--dim 128] [--decoder bilinear-diag] [--inter min]"""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import build_layout, init_params
from graphqembed_amd import synth
from graphqembed_amd.engine import Engine
from graphqembed_amd.tensorize import FormulaPlan, pack_margin_batches, pack_forward_batches

ap = argparse.ArgumentParser()
ap.add_argument("--dim", type=int, default=128)
ap.add_argument("--decoder", default="bilinear-diag")
ap.add_argument("--inter", default="min")
ap.add_argument("--reps", type=int, default=50)
ap.add_argument("--workload", default="bio-synth", help="--durations: bio-synth | reddit-synth")
ap.add_argument("--durations", type=int, default=0, help="only the per-tile phase DURATIONS of the full mix at this batch size (many tiles per CU)")
args = ap.parse_args()
d = args.dim
if args.durations and args.workload != "bio-synth":
    from bench import Workload
    wl = Workload(args.workload, d, args.decoder, args.inter, synth.FULL_MIX, args.durations, n_distinct=2)
    weng = wl.engine()
    wps = wl.prepare(weng)[0]
g = synth.bio_synth(seed=0)
layout = build_layout(g, d, args.decoder, args.inter)
eng = Engine(d, args.decoder, args.inter, layout, max_queries=9 * 4096, max_batches=16)
init_params(eng, d, 0)
types = ["1-chain", "2-chain", "3-chain", "2-inter", "3-inter", "3-inter_chain", "3-chain_inter"]
pools = synth.make_pools(g, types, formulas_per_type=4, pool_size=16384, seed=0)
plans = {}
def items_for(mix, B, step=0):
    out = []
    for (f, t, ng, a, w, m) in synth.mix_iteration(pools, mix, step, B):
        if f not in plans: plans[f] = FormulaPlan(f, layout, args.inter)
        out.append((plans[f], t, ng, a, w, m))
    return out
def time_margin(mix, B, label):
    sets = []
    for s in range(4):
        descs, idx, _ = pack_margin_batches(items_for(mix, B, s))
        sets.append(eng.prepare_margin(descs, torch.from_numpy(idx).cuda()))
    for ps in sets: eng.run_margin(ps); eng.materialize()
    torch.cuda.synchronize(); eng.timing_enable(1)
    for r in range(args.reps): eng.run_margin(sets[r % 4]); eng.materialize()
    torch.cuda.synchronize()
    f, nf = eng.timing_read(0); gm, ng_ = eng.timing_read(1); eng.timing_enable(0)
    tiles = sum((B + 15) // 16 for _ in mix)
    print("%-38s B=%5d tiles=%5d fused %8.2f us  pair_gemm %7.2f us" % (label, B, tiles, f * 1e3, gm * 1e3), flush=True)
    eng.grads.zero_()
def time_forward(mix, B, label):
    its = items_for(mix, B)
    descs, idx, n = pack_forward_batches([(p, t, a) for (p, t, ng, a, w, m) in its])
    didx = torch.from_numpy(idx).cuda(); out = torch.empty(n, device="cuda")
    for _ in range(3): eng.forward(descs, didx, n, out=out)
    torch.cuda.synchronize(); eng.timing_enable(1)
    for r in range(args.reps): eng.forward(descs, didx, n, out=out)
    torch.cuda.synchronize(); f, nf = eng.timing_read(0); eng.timing_enable(0)
    print("%-38s B=%5d forward-only %8.2f us" % (label, B, f * 1e3), flush=True)
def phase_durations(mix, B, eng=eng, ps=None):
    """Per tile: time between consecutive phase stamps (median / 90th percentile over the tiles of a batch type)."""
    import ctypes as C
    if ps is None:
        its = items_for(mix, B)
        descs, idx, _ = pack_margin_batches(its)
        ps = eng.prepare_margin(descs, torch.from_numpy(idx).cuda())
    tiles = sum((B + 15) // 16 for _ in mix)
    stamps = torch.zeros((tiles + 16384) * 64, dtype=torch.int64, device="cuda")
    eng.run_margin(ps); eng.materialize(); torch.cuda.synchronize()
    eng._check(eng.lib.gqe_debug_profile(eng.ctx, stamps.data_ptr()))
    eng.run_margin(ps); eng.materialize(); torch.cuda.synchronize()
    eng._check(eng.lib.gqe_debug_profile(eng.ctx, None))
    st = stamps.cpu().numpy().reshape(tiles + 16384, 64).astype(np.float64)[:tiles]
    print("== per-tile phase durations, full mix B=%d (%d tiles); launch span %.1f us" % (B, tiles, (st[:, 8].max() - st[:, 0].min()) / 100.0))
    seq = [(0, "start"), (1, "idx"), (9, "rows+norm"), (2, "init"), (10, "vec+bar"), (11, "pre_mfma"), (3, "pre_bar"), (4, "post/final"),
           (5, "score"), (6, "postT"), (12, "gz+bar"), (13, "preT_mfma"), (14, "bar"), (7, "hops+scatter"), (8, "commit")]
    n_anchor = {"1-chain": 1, "2-chain": 1, "3-chain": 1, "2-inter": 2, "3-inter": 3, "3-inter_chain": 2, "3-chain_inter": 2}
    hops = {"1-chain": 1, "2-chain": 2, "3-chain": 3, "2-inter": 2, "3-inter": 3, "3-inter_chain": 3, "3-chain_inter": 3}
    mlp = not args.inter.endswith("simple")
    bil = args.decoder == "bilinear"
    def cost(qt):
        na, chain = n_anchor[qt], qt.endswith("chain") and "inter" not in qt
        extra = (4 * hops[qt] if bil else 0) if chain else ((6 + 2 * na) if mlp else 2) + (2 * hops[qt] if bil else 0)
        return 2 + na + extra
    order = sorted(range(len(mix)), key=lambda k: -cost(mix[k][0]))
    off = 0
    for (qt, w, hard) in [mix[k] for k in order]:
        nt = (B + 15) // 16
        blk = st[off:off + nt]; off += nt
        cells, prev = [], 0
        for (k, nm) in seq[1:]:
            ok = (blk[:, k] > 0) & (blk[:, prev] > 0)
            if not ok.any():
                continue
            dur = (blk[ok, k] - blk[ok, prev]) / 100.0
            cells.append("%s %.1f/%.1f" % (nm, np.median(dur), np.percentile(dur, 90)))
            prev = k
        tot = (blk[:, 8] - blk[:, 0]) / 100.0
        print("%-15s tile %.1f/%.1f us | %s" % (qt + ("*" if hard else ""), np.median(tot), np.percentile(tot, 90), "  ".join(cells)))
if args.durations and args.workload != "bio-synth":
    phase_durations(list(synth.FULL_MIX), args.durations, weng, wps)
    sys.exit(0)
if args.durations:
    time_margin(list(synth.FULL_MIX), args.durations, "full mix")
    phase_durations(list(synth.FULL_MIX), args.durations)
    sys.exit(0)
for qt in types:
    hard = "inter" in qt
    time_margin([(qt, 0.01, hard)] * 9, 512, "9x " + qt)
for qt in ["1-chain", "2-inter", "3-inter"]:
    time_margin([(qt, 0.01, False)] * 1, 16, "1 tile " + qt)
    time_margin([(qt, 0.01, False)] * 9, 64, "36 tiles " + qt)
    time_margin([(qt, 0.01, False)] * 8, 512, "256 tiles " + qt)
time_margin(list(synth.FULL_MIX), 512, "full mix")
time_margin(list(synth.FULL_MIX), 4096, "full mix")
time_forward(list(synth.FULL_MIX), 512, "full mix")

# ---- phase profile of the full mix (wall_clock64 stamps, 100 MHz) ----
import ctypes as C
def phase_profile(mix, B, label):
    its = items_for(mix, B)
    descs, idx, _ = pack_margin_batches(its)
    ps = eng.prepare_margin(descs, torch.from_numpy(idx).cuda())
    tiles = sum((B + 15) // 16 for _ in mix)
    GEMM_WGS = 2048   # the pair-GEMM launch stamps behind the tiles (one row per workgroup; unused rows stay 0)
    stamps = torch.zeros((tiles + GEMM_WGS) * 64, dtype=torch.int64, device="cuda")
    eng.run_margin(ps); eng.materialize(); torch.cuda.synchronize()
    eng._check(eng.lib.gqe_debug_profile(eng.ctx, stamps.data_ptr()))
    eng.run_margin(ps); eng.materialize(); torch.cuda.synchronize()
    eng._check(eng.lib.gqe_debug_profile(eng.ctx, None))
    allst = stamps.cpu().numpy().reshape(tiles + GEMM_WGS, 64).astype(np.float64)
    st = allst[:tiles]
    t0 = st[:, 0].min()
    names = ["start", "idx", "rows", "branches", "post/final", "score", "postT", "branches_bwd", "loss"]
    print("== phase profile %s B=%d (us since first block start; per batch type: median over its tiles)" % (label, B))
    # tiles are laid out longest batch first (gqe_host.cpp, run_queries): intersections by branch count, chains last
    n_anchor = {"1-chain": 1, "2-chain": 1, "3-chain": 1, "2-inter": 2, "3-inter": 3, "3-inter_chain": 2, "3-chain_inter": 2}
    mlp = not args.inter.endswith("simple")
    hops = {"1-chain": 1, "2-chain": 2, "3-chain": 3, "2-inter": 2, "3-inter": 3, "3-inter_chain": 3, "3-chain_inter": 3}
    bil = args.decoder == "bilinear"
    def cost(qt):
        na, chain = n_anchor[qt], qt.endswith("chain") and "inter" not in qt
        extra = (4 * hops[qt] if bil else 0) if chain else ((6 + 2 * na) if mlp else 2) + (2 * hops[qt] if bil else 0)
        return 2 + na + extra
    order = sorted(range(len(mix)), key=lambda k: -cost(mix[k][0]))
    off = 0
    for (qt, w, hard) in [mix[k] for k in order]:
        nt = (B + 15) // 16
        blk = st[off:off + nt]; off += nt
        rel = (blk - t0) / 100.0
        cells = []
        for k in range(9):
            col = rel[:, k][blk[:, k] > 0]
            cells.append("%s=%6.1f" % (names[k], np.median(col)) if len(col) else "%s=   n/a" % names[k])
        print("%-14s %s  end(max)=%6.1f" % (qt + ("*" if hard else ""), " ".join(cells), rel[:, 8].max()))
        extra = []
        for k, nm in ((9, "norm"), (10, "vec+bar"), (11, "pre_mfma"), (12, "gz+bar"), (13, "preT_mfma"), (14, "bar")):
            col = rel[:, k][blk[:, k] > 0]
            if len(col):
                extra.append("%s=%6.1f" % (nm, np.median(col)))
        if extra:
            print("%-14s   detail: %s" % ("", " ".join(extra)))
        wn = ["score_end", "gq_parked", "postT_go", "postT_done", "bar", "commit", "preT_go", "preT_done", "hops_done", "(unused)", "vg+links"]
        for pidx, nm in enumerate(wn):
            vals = []
            for k in range(4):
                col = rel[:, 16 + pidx * 4 + k][blk[:, 16 + pidx * 4 + k] > 0]
                vals.append("%6.1f" % np.median(col) if len(col) else "   n/a")
            if any(v.strip() != "n/a" for v in vals) and qt == "3-inter" and not hard:
                print("%-14s   waves 0/4/8/12 %-10s %s" % ("", nm, " ".join(vals)))
    # ---- the pair-GEMM launch ----
    gs = allst[tiles:]
    gs = gs[gs[:, 0] > 0]
    if len(gs):
        g0 = gs[:, 0].min()
        fused_end = st[:, 8].max()
        names_g = ["start", "decoded", "panels_in", "lds0", "mfma0", "lds1", "mfma1", "atomics_issued"]
        units = gs[gs[:, 7] > 0]
        print("== pair GEMM: %d workgroups stamped (%d units); first start %.1f us after the last fused tile ended" % (len(gs), len(units), (g0 - fused_end) / 100.0))
        print("   median over units, us since the launch's first start: " + " ".join("%s=%.1f" % (names_g[k], np.median((units[:, k] - g0) / 100.0)) for k in range(8)))
        print("   max   over units:                                     " + " ".join("%s=%.1f" % (names_g[k], np.max((units[:, k] - g0) / 100.0)) for k in range(8)))
        fin = gs[gs[:, 7] == 0]
        if len(fin): print("   finalize block start=%.1f" % ((fin[0, 0] - g0) / 100.0))
phase_profile(list(synth.FULL_MIX), 512, "full mix")
