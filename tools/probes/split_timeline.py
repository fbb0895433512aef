#!/usr/bin/env python3
"""Timeline of ONE split step's fused launch (GQE_SPLIT_PROF=1): when tiles and rider workgroups start / end, where they ran.
GQE_SPLIT_PROF=1 [GQE_SPLIT_SHAPE=8|16] [GQE_SPLIT_ROUNDS=n] python tools/probes/split_timeline.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("GQE_SPLIT_PROF", "1")
import torch
import bench
from graphqembed_amd import synth
wl = bench.Workload("bio-synth", 128, os.environ.get("TIMELINE_DECODER", "bilinear-diag"), os.environ.get("TIMELINE_INTER", "min"), synth.FULL_MIX, 512, n_distinct=4)
eng = wl.engine()
prep = wl.prepare(eng)
for i in range(6):
    eng.run_train_step(prep[i % 4], prep[i % 4]["adam"])
torch.cuda.synchronize()
tiles = 9 * 32
cap = tiles + 16384
stamps = torch.zeros(cap * 64, dtype=torch.int64, device="cuda")
for k in range(8):     # launch spans of several steps (the detailed timeline below is the last one's)
    stamps.zero_()
    torch.cuda.synchronize()
    steady = os.environ.get("TIMELINE_STEADY", "1") != "0"
    if steady:          # the profiled step sits inside a run of back-to-back steps (no synchronisation around it)
        for i in range(40):
            eng.run_train_step(prep[i % 4], prep[i % 4]["adam"])
    eng._check(eng.lib.gqe_debug_profile(eng.ctx, stamps.data_ptr()))
    eng.run_train_step(prep[k % 4], prep[k % 4]["adam"])
    eng._check(eng.lib.gqe_debug_profile(eng.ctx, None))
    if steady:
        for i in range(10):
            eng.run_train_step(prep[i % 4], prep[i % 4]["adam"])
    torch.cuda.synchronize()
    st = stamps.cpu().numpy().reshape(cap, 64)
    u = np.flatnonzero(st[:, 0] != 0)
    tl, rd = u[u < tiles], u[u >= tiles]
    z = st[u, 0].min()
    print("step %d: tiles end %.1f  riders end %.1f  (launch span %.1f us)" % (k, (st[tl, 8].max() - z) / 100.0, ((st[rd, 8].max() - z) / 100.0) if len(rd) else 0.0,
                                                                              (st[u, 8].max() - z) / 100.0))
used = np.flatnonzero(st[:, 0] != 0)
t0 = st[used, 0].min()
us = lambda x: (x - t0) / 100.0
T, R = used[used < tiles], used[used >= tiles]
print("split steps", eng.split_steps(), "tiles", len(T), "rider workgroups", len(R))
print("tiles : start min/med/max %.1f %.1f %.1f   end min/med/max %.1f %.1f %.1f" % (
    us(st[T, 0]).min(), np.median(us(st[T, 0])), us(st[T, 0]).max(), us(st[T, 8]).min(), np.median(us(st[T, 8])), us(st[T, 8]).max()))
for b in range(9):
    tt = T[(T >= 32 * b) & (T < 32 * (b + 1))]
    print("   tile block %d: start med %.1f  end med %.1f  dur med %.1f max %.1f" % (b, np.median(us(st[tt, 0])), np.median(us(st[tt, 8])),
          np.median((st[tt, 8] - st[tt, 0]) / 100.0), ((st[tt, 8] - st[tt, 0]) / 100.0).max()))
if len(R):
    rs, re = us(st[R, 0]), us(st[R, 8])
    print("riders: start min/med/max %.1f %.1f %.1f   end min/med/max %.1f %.1f %.1f   duration med %.1f p90 %.1f max %.1f" % (
        rs.min(), np.median(rs), rs.max(), re.min(), np.median(re), re.max(), np.median(re - rs), np.percentile(re - rs, 90), (re - rs).max()))
    edges = np.arange(0, re.max() + 5, 5.0)
    print("  t(us)  riders running  riders finished  tiles running")
    for e in edges:
        print("  %5.0f  %6d  %6d  %6d" % (e, int(((rs <= e) & (re > e)).sum()), int((re <= e).sum()), int(((us(st[T, 0]) <= e) & (us(st[T, 8]) > e)).sum())))
    cu = st[R, 1]
    print("distinct (xcc, hw_id>>8 & 0xf cu, se) of riders:", len(set((int(c >> 32), int(c & 0xffffffff) >> 8 & 0xf, int(c & 0xffffffff) >> 13 & 0x7) for c in cu)))

if len(R) and st[T, 63].any():   # (debug build with GQE_LEAN_PROF: where every tile ran)
    key = lambda c: (int(c >> 32), int(c & 0xffffffff) >> 8 & 0xf, int(c & 0xffffffff) >> 13 & 0x7)
    lead = R[us(st[R, 0]) < 5.0]
    lead_cus = {}
    for r in lead:
        lead_cus[key(st[r, 1])] = us(st[r, 8])
    late = T[us(st[T, 0]) > 5.0]
    on_lead = [t for t in late if key(st[t, 63]) in lead_cus]
    print("second-round tiles: %d, of which on a lead rider's CU: %d" % (len(late), len(on_lead)))
    if on_lead:
        gap = [us(st[t, 0]) - lead_cus[key(st[t, 63])] for t in on_lead]
        print("   start of those tiles: min/med/max %.1f %.1f %.1f; gap after the rider's end stamp: min/med/max %.1f %.1f %.1f" % (
            min(us(st[on_lead, 0])), np.median(us(st[on_lead, 0])), max(us(st[on_lead, 0])), min(gap), np.median(gap), max(gap)))
    off = [t for t in late if key(st[t, 63]) not in lead_cus]
    if off:
        print("   the others start: min/med/max %.1f %.1f %.1f" % (min(us(st[off, 0])), np.median(us(st[off, 0])), max(us(st[off, 0]))))
    print("   lead riders end: min/med/max %.1f %.1f %.1f" % (min(lead_cus.values()), np.median(list(lead_cus.values())), max(lead_cus.values())))
