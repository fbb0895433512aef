#!/usr/bin/env python3
"""Is a captured HIP graph of the iteration faster than the three eager launches?  (SURVEY.md §8f-2 lists "HIP-graph
whole iteration".)  One pre-sampled bio-synth full-mix iteration — fused forward/backward, pair GEMM, fused Adam — is
captured with torch.cuda.CUDAGraph (hipStreamBeginCapture under it; the library's launches go to the capturing stream) and
replayed; against the same iteration launched eagerly.  The replay re-applies the captured kernel arguments — the Adam bias
corrections of ONE step count and ONE index feed — so it is a timing probe, not a training loop: a real graph would
need its kernel nodes' parameters updated every iteration, which is what the kernel-argument plan already does at launch.
    python tools/graph_probe.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from graphqembed_amd import synth

wl = bench.Workload("bio-synth", 128, "bilinear-diag", "min", synth.FULL_MIX, 512, n_distinct=1)
eng = wl.engine()
ps = wl.prepare(eng)[0]


def eager():
    eng.run_margin(ps)
    eng.run_adam(ps["adam"])


def timed(fn, n=2000):
    for _ in range(100):
        fn()
    torch.cuda.synchronize()
    best = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        best.append((time.perf_counter() - t0) / n)
    return np.median(best) * 1e6


t_eager = timed(eager)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        eager()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(graph):
        eager()
    t_graph = timed(graph.replay)
    print("one full-mix iteration (fused + pair GEMM + Adam), 1 x MI355X: eager launches %.1f us, hipGraph replay %.1f us" % (t_eager, t_graph))
except Exception as e:
    print("one full-mix iteration: eager launches %.1f us; graph capture failed: %s" % (t_eager, repr(e)[:200]))
eng.close()
