#!/usr/bin/env python3
"""What running the optimiser's streaming pass NEXT TO the fused + pair-GEMM launches would buy: engine 1 runs its margin
launches on stream A, engine 2 (same workload, its own arenas, empty gradient lists: a pure p/m/v stream) runs its Adam
pass on stream B; the two are tied together by the events a split step would need (A(t) waits for B(t-1), B(t) for A(t-1)).
python tools/probes/overlap_probe.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from graphqembed_amd import synth

wl = bench.Workload("bio-synth", 128, "bilinear-diag", "min", synth.FULL_MIX, 512)
e1, e2 = wl.engine(), wl.engine()
p1, p2 = wl.prepare(e1), wl.prepare(e2)
import ctypes as C
hip = C.CDLL("libamdhip64.so")


def masked_stream(lo, hi):
    """a stream whose kernels run on CUs [lo, hi) of the mask order only (hipExtStreamCreateWithCUMask; on a multi-XCD part the
    driver deals the mask bits round-robin over the XCDs, so 64 low bits = 8 CUs of every XCD)"""
    words = (C.c_uint32 * 8)()
    for i in range(lo, hi):
        words[i // 32] |= 1 << (i % 32)
    st = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value)


A, B = torch.cuda.Stream(), torch.cuda.Stream()
SPLIT = int(os.environ.get("SPLIT", "0"))      # CUs given to the Adam stream
if SPLIT:
    B = masked_stream(0, SPLIT)
    if os.environ.get("BOTH"):
        A = masked_stream(SPLIT, 256)
n = wl.n_distinct


def reset(ps):
    e1._check(e1.lib.gqe_zero_grads(e1.ctx, ps["adam"]["arr"], ps["adam"]["n"], e1._stream()))


def run(mode, steps):
    evA = [torch.cuda.Event() for _ in range(2)]
    evB = [torch.cuda.Event() for _ in range(2)]
    for i in range(steps):
        ps1, ps2 = p1[i % n], p2[i % n]
        if mode == "serial":                      # everything on A: margin of engine 1, then the Adam stream of engine 2
            with torch.cuda.stream(A):
                e1.run_margin(ps1)
                reset(ps1)                        # (keeps engine 1's lists empty: a pass over the list heads, counted in every mode)
                e2.run_adam(ps2["adam"])
        elif mode == "margin":
            with torch.cuda.stream(A):
                e1.run_margin(ps1)
                reset(ps1)
        elif mode == "adam":
            with torch.cuda.stream(B):
                e2.run_adam(ps2["adam"])
        elif mode == "free":                      # both streams, no dependency at all
            with torch.cuda.stream(A):
                e1.run_margin(ps1)
                reset(ps1)
            with torch.cuda.stream(B):
                e2.run_adam(ps2["adam"])
        else:
            with torch.cuda.stream(A):
                if i > 0:
                    A.wait_event(evB[(i - 1) & 1])
                e1.run_margin(ps1)
                reset(ps1)
                evA[i & 1].record(A)
            with torch.cuda.stream(B):
                if i > 0:
                    B.wait_event(evA[(i - 1) & 1])
                e2.run_adam(ps2["adam"])
                evB[i & 1].record(B)


for mode in ("margin", "adam", "serial", "overlap", "free"):
    run(mode, 50)
    torch.cuda.synchronize()
    ts = []
    for rep in range(10):
        t0 = time.perf_counter()
        run(mode, 100)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 100)
    line = "%-12s %7.1f us/step (median of 10 x 100)" % (mode, np.median(ts) * 1e6)
    if True:     # kernel times between the library's events, alone and side by side
        e1.grads.zero_(); torch.cuda.synchronize()
        e1.timing_enable(1); e2.timing_enable(1)
        run(mode, 100)
        torch.cuda.synchronize()
        f, _ = e1.timing_read(0); g, _ = e1.timing_read(1); o, _ = e2.timing_read(2)
        e1.timing_enable(0); e2.timing_enable(0)
        line += "   fused %.1f  gemm %.1f  opt %.1f us" % (f * 1e3, g * 1e3, o * 1e3)
    print(line, flush=True)
