#!/usr/bin/env python3
"""Latency of a cross-stream dependency: a tiny kernel ping-pongs between two streams through events (record on one, wait on
the other), against the same kernels back to back on one stream.   python tools/probes/event_hop_probe.py"""
import time
import torch
x = torch.zeros(1024, device="cuda")
A, B = torch.cuda.Stream(), torch.cuda.Stream()
N = 2000
def pingpong():
    evs = [torch.cuda.Event() for _ in range(4)]
    for i in range(N):
        with torch.cuda.stream(A):
            if i: A.wait_event(evs[(2 * i - 1) & 3])
            x.add_(1.0); evs[(2 * i) & 3].record(A)
        with torch.cuda.stream(B):
            B.wait_event(evs[(2 * i) & 3])
            x.add_(1.0); evs[(2 * i + 1) & 3].record(B)
def serial():
    with torch.cuda.stream(A):
        for i in range(2 * N): x.add_(1.0)
def signaled_wait():      # B is far ahead: every event A waits for has long fired
    ev = torch.cuda.Event()
    with torch.cuda.stream(B):
        x.add_(1.0); ev.record(B)
    torch.cuda.synchronize()
    with torch.cuda.stream(A):
        for i in range(2 * N):
            A.wait_event(ev); x.add_(1.0)
for name, fn in (("serial", serial), ("ping-pong", pingpong), ("wait on a fired event", signaled_wait)):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
    print("%-24s %6.2f us per kernel" % (name, (time.perf_counter() - t0) / (2 * N) * 1e6), flush=True)
