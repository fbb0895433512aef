#!/usr/bin/env python3
"""Randomised differential run of train_helpers.run_train: its native runs against the per-batch path (GQE_RUN_TRAIN_NATIVE=0) over
random seeds, decoders, batch sizes, optimisers, burn-in lengths and validation intervals on the tiny golden world — log lines to
float-atomics noise, generator states and step counters exactly.  python tools/probes/run_train_soak.py  (24 trials, ~10 s)"""
import os, sys, random, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from test_gpu_api import build_world, rebuild_queries
from graphqembed_amd import train_helpers
from graphqembed_amd.model import FusedAdam, FusedSGD

class Log(object):
    def __init__(self): self.lines = []
    def info(self, m): self.lines.append(m)

def run(native, seed, dec, inter, B, opt, burn, val_every, iters):
    os.environ["GQE_RUN_TRAIN_NATIVE"] = "1" if native else "0"
    model, _ = build_world(dec, inter, 32, "train_%s_%s_d32.npz" % (dec, inter))
    train, test = rebuild_queries()
    random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
    log = Log()
    o = FusedAdam(model, lr=0.01) if opt == "adam" else FusedSGD(model, lr=0.05)
    train_helpers.run_train(model, o, train, test, test, log, max_burn_in=burn, batch_size=B, log_every=1, val_every=val_every, max_iter=iters)
    return log.lines, random.getstate(), np.random.get_state(), dict(model.engine.steps)

rng = np.random.RandomState(0)
bad = 0
for trial in range(int(os.environ.get("SOAK_TRIALS", "24"))):
    dec, inter = [("bilinear-diag", "min"), ("bilinear", "mean"), ("transe", "min-simple")][trial % 3]
    B = int(rng.choice([1, 3, 17, 23, 64, 500]))
    opt = "adam" if trial % 4 else "sgd"
    burn = int(rng.randint(1, 9)); val_every = int(rng.choice([2, 3, 5, 1000])); iters = int(rng.randint(3, 25))
    a = run(True, trial, dec, inter, B, opt, burn, val_every, iters)
    b = run(False, trial, dec, inter, B, opt, burn, val_every, iters)
    ok = a[1] == b[1] and np.array_equal(a[2][1], b[2][1]) and a[2][2] == b[2][2] and a[3] == b[3] and len(a[0]) == len(b[0])
    if ok:
        for x, y in zip(a[0], b[0]):
            tx, ty = x.split(), y.split()
            if len(tx) != len(ty): ok = False; break
            for u, v in zip(tx, ty):
                try:
                    if abs(float(u.strip(";")) - float(v.strip(";"))) > 3e-2 * max(1.0, abs(float(v.strip(";")))): ok = False
                except ValueError:
                    if u != v: ok = False
    print(trial, dec, inter, B, opt, burn, val_every, iters, "OK" if ok else "MISMATCH", flush=True)
    bad += 0 if ok else 1
print("mismatches:", bad)
