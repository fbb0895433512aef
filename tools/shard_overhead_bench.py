#!/usr/bin/env python3
"""What the row-sharded protocol costs when nothing has to travel: ONE rank over the real nccl (RCCL) backend, so every
all-to-all / all-reduce is a self-copy executed by RCCL — serve, fetch, fused launch on fetched rows, contributions back,
link, all-reduce, Adam — against the plain single-GPU step on the same batches.  The difference is the fixed cost of the
collectives + the serve kernel for the other ranks (none at one rank since round 4: own rows are read in place), i.e. the floor under the
exchange time of an N-GPU run.   python tools/shard_overhead_bench.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist
import bench
from graphqembed_amd import parallel, synth

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29877")
os.environ["GQE_SHARD_PROFILE"] = "1"                # host time per phase inside the library (gqe_shard_profile)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
wl = bench.Workload("bio-synth", 128, "bilinear-diag", "min", synth.FULL_MIX, 512)
from graphqembed_amd.tensorize import FormulaPlan, pack_margin_batches
for label, shard in (("plain", None), ("plain + gqe_set_deferred_gemm (the sharded step cannot defer)", None), ("phases driven from Python (torch.distributed collectives), planned ONCE", (0, 1)),
                     ("gqe_shard_step, planned EVERY step, every block through RCCL", (0, 1)),
                     ("gqe_shard_step, planned EVERY step, own block kept in place (default)", (0, 1))):
    one_call = label.startswith("gqe_shard_step")
    if "through RCCL" in label:
        os.environ["GQE_SHARD_SELF_VIA_RCCL"] = "1"       # read by gqe_shard_open
    else:
        os.environ.pop("GQE_SHARD_SELF_VIA_RCCL", None)
    eng = wl.engine(shard=shard)
    if "gqe_set_deferred_gemm" in label:
        eng.set_deferred_gemm(True)
    prepared = wl.prepare(eng, dist)                  # sharded: host feeds for gqe_shard_post
    if shard and not one_call:                        # the round-2 form: requests exchanged once per pre-sampled iteration, outside the loop
        legacy = []
        for items in wl.item_sets:
            packed = [(FormulaPlan(f, eng.layout, wl.inter), t, ng, a, w, m) for (f, t, ng, a, w, m) in items]
            descs, idx, _ = pack_margin_batches(packed)
            ps = parallel.shard_prepare(eng, dist, descs, idx)
            ps["adam"] = eng.prepare_adam(set().union(*[p[0].touched for p in packed]))
            legacy.append(ps)
    if one_call:
        comm = parallel.RcclComm(0, 1)                # the library's own RCCL calls: self send / recv groups + all-reduce
        eng.shard_open(None, nccl_comm=comm.handle)
        state = {"next": None}

    def step(i):
        if one_call:
            if state["next"] != i:
                eng.shard_post(prepared[i % wl.n_distinct])
            eng.shard_post(prepared[(i + 1) % wl.n_distinct])     # host planning of the next step
            state["next"] = i + 1
            eng.shard_step(prepared[i % wl.n_distinct])
            return
        if shard:
            ps = legacy[i % wl.n_distinct]
            parallel.shard_fetch(eng, dist, ps)
            eng.run_margin(ps)
            parallel.shard_exchange(eng, dist, ps)
        else:
            ps = prepared[i % wl.n_distinct]
            eng.run_margin(ps)
        eng.run_adam(ps["adam"])
    for i in range(50):
        step(i)
    torch.cuda.synchronize()
    times = []
    for rep in range(20):
        t0 = time.perf_counter()
        for i in range(100):
            step(50 + rep * 100 + i)
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) / 100)
    t0 = time.perf_counter()                                  # host time alone: enqueue without waiting
    for i in range(2050, 2250):
        step(i)
    host = (time.perf_counter() - t0) / 200
    torch.cuda.synchronize()
    line = "%-74s %7.1f us/step (median of 20 x 100 steps)" % (label, np.median(times) * 1e6)
    if one_call:
        # the un-synchronised loop runs ahead of the GPU until the ring of 8 pinned feeds is full and then waits for step t - 8:
        # what it takes beyond the library's own work is back-pressure, not host work
        plan_us, main_us, n = eng.shard_profile()
        line += ("; host: main thread %.1f us/step inside gqe_shard_step, planning thread %.1f us/step (owner sort + publication, off the "
                 "critical path), un-synchronised loop %.1f us/step of which %.1f us is back-pressure (waiting for the GPU)"
                 % (main_us, plan_us, host * 1e6, max(host * 1e6 - main_us, 0.0)))
    else:
        line += ", host enqueue %6.1f us/step" % (host * 1e6)
    print(line, flush=True)
    eng.close()
dist.destroy_process_group()
