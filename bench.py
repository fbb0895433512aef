#!/usr/bin/env python3
"""bench.py — queries/sec of the fused MI355X training step on the BASELINE workload.

    python bench.py [--gpus N --steps K --warmup W]            # N=1 directly
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

step      = one training iteration of the reference's post-burn-in schedule over the
            "Bio full conjunctive mix" (train_helpers.py:50-79): 9 batches x B=512 =
            4 608 (query, negative) pairs -> fused forward/backward (one grouped launch),
            deferred matrix gradients, [N>1: gradient exchange — one all-gather of per-rank
            contribution slabs, or --exchange dense: all-reduce of the gradient arena],
            one fused dense Adam step (+ grad re-zero).
workload  = "bio-synth" (SURVEY.md §8d C3): 5 modes / 97 000 nodes / 14 directed relations,
            d=128, bilinear-diag decoder + SetIntersection(min), P = 12 582 912 parameters;
            index feeds of 32 distinct pre-sampled iterations are resident in HBM.
value     = whole-job queries/s = K * 4608 * N / max-over-ranks wall time (weak scaling:
            every rank trains its own 4 608 queries per step; gradients are averaged).
roofline  = the dominant kernel (fused Adam pass): algorithmic bytes 32 B/param/step
            (SURVEY.md §8d A_step) / its mean launch duration measured with hipEvents on the
            launch stream inside the timed region (every 4th launch); peak 8 TB/s
            (MI355X_MICROARCH.md).  ``traffic`` = PMC HBM bytes of the last profiled run: it is
            BELOW the algorithmic bytes because embedding-row gradients are kept as per-row
            lists, so the dense table gradient is neither read nor re-zeroed (24 instead of
            32 B/param) — see DESIGN.md §3.
lazy_exact_adam (N=1, extra key, NOT the headline) = the same loop with gqe_set_lazy_adam:
            zero-gradient Adam steps of untouched rows are deferred and replayed bit-exactly
            when the row is next needed; the final sync is inside its timed region.
cpu_baseline = oracle/netquery_torch.py (torch-CPU port of the reference's iteration: two
            eager forwards per batch, one autograd backward, dense torch.optim.Adam) on the
            same parameters and the same batches, timed on this host (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0


def pmc_traffic(kernel):
    """HBM bytes per launch from the newest committed PMC summary (profiles/r*_pmc_traffic.json); the
    counters cannot be collected from inside this process, so this is the last profiled value."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not files:
        return None
    with open(files[-1]) as f:
        return json.load(f).get(kernel, {}).get("hbm_bytes_per_launch")


def build_layout(g, d, decoder, inter):
    from graphqembed_amd.engine import ArenaLayout
    from graphqembed_amd.tensorize import post_key, pre_key, rel_key, table_key
    layout = ArenaLayout()
    for m in g.modes:
        layout.add(table_key(m), (g.mode_sizes[m] + 2, d))        # len(node_maps)+1 rows (bio/data_utils.py:14-17)
    for m in g.relations:                                         # decoders.py:136-140 order
        for (to, name) in g.relations[m]:
            layout.add(rel_key((m, name, to)), (d, d) if decoder == "bilinear" else (d,))
    if not inter.endswith("simple"):
        for m in g.modes:
            layout.add(pre_key(m), (d, d))
            layout.add(post_key(m), (d, d))
    return layout


def init_params(eng, d, seed):
    """The reference's initial distributions (bio/data_utils.py:19, decoders.py:139,225,282-285)."""
    import torch
    gen = torch.Generator(device=eng.device)
    gen.manual_seed(seed)
    for k, (off, shape) in eng.layout.entries.items():
        v = eng.layout.view(eng.params, k)
        if k.startswith("enc."):
            v.normal_(0, 1.0 / d, generator=gen)
        elif len(shape) == 1:
            v.uniform_(-6.0 / np.sqrt(d), 6.0 / np.sqrt(d), generator=gen)
        else:
            lim = np.sqrt(6.0 / (shape[0] + shape[1]))
            v.uniform_(-lim, lim, generator=gen)


def algorithmic_bytes_per_query(qtype, d):
    rows = {"1-chain": 3, "2-chain": 3, "3-chain": 3, "2-inter": 4, "3-inter_chain": 4, "3-chain_inter": 4, "3-inter": 5}[qtype]
    return rows * (8 * d + 4)                                      # SURVEY.md §8d A_q


def cpu_baseline(eng, decoder, inter, item_sets, budget_s, queries_per_iter):
    """Time the torch-CPU port on the same parameters / batches (bounded sample).  torch's default
    (one thread per hardware thread) is far from its best on a 2x64-core host, so a short sweep over
    thread counts picks the fastest setting and the remaining budget is spent measuring it."""
    import torch
    from oracle.netquery_numpy import make_plan
    from oracle.netquery_torch import TorchPort
    torch.cuda.synchronize()
    host = eng.params.cpu().numpy()
    params = {k: host[off:off + int(np.prod(shape))].reshape(shape).copy() for k, (off, shape) in eng.layout.entries.items()}
    port = TorchPort(params, decoder, inter)
    sets = [[(make_plan(f.query_type, f.rels), t, g, a, w, m) for (f, t, g, a, w, m) in items] for items in item_sets]
    default_threads = torch.get_num_threads()
    port.train_iteration(sets[0])                                  # warm-up (allocator, Adam state)
    best = (None, 1e30)
    for nt in sorted(set([8, 16, 32, 64, default_threads])):
        if nt > default_threads:
            continue
        torch.set_num_threads(nt)
        port.train_iteration(sets[0])
        t0 = time.time()
        for k in range(2):
            port.train_iteration(sets[(k + 1) % len(sets)])
        dt = (time.time() - t0) / 2
        if dt < best[1]:
            best = (nt, dt)
    torch.set_num_threads(best[0])
    n, t0 = 0, time.time()
    while True:
        port.train_iteration(sets[(n + 1) % len(sets)])
        n += 1
        el = time.time() - t0
        if el >= budget_s or n >= 400:
            break
    torch.set_num_threads(default_threads)
    return {"value": round(n * queries_per_iter / el, 1), "unit": "queries/s", "cores": int(best[0]),
            "kind": "port", "sample": "%d full-mix iterations (9x512 queries, P=%d, dense torch Adam) in %.1f s with %d threads "
            "(best of a sweep; host exposes %d); oracle/netquery_torch.py, torch %s"
            % (n, eng.layout.total, el, best[0], default_threads, torch.__version__)}


def lazy_measurement(args, layout, d, qpi, item_sets, plans, n_distinct):
    """The same training loop with gqe_set_lazy_adam (include/gqe.h): rows without a gradient are not streamed every
    step, their zero-gradient Adam steps are replayed — with the eager pass's exact arithmetic — when the row is next
    read or stepped.  Reported NEXT TO the headline value, never as it: `value` above is the eager schedule.  The
    timed region ends with gqe_optimizer_sync, so every deferred step is paid for inside it."""
    import torch
    from graphqembed_amd.engine import Engine
    from graphqembed_amd.tensorize import pack_margin_batches
    eng = Engine(d, args.decoder, args.inter_decoder, layout, max_queries=qpi, max_batches=len(item_sets[0]), lazy_adam=True)
    init_params(eng, d, seed=0)
    prepared = []
    for items in item_sets:
        packed = [(plans[f], t, ng, a, w, m) for (f, t, ng, a, w, m) in items]
        descs, idx, _ = pack_margin_batches(packed)
        ps = eng.prepare_margin(descs, torch.from_numpy(idx).to(eng.device))
        ps["adam"] = eng.prepare_adam(set().union(*[p[0].touched for p in packed]))
        prepared.append(ps)

    def step(i):
        ps = prepared[i % n_distinct]
        eng.run_margin(ps)
        eng.run_adam(ps["adam"])

    eng.timing_enable(4)
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    for k in range(5):
        eng.timing_read(k)
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    eng.sync()                                  # settle every deferred step inside the timed region
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    names = ["fused_fwd_bwd", "pair_gemm", "optimiser_rows_or_full_pass", "optimiser_small_tensors", "catch_up_before_read"]
    kernels = {}
    for k, nm in enumerate(names):
        ms, n = eng.timing_read(k)
        kernels[nm] = {"avg_launch_ms": round(ms, 5), "launches": n}
    eng.timing_enable(0)
    loss = float(prepared[(args.warmup + args.steps - 1) % n_distinct]["losses"][-1].item())
    eng.close()
    return {"value": round(args.steps * qpi / elapsed, 1), "unit": "queries/s", "ms_per_step": round(elapsed * 1e3 / args.steps, 4),
            "final_loss": round(loss, 6), "kernels": kernels,
            "note": "same workload and step count; deferred zero-gradient Adam steps are replayed bit-exactly on demand "
                    "(tests/test_gpu_parity.py::test_lazy_adam_is_bit_identical_to_the_eager_schedule); a full pass every "
                    "<= 62 steps per table and the final gqe_optimizer_sync are inside the timed region"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--batch-size", type=int, default=512)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--decoder", default="bilinear-diag")
    ap.add_argument("--inter-decoder", default="min")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for --gpus > 1 (nccl = RCCL; gloo only to "
                    "smoke-test the multi-rank path on a single GPU)")
    ap.add_argument("--exchange", default="sparse", choices=["sparse", "dense"],
                    help="--gpus > 1: all-gather the gradient contribution entries (sparse) or all-reduce the dense arena")
    ap.add_argument("--lazy-adam", action="store_true", help="run the MAIN measurement in lazy-Adam mode (non-default; the config "
                    "then says so).  Works with --gpus N and the sparse exchange.")
    ap.add_argument("--no-lazy", action="store_true", help="skip the secondary measurement of the lazy (deferred, bit-exact) Adam mode")
    ap.add_argument("--check-replicas", action="store_true", help="after the run, verify that all ranks hold identical parameters")
    args = ap.parse_args()

    import torch
    from graphqembed_amd import parallel
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%s: launch with torch.distributed.run --nproc-per-node %d"
                         % (args.gpus, os.environ.get("WORLD_SIZE", "1"), args.gpus))
    n_dev = torch.cuda.device_count()
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % max(n_dev, 1))
    rank, world, local_rank, dist = parallel.init_from_env(args.backend)

    from graphqembed_amd import synth
    from graphqembed_amd.engine import Engine
    from graphqembed_amd.tensorize import FormulaPlan, pack_margin_batches

    d, B = args.dim, args.batch_size
    g = synth.bio_synth(seed=0)
    layout = build_layout(g, d, args.decoder, args.inter_decoder)
    mix = synth.FULL_MIX
    qpi = B * len(mix)                                             # queries per iteration per GPU
    sparse = world > 1 and args.exchange == "sparse"
    eng = Engine(d, args.decoder, args.inter_decoder, layout, max_queries=qpi, max_batches=len(mix),
                 rank=rank if sparse else 0, world=world if sparse else 1, lazy_adam=args.lazy_adam)
    init_params(eng, d, seed=0)                                    # same seed on every rank: replicas start equal
    pools = synth.make_pools(g, sorted(set(m[0] for m in mix)), formulas_per_type=6, pool_size=max(16 * B, 8192), seed=0)

    n_distinct = 32
    item_sets, prepared = [], []
    plans = {}
    for s in range(n_distinct):
        items = synth.mix_iteration(pools, mix, s, B, rank=rank, world=world)
        item_sets.append(items)
        packed = []
        for (f, t, ng, a, w, m) in items:
            if f not in plans:
                plans[f] = FormulaPlan(f, layout, args.inter_decoder)
            packed.append((plans[f], t, ng, a, w, m))
        descs, idx, _ = pack_margin_batches(packed)
        ps = eng.prepare_margin(descs, torch.from_numpy(idx).to(eng.device))
        ps["adam"] = eng.prepare_adam(set().union(*[p[0].touched for p in packed]))
        ps["n_entries"] = sum((2 + a.shape[0]) * len(t) for (_, t, _, a, _, _) in items)
        ps["aq_bytes"] = sum(algorithmic_bytes_per_query(f.query_type, d) * len(t) for (f, t, _, _, _, _) in items)
        ps["p_touched"] = sum(layout.numel(k) for k in ps["adam"]["keys"])
        prepared.append(ps)

    mode = {"sparse": sparse}

    def step(i):
        ps = prepared[i % n_distinct]
        eng.run_margin(ps)
        if mode["sparse"]:                                         # contribution entries all-gathered over xGMI
            try:
                parallel.exchange_sparse(eng, dist)
            except Exception as e:                                 # argument-level refusal by the backend: same on all ranks
                if i != 0:
                    raise
                sys.stderr.write("bench: sparse exchange refused (%s); falling back to the dense all-reduce\n" % e)
                mode["sparse"] = False
                parallel.exchange_gradients(eng.grads, dist, engine=eng)
        elif dist is not None:                                     # lists -> dense arena, RCCL sum over xGMI
            parallel.exchange_gradients(eng.grads, dist, engine=eng)
        eng.run_adam(ps["adam"])

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    eng.timing_enable(4)                                           # every 4th launch; hipEvent pairs are recycled
    for i in range(args.warmup):
        step(i)
    fence()
    for k in range(3):
        eng.timing_read(k)
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    eng.sync()                                                     # lazy Adam: deferred steps are settled inside the timed region
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=eng.device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    ms_fused, n_fused = eng.timing_read(0)
    ms_gemm, n_gemm = eng.timing_read(1)
    ms_opt, n_opt = eng.timing_read(2)
    eng.timing_enable(0)
    loss = float(prepared[(args.warmup + args.steps - 1) % n_distinct]["losses"][-1].item())
    if not np.isfinite(loss):
        raise SystemExit("non-finite loss")

    ms_per_step = elapsed * 1e3 / args.steps
    value = args.steps * qpi * world / elapsed
    used = [prepared[(args.warmup + i) % n_distinct] for i in range(min(args.steps, n_distinct))]
    a_step = 32.0 * np.mean([p["p_touched"] for p in used])        # bytes per optimiser launch
    opt_kernel = "gqe_opt_kernel<ADAM> (fused Adam + grad re-zero)"
    if args.lazy_adam:                                             # the row launch: 32 B per parameter of the rows it names
        a_step = 32.0 * d * world * np.mean([p["n_entries"] for p in used])
        opt_kernel = "gqe_rows_kernel (lazy Adam: rows of the step; duplicates counted once per entry)"
    a_q = float(np.mean([p["aq_bytes"] for p in used]))            # bytes per fused fwd/bwd launch
    achieved = a_step / (ms_opt * 1e-3) / 1e9 if ms_opt > 0 else 0.0
    out = {
        "metric": "queries/sec, Bio full conjunctive mix d=%d, at 1/2/4/8 MI355X" % d,
        "value": round(value, 1), "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "bio-synth full mix (1/2/3-chain, 2/3-inter x{neg,hard}, 3-inter_chain x{neg,hard}): "
                               "%d batches x B=%d per GPU per step, d=%d, %s + SetIntersection(%s), P=%d, Adam lr 0.01"
                               % (len(mix), B, d, args.decoder, args.inter_decoder, layout.total),
                   "graph": "5 modes, 97000 nodes, 14 directed relations, 60000 edges/kind, seed 0",
                   "queries_per_step_per_gpu": qpi, "parallelism": "dp%d" % world,
                   "optimizer": "lazy (deferred, bit-exact) Adam — NON-DEFAULT mode" if args.lazy_adam else "eager dense Adam",
                   "gradient_exchange": "none" if world == 1 else
                   ("one all-gather per step of per-rank slabs: %d contribution entries x (%d floats + row id) + the dense "
                    "relation/Pre/Post gradients" % (prepared[0]["n_entries"], d)) if mode["sparse"] else
                   "all-reduce of the %d-float gradient arena" % layout.total},
        "roofline": {"bound": "hbm", "kernel": opt_kernel,
                     "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4),
                     "traffic": pmc_traffic("gqe_opt_kernel") if (d, B, args.decoder, args.inter_decoder, args.lazy_adam) == (128, 512, "bilinear-diag", "min", False) else None,
                     "algorithmic_bytes_per_launch": a_step, "avg_launch_ms": round(ms_opt, 5), "launches": n_opt},
        "kernels": {"fused_fwd_bwd": {"avg_launch_ms": round(ms_fused, 5), "launches": n_fused,
                                      "algorithmic_bytes_per_launch": a_q,
                                      "achieved_GBs": round(a_q / (ms_fused * 1e-3) / 1e9, 1) if ms_fused > 0 else None},
                    "pair_gemm": {"avg_launch_ms": round(ms_gemm, 5), "launches": n_gemm}},
        "step_roofline": {"algorithmic_bytes_per_step": a_step + a_q,
                          "achieved_GBs": round((a_step + a_q) / (ms_per_step * 1e-3) / 1e9, 1),
                          "frac": round((a_step + a_q) / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
        "final_loss": round(loss, 6),
    }
    if world == 1 and not args.no_lazy and not args.lazy_adam:
        out["lazy_exact_adam"] = lazy_measurement(args, layout, d, qpi, item_sets, plans, n_distinct)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(eng, args.decoder, args.inter_decoder, item_sets[:8], args.cpu_seconds, qpi)
    elif rank == 0:
        out["cpu_baseline"] = None
    if args.check_replicas and dist is not None:
        ref = eng.params.clone()
        dist.broadcast(ref, 0)
        same = torch.tensor([int(torch.equal(ref, eng.params))], device=eng.device)
        dist.all_reduce(same, op=dist.ReduceOp.MIN)
        out["replicas_identical"] = bool(same.item() == 1)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
