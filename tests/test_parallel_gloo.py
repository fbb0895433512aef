"""world_size-2 gloo test of the data-parallel protocol (graphqembed_amd/parallel.py) on CPU:
2 ranks x half batch, loss weights / 2, one sum all-reduce of the flat gradient arena
== 1 rank x full batch.  The per-rank compute is the numpy oracle (no GPU here); what is
under test is the sharding rule, the weight scaling and the collective on the arena layout."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from graphqembed_amd import parallel
    from graphqembed_amd.engine import ArenaLayout
    from gpu_utils import TOY_FORMULAS, TOY_KINDS, TOY_SIZES, random_params, toy_batch
    from oracle import netquery_numpy as O
    r, w, _, d_ = parallel.init_from_env("gloo")
    assert (r, w) == (rank, world) and d_ is not None
    rng = np.random.RandomState(3)                       # same seed on every rank: same params, same global batch
    dec, inter, d = "bilinear-diag", "min", 32
    params = random_params(rng, d, dec, inter, TOY_SIZES, TOY_KINDS)
    layout = ArenaLayout()
    for k, v in params.items():
        layout.add(k, v.shape)
    mix = [("1-chain", 1.0), ("2-inter", 0.005), ("3-inter_chain", 0.005)]
    n_pool, B = 200, 32
    flat = torch.zeros(layout.total, dtype=torch.float64)
    full = O.zero_grads_like(params)
    loss_local, loss_full = 0.0, 0.0
    for qtype, wgt in mix:
        t, g, a = toy_batch(rng, qtype, n_pool)
        plan = O.make_plan(qtype, TOY_FORMULAS[qtype])
        grads = O.zero_grads_like(params)
        s, e = parallel.rank_slice(n_pool, B, step=1, rank=rank, world=world)
        l, _, _, _ = O.margin_fwd_bwd(params, plan, dec, inter, t[s:e], g[s:e], a[:, s:e],
                                      weight=parallel.dp_weight(wgt, world), grads=grads)
        loss_local += parallel.dp_weight(wgt, world) * l
        for k, gk in grads.items():
            layout.view(flat, k).add_(torch.from_numpy(gk))
        # the single-rank reference: the concatenation of both ranks' slices, full weight
        sl = [parallel.rank_slice(n_pool, B, 1, rr, world) for rr in range(world)]
        cat = np.concatenate([np.arange(s0, e0) for s0, e0 in sl])
        lf, _, _, _ = O.margin_fwd_bwd(params, plan, dec, inter, t[cat], g[cat], a[:, cat], weight=wgt, grads=full)
        loss_full += wgt * lf
    parallel.exchange_gradients(flat, d_)
    tl = torch.tensor([loss_local], dtype=torch.float64)
    d_.all_reduce(tl)
    got = {k: layout.view(flat, k).numpy() for k in params}
    for k in params:
        np.testing.assert_allclose(got[k], full[k], rtol=1e-9, atol=1e-12, err_msg=k)
    np.testing.assert_allclose(tl.item(), loss_full, rtol=1e-10)
    # slices of one step are consecutive and disjoint
    s0, e0 = parallel.rank_slice(n_pool, B, 1, 0, world)
    s1, e1 = parallel.rank_slice(n_pool, B, 1, 1, world)
    assert e0 == s1 and e1 - s1 == B
    with open(os.path.join(out_dir, "ok%d" % rank), "w") as f:
        f.write("ok")
    d_.barrier()
    d_.destroy_process_group()


def test_two_rank_data_parallel_matches_single_rank(tmp_path):
    port = 29600 + os.getpid() % 200
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(tmp_path / "ok0") and os.path.exists(tmp_path / "ok1")


def test_rank_slice_matches_reference_rule():
    from graphqembed_amd import parallel
    for n in (7, 100, 512, 513, 1000):
        for it in range(12):
            s, e = parallel.rank_slice(n, 64, it, 0, 1)
            start = (it * 64) % n
            end = min(((it + 1) * 64) % n, n)
            end = n if end <= start else end
            assert (s, e) == (start, end)
