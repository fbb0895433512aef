"""world_size 2 / 3 / 4 (/ 8) gloo tests of the data-parallel protocol (graphqembed_amd/parallel.py) on CPU:
W ranks x their slice of the batch, loss weights n_rank / n_all, one sum all-reduce of the flat gradient arena
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
        # the single-rank reference: the concatenation of the ranks' slices, full weight
        sl = [parallel.rank_slice(n_pool, B, 1, rr, world) for rr in range(world)]
        cat = np.concatenate([np.arange(s0, e0) for s0, e0 in sl])
        # a rank's weight is n_rank / n_all of the batch's (equal slices: dp_weight = 1 / W; a window that ends at the pool's end is shorter)
        w_local = wgt * (e - s) / float(len(cat))
        if all(e0 - s0 == B for s0, e0 in sl):
            assert abs(w_local - parallel.dp_weight(wgt, world)) < 1e-15
        l, _, _, _ = O.margin_fwd_bwd(params, plan, dec, inter, t[s:e], g[s:e], a[:, s:e], weight=w_local, grads=grads)
        loss_local += w_local * l
        for k, gk in grads.items():
            layout.view(flat, k).add_(torch.from_numpy(gk))
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
    bounds = [parallel.rank_slice(n_pool, B, 1, rr, world) for rr in range(world)]
    for (s0, e0), (s1, e1) in zip(bounds[:-1], bounds[1:]):
        assert e0 == s1 or e0 == n_pool     # (a window that reaches the pool's end is cut there; the next one starts at (it B) mod n: train_helpers.py:100-105)
    assert all(0 < e - s <= B for s, e in bounds)
    with open(os.path.join(out_dir, "ok%d" % rank), "w") as f:
        f.write("ok")
    d_.barrier()
    d_.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 4])
def test_data_parallel_ranks_match_single_rank(tmp_path, world):
    """W ranks x their slice, weights scaled by n_rank / n_all, one sum all-reduce == one rank x the concatenated batch
    (W = 3: the third rank's window wraps around the 200-query pool)."""
    port = 29600 + os.getpid() % 200 + 7 * world
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / ("ok%d" % k)) for k in range(world))


def test_rank_slice_matches_reference_rule():
    from graphqembed_amd import parallel
    for n in (7, 100, 512, 513, 1000):
        for it in range(12):
            s, e = parallel.rank_slice(n, 64, it, 0, 1)
            start = (it * 64) % n
            end = min(((it + 1) * 64) % n, n)
            end = n if end <= start else end
            assert (s, e) == (start, end)


def _shard_worker(rank, world, port, out_dir):
    """Row-sharded protocol ("owner computes", graphqembed_amd/parallel.py) on CPU with the real gloo collectives:
    plan -> request all-to-all -> owners serve rows -> contributions back to the owners -> sharded Adam.  The per-entry
    arithmetic is a stand-in (the fused kernel is GPU-only); ownership, ordering, split sizes and the optimiser
    equivalence are what is under test."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from graphqembed_amd import parallel
    from oracle import netquery_numpy as O
    r, w, _, d_ = parallel.init_from_env("gloo")
    rng = np.random.RandomState(7)                      # same stream on every rank: same tables, both ranks' feeds known
    d = 8
    rows = [53, 20, 31]                                 # three tables; 53 and 31 are not multiples of the world size
    tables = [rng.randn(n, d) for n in rows]
    local_rows = [parallel.shard_rows(n, w) for n in rows]
    head_base = np.concatenate([[0], np.cumsum(local_rows)[:-1]])
    shards = [parallel.shard_of(t, r, w) for t in tables]
    assert all(s.shape[0] == n for s, n in zip(shards, local_rows))
    feeds = []
    for rr in range(w):                                 # rank rr's index feed: ragged, with hub rows (duplicates)
        n = 40 + 9 * rr
        tid = rng.randint(0, 3, size=n)
        idx = np.array([rng.randint(0, rows[t]) for t in tid])
        idx[: n // 4] = idx[0]
        tid[: n // 4] = tid[0]
        feeds.append((idx, tid, rng.randn(n, d)))      # ... and a gradient contribution per index
    idx, tid, contrib = feeds[r]
    pos, req, send = parallel.shard_plan_numpy(idx, tid, head_base, w)
    # requests are grouped by owner, and a request is the owner-local list head of the row
    owner_sorted = (idx % w)[np.argsort(pos)]
    assert np.all(np.diff(owner_sorted) >= 0) and np.array_equal(np.bincount(idx % w, minlength=w), send)
    assert np.array_equal(req[pos], head_base[tid] + idx // w)
    # counts, then the requests themselves
    cs, cr = torch.tensor(send), torch.zeros(w, dtype=torch.int64)
    d_.all_to_all_single(cr, cs)
    recv = [int(c) for c in cr]
    for rr in range(w):                                 # what I receive from rr is what rr's plan sends me
        assert recv[rr] == int(np.bincount(feeds[rr][0] % w, minlength=w)[r])
    req_recv = torch.zeros(sum(recv), dtype=torch.int32)
    d_.all_to_all_single(req_recv, torch.from_numpy(req), output_split_sizes=recv, input_split_sizes=[int(c) for c in send])
    # serve: a request names (table, local row) through the local head index
    flat_shard = np.concatenate(shards)                 # local rows in head order
    served = torch.from_numpy(flat_shard[req_recv.numpy()])
    fetched = torch.zeros(len(idx), d, dtype=torch.float64)
    d_.all_to_all_single(fetched, served, output_split_sizes=[int(c) for c in send], input_split_sizes=recv)
    want_rows = np.stack([tables[t][i] for i, t in zip(idx, tid)])
    np.testing.assert_array_equal(fetched.numpy()[pos], want_rows)      # the position feed finds every row of the index feed
    # contributions: written at the row's position, sent back along the same splits, linked (here: added) by the owner
    csend = torch.zeros(len(idx), d, dtype=torch.float64)
    csend[torch.from_numpy(pos).long()] = torch.from_numpy(contrib)
    crecv = torch.zeros(sum(recv), d, dtype=torch.float64)
    d_.all_to_all_single(crecv, csend, output_split_sizes=recv, input_split_sizes=[int(c) for c in send])
    grad_local = np.zeros_like(flat_shard)
    np.add.at(grad_local, req_recv.numpy(), crecv.numpy())
    grad_full = [np.zeros_like(t) for t in tables]      # the single-rank statement: every rank's contributions scattered
    for (i2, t2, c2) in feeds:
        for i, t, c in zip(i2, t2, c2):
            grad_full[t][i] += c
    want_local = np.concatenate([parallel.shard_of(g, r, w) for g in grad_full])
    np.testing.assert_allclose(grad_local, want_local, rtol=1e-12, atol=1e-12)
    # Adam on the own shards == the rows of Adam on the whole tables (dense: every row moves, gradient or not)
    full_p = {"t%d" % k: t.copy() for k, t in enumerate(tables)}
    full_g = {"t%d" % k: g for k, g in enumerate(grad_full)}
    st_full = {}
    loc_p = {"t%d" % k: s.copy() for k, s in enumerate(shards)}
    off = np.concatenate([[0], np.cumsum(local_rows)])
    loc_g = {"t%d" % k: grad_local[off[k]:off[k + 1]] for k in range(3)}
    st_loc = {}
    for _ in range(3):
        O.adam_step(full_p, full_g, st_full, list(full_p))
        O.adam_step(loc_p, loc_g, st_loc, list(loc_p))
    for k in range(3):
        got, want = loc_p["t%d" % k], parallel.shard_of(full_p["t%d" % k], r, w)
        n_real = len(tables[k][r::w])
        np.testing.assert_allclose(got[:n_real], want[:n_real], rtol=1e-12, atol=1e-12)
    with open(os.path.join(out_dir, "shard_ok%d" % rank), "w") as f:
        f.write("ok")
    d_.barrier()
    d_.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 4, 8])
def test_row_sharded_protocol_ranks(tmp_path, world):
    """Tables of 53 / 20 / 31 rows over 2, 3, 4 and 8 ranks: ceil(rows / W) leaves short last shards (at W = 8 the 20-row table
    gives ranks 4 .. 7 two rows and the others three), the owner sort has W buckets, every all-to-all W blocks."""
    port = 29300 + os.getpid() % 90 + 11 * world
    mp.spawn(_shard_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / ("shard_ok%d" % k)) for k in range(world))
