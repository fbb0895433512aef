"""Data-parallel gradient exchange on the GPU: 2 gloo ranks sharing cuda:0 (the only GPU of a test box).

  * sparse exchange (one all-gather of per-rank slabs + gqe_import_entries) == dense exchange
    (gqe_materialize_grads + all-reduce of the arena) on the same sharded batches;
  * both == the single-rank gradient of the concatenated batch (the numpy oracle);
  * after the optimiser step the replicas are BIT-identical (lists are summed in entry order), including rows
    whose lists are longer than two entries;
  * ranks with different batch sizes (slab reservation);
  * a lazy-Adam engine in exchange mode: replicas bit-identical, parameters equal to the eager engine's up to atomics noise;
  * state machine: a second margin call before the step, and a second import, are refused.
"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _worker(rank, world, port, out_dir, dec, inter, bag_modes=()):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from graphqembed_amd import parallel
    from graphqembed_amd.engine import GqeError
    from graphqembed_amd.tensorize import pack_margin_batches
    from gpu_utils import TOY_FORMULAS, TOY_KINDS, TOY_SIZES, engine_from_params, plan_for, random_params, read_arena, toy_batch
    from oracle import netquery_numpy as O
    r, w, _, dist = parallel.init_from_env("gloo")
    rng = np.random.RandomState(11)
    d = 32
    params = random_params(rng, d, dec, inter, TOY_SIZES, TOY_KINDS, bag_modes=bag_modes)
    mix = [("1-chain", 1.0), ("2-chain", 0.3), ("2-inter", 0.5), ("3-inter_chain", 0.5)]
    n_pool, B = 400, 96
    sparse = engine_from_params(params, d, dec, inter, rank=r, world=w)      # steps from the lists
    sparse2 = engine_from_params(params, d, dec, inter, rank=r, world=w)     # lists folded, to compare gradients
    dense = engine_from_params(params, d, dec, inter)
    assert sparse.sparse_exchange and not dense.sparse_exchange
    slab = sum((2 + len(O.make_plan(q, TOY_FORMULAS[q])["anchor_modes"])) * B for q, _ in mix)
    sparse.exchange_reserve(slab)
    sparse2.exchange_reserve(slab)
    full = O.zero_grads_like(params)
    items = []
    for qtype, wgt in mix:
        t, g, a = toy_batch(rng, qtype, n_pool, hub=(qtype == "2-inter"))     # hub rows: lists of ~100 entries
        # step 2 wraps around the pool: rank 0 gets a short slice (16 queries), rank 1 a full one -> unequal entry
        # counts (Engine.exchange_reserve) and loss weights n_r / n_total instead of 1 / world
        s, e = parallel.rank_slice(n_pool, B, 2, r, w)
        cat = np.concatenate([np.arange(*parallel.rank_slice(n_pool, B, 2, rr, w)) for rr in range(w)])
        assert e - s != len(cat) - (e - s)
        items.append((qtype, t[s:e], g[s:e], a[:, s:e], wgt * (e - s) / float(len(cat))))
        O.margin_fwd_bwd(params, O.make_plan(qtype, TOY_FORMULAS[qtype]), dec, inter, t[cat], g[cat], a[:, cat],
                         weight=wgt, grads=full)

    def launch(eng):
        packed = [(plan_for(eng, q, TOY_FORMULAS[q]), t, g, a, wgt, 1.0) for (q, t, g, a, wgt) in items]
        descs, idx, _ = pack_margin_batches(packed)
        eng.margin_fwd_bwd(descs, idx)
        return set().union(*[p[0].touched for p in packed])

    keys = launch(sparse)
    launch(sparse2)
    launch(dense)
    with pytest.raises(GqeError):           # exchange mode: one margin call per optimiser step
        launch(sparse)
    parallel.exchange_sparse(sparse, dist)
    with pytest.raises(GqeError):           # the step's entries are imported once
        sparse.import_entries(sparse.world)
    parallel.exchange_sparse(sparse2, dist)
    parallel.exchange_gradients(dense.grads, dist, engine=dense)
    g_sparse = read_arena(sparse2, sparse2.grads)       # materialises the (local + imported) lists
    g_dense = read_arena(dense, dense.grads)
    for k in params:
        if k == O.BAGS_KEY:
            continue
        scale = max(1e-6, float(np.abs(full[k]).max()))
        np.testing.assert_allclose(g_sparse[k], g_dense[k], rtol=0, atol=2e-5 * scale, err_msg="sparse vs dense " + k)
        np.testing.assert_allclose(g_sparse[k], full[k], rtol=0, atol=2e-4 * scale, err_msg="sparse vs oracle " + k)
    # the toy tables are small: some rows must have collected more than two entries (the re-summed case)
    longest = 0
    for qtype, t, g, a, _ in items:
        longest = max(longest, int(np.bincount(np.concatenate([t, g])).max()))
    longest = torch.tensor([longest])
    dist.all_reduce(longest, op=dist.ReduceOp.MAX)
    assert int(longest.item()) > 40
    # a lazy-Adam engine in exchange mode next to the eager one: its row launch walks the gathered slabs
    lazy = engine_from_params(params, d, dec, inter, rank=r, world=w, lazy_adam=True)
    lazy.exchange_reserve(slab)
    for step in range(6):
        for eng in (sparse, lazy):
            if step or eng is lazy:
                launch(eng)
                parallel.exchange_sparse(eng, dist)
            eng.adam_step(keys, 0.01)
    torch.cuda.synchronize()
    assert not torch.equal(lazy._params, sparse._params)      # rows the six steps never touched still owe their steps
    for eng, what in ((sparse, "eager"), (lazy, "lazy")):
        mine = eng.params.clone()                              # (the property settles the lazy engine's debts)
        ref = mine.clone()
        dist.broadcast(ref, 0)
        same = torch.tensor([int(torch.equal(mine, ref))])
        dist.all_reduce(same, op=dist.ReduceOp.MIN)
        assert int(same.item()) == 1, "%s replicas diverged" % what
    # lazy == eager up to the run-to-run noise of the dense gradients' atomics, which Adam amplifies to ~lr per step
    diff = (lazy.params - sparse.params).abs()
    assert float(diff.max()) < 0.03 and float((diff > 1e-4).float().mean()) < 0.02, (float(diff.max()), float((diff > 1e-4).float().mean()))
    lazy.close()
    with open(os.path.join(out_dir, "ok%d" % rank), "w") as f:
        f.write("ok")
    dist.barrier()
    dist.destroy_process_group()
    for e in (sparse, sparse2, dense):
        e.close()


@pytest.mark.parametrize("dec,inter,bag_modes", [("bilinear-diag", "min", ()), ("bilinear", "mean", ()), ("transe", "min-simple", ()),
                                                 ("bilinear-diag", "min", ("a",))])
def test_sparse_exchange_two_ranks(tmp_path, dec, inter, bag_modes):
    """bag_modes=("a",): mode a is an EmbeddingBag mode (Reddit posts) — its contributions travel as (vector, bag id)
    and the importer re-expands the bag into link nodes."""
    port = 29800 + os.getpid() % 150
    mp.spawn(_worker, args=(2, port, str(tmp_path), dec, inter, bag_modes), nprocs=2, join=True)
    assert os.path.exists(tmp_path / "ok0") and os.path.exists(tmp_path / "ok1")


def _trainer_worker(rank, world, port, out_dir):
    """TensorizedTrainer, 2 ranks: same formula draws, own query slices and negatives, sparse exchange every
    iteration -> replicas stay bit-identical and the loss falls."""
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from bench import build_layout, init_params
    from graphqembed_amd import parallel, synth
    from graphqembed_amd.data_utils import BIO_TINY_EDGES_PER_KIND, BIO_TINY_SIZES
    from graphqembed_amd.engine import Engine
    from graphqembed_amd.tensorize import FormulaPlan, pack_margin_batches
    from graphqembed_amd.trainer import TensorizedTrainer
    r, w, _, dist = parallel.init_from_env("gloo")
    d, dec, inter, B = 32, "bilinear-diag", "min", 64
    g = synth.bio_synth(seed=1, sizes=BIO_TINY_SIZES, edges_per_kind=BIO_TINY_EDGES_PER_KIND)
    layout = build_layout(g, d, dec, inter)
    eng = Engine(d, dec, inter, layout, max_queries=9 * B, max_batches=9, rank=r, world=w)
    init_params(eng, d, 0)
    types = ["1-chain", "2-chain", "3-chain", "2-inter", "3-inter", "3-inter_chain"]
    pools = synth.make_pools(g, types, formulas_per_type=3, pool_size=1000, seed=0)    # 1000 % 64 != 0: ragged slices

    class Shim(object):
        def __init__(self):
            self.plans, self.touched = {}, set()

        def margin_step(self, items):
            packed = []
            for (f, t, ng, a, wt, m) in items:
                if f not in self.plans:
                    self.plans[f] = FormulaPlan(f, layout, inter)
                packed.append((self.plans[f], t, ng, a, wt, m))
                self.touched |= self.plans[f].touched
            descs, idx, n = pack_margin_batches(packed)
            return eng.margin_fwd_bwd(descs, idx, n)

        def step(self):
            eng.adam_step(self.touched)
            self.touched = set()

    shim = Shim()
    all_rows = {m: np.arange(1, g.mode_sizes[m] + 1, dtype=np.int32) for m in g.modes}
    tr = TensorizedTrainer(shim, shim, pools, all_rows, batch_size=B, seed=0, dist=dist, rank=r, world=w, engine=eng)
    first = float(tr.run(10, log_every=0)[-1].item())
    last = float(tr.run(60, log_every=0)[-1].item())
    torch.cuda.synchronize()
    ref = eng.params.clone()
    dist.broadcast(ref, 0)
    same = torch.tensor([int(torch.equal(ref, eng.params))])
    dist.all_reduce(same, op=dist.ReduceOp.MIN)
    assert int(same.item()) == 1, "replicas diverged"
    assert np.isfinite(last) and last < first, (first, last)
    with open(os.path.join(out_dir, "ok%d" % rank), "w") as f:
        f.write("%r %r" % (first, last))
    dist.barrier()
    dist.destroy_process_group()
    eng.close()


def test_data_parallel_trainer_two_ranks(tmp_path):
    port = 29950 + os.getpid() % 40
    mp.spawn(_trainer_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(tmp_path / "ok0") and os.path.exists(tmp_path / "ok1")
