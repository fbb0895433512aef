"""-m gpu: row-sharded data parallelism ("owner computes", include/gqe.h gqe_set_shard) on 2 gloo ranks sharing cuda:0.

  * gqe_shard_plan == the numpy statement of the plan; fetched rows are the rows the index feed names;
  * the gradients that arrive at the owners (lists folded into the local dense gradient) + the all-reduced relation / Pre /
    Post gradients == the single-rank oracle gradient of the CONCATENATED batch, shard by shard;
  * after the fused Adam pass over the own shards, the shards equal the rows of a single-rank engine stepped on the
    concatenated batch (up to the summation-order noise Adam amplifies), and the replicated relation / Pre / Post tensors
    are BIT-identical on every rank;
  * forward scores on fetched rows equal the single-rank engine's;
  * state machine: a second margin call before the contributions were linked is refused.
"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _worker(rank, world, port, out_dir, dec, inter, d, bag_modes=(), odd=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from graphqembed_amd import parallel
    from graphqembed_amd.engine import ArenaLayout, Engine, GqeError
    from graphqembed_amd.tensorize import pack_forward_batches, pack_margin_batches
    from gpu_utils import TOY_FORMULAS, TOY_KINDS, TOY_SIZES, TOY_SIZES_ODD, engine_from_params, plan_for, random_params, read_arena, toy_batch
    from oracle import netquery_numpy as O
    r, w, _, dist = parallel.init_from_env("gloo")
    rng = np.random.RandomState(21)
    SIZES = TOY_SIZES_ODD if odd else TOY_SIZES      # odd: 91 / 71 / 53 table rows — ceil(rows / W) leaves a short last shard
    params = random_params(rng, d, dec, inter, SIZES, TOY_KINDS, bag_modes=bag_modes)
    bag_keys = [O.table_key(m) for m in bag_modes]                     # EmbeddingBag tables stay replicated
    bags = {O.table_key(m): csr for m, csr in params.get(O.BAGS_KEY, {}).items()}
    params_only = {k: v for k, v in params.items() if k != O.BAGS_KEY}
    tables = [k for k in params_only if k.startswith("enc.") and k not in bag_keys]    # the sharded ones

    def sharded_engine():
        layout = ArenaLayout()
        for k, v in params_only.items():
            layout.add(k, (parallel.shard_rows(v.shape[0], w), d) if k in tables else v.shape)
        eng = Engine(d, dec, inter, layout, shard=(r, w), max_queries=1024, max_batches=8, bags=bags)
        for k, v in params_only.items():
            src = parallel.shard_of(v, r, w) if k in tables else v
            layout.view(eng.params, k).copy_(torch.from_numpy(np.ascontiguousarray(src)))
        return eng

    def gather_full(eng):
        """The whole model a sharded engine's ranks hold together (tables re-interleaved), as numpy arrays."""
        out = {}
        if O.BAGS_KEY in params:
            out[O.BAGS_KEY] = params[O.BAGS_KEY]
        for k in params_only:
            mine = eng.layout.view(eng.params, k).cpu()
            if k in tables:
                parts = [torch.zeros_like(mine) for _ in range(w)]
                dist.all_gather(parts, mine)
                full_t = torch.zeros(params[k].shape[0], d)
                for rr in range(w):
                    full_t[rr::w] = parts[rr][:len(full_t[rr::w])]
                out[k] = full_t.numpy()
            else:
                out[k] = mine.numpy().copy()
        return out

    single = engine_from_params(params, d, dec, inter)                 # the reference: one rank, the concatenated batch
    runs = [sharded_engine()]
    mix = [("1-chain", 1.0), ("2-chain", 0.3), ("2-inter", 0.5), ("3-inter", 0.5), ("3-inter_chain", 0.5), ("3-chain_inter", 0.2)]
    n_pool, B = 400, 64
    for step in range(3):
        items, cat_items = [], []
        full = O.zero_grads_like(params)
        cur = gather_full(runs[0])                                     # what the sharded ranks hold together right now
        want_loss = 0.0
        for qtype, wgt in mix:
            t, g, a = toy_batch(rng, qtype, n_pool, hub=(qtype == "2-inter" and step == 0), sizes=SIZES)   # hub rows: long lists at one owner
            if qtype == "2-inter" and step == 0 and w > 2:              # ... and that owner is the LAST rank (its shard is the short one)
                nt = SIZES[O.make_plan(qtype, TOY_FORMULAS[qtype])["target_mode"]]
                t[: n_pool // 2] = max(x for x in range(1, nt + 1) if x % w == w - 1)
            s, e = parallel.rank_slice(n_pool, B, step + 5, r, w)      # step 6 wraps around the pool: unequal slices
            cat = np.concatenate([np.arange(*parallel.rank_slice(n_pool, B, step + 5, rr, w)) for rr in range(w)])
            items.append((qtype, t[s:e], g[s:e], a[:, s:e], wgt * (e - s) / float(len(cat))))
            cat_items.append((qtype, t[cat], g[cat], a[:, cat], wgt))
            l, _, _, _ = O.margin_fwd_bwd(cur, O.make_plan(qtype, TOY_FORMULAS[qtype]), dec, inter, t[cat], g[cat], a[:, cat], weight=wgt, grads=full)
            want_loss += wgt * l
        # ---- the reference engine on the concatenated batch ----
        packed = [(plan_for(single, q, TOY_FORMULAS[q]), t, g, a, wgt, 1.0) for (q, t, g, a, wgt) in cat_items]
        descs, idx, n_sc = pack_margin_batches(packed)
        ref_losses, _, _ = single.margin_fwd_bwd(descs, idx, n_sc)
        keys = set().union(*[p[0].touched for p in packed])
        # ---- the sharded engines ----
        shard_losses = []
        for eng in runs:
            packed = [(plan_for(eng, q, TOY_FORMULAS[q]), t, g, a, wgt, 1.0) for (q, t, g, a, wgt) in items]
            descs, idx, _ = pack_margin_batches(packed)
            ps = parallel.shard_prepare(eng, dist, descs, idx)
            if eng is runs[0]:                                         # the library's plan == the numpy statement of it
                pos, req, cnt = eng.shard_plan(descs, idx)
                hb, tid, run = {}, [], 0
                for k in eng.layout.entries:
                    if k in tables or k in bag_keys:
                        hb[eng.layout.offset(k)] = (len(hb), run)
                        run += eng.layout.entries[k][1][0]
                bag_ids = [hb[eng.layout.offset(k)][0] for k in bag_keys]
                for dsc in descs:
                    tid += [hb[dsc["target_table"]][0]] * (2 * dsc["n"])
                    for at in dsc["anchor_table"]:
                        tid += [hb[at][0]] * dsc["n"]
                base = [v[1] for v in sorted(hb.values())]
                p2, r2, c2 = parallel.shard_plan_numpy(idx, tid, base, w, bag_tables=bag_ids)
                assert np.array_equal(pos, p2) and np.array_equal(req[:len(r2)], r2) and np.array_equal(cnt, c2)
                assert ps["n_send"] == len(r2) == int(np.sum(~np.isin(tid, bag_ids)))
            parallel.shard_fetch(eng, dist, ps)
            if eng is runs[0]:                                         # fetched rows = the rows the feed names
                torch.cuda.synchronize()
                fetched = eng.shard_views()["fetched"][:ps["n_send"]].cpu().numpy()
                off = 0
                cur32 = cur
                for (q, t, g, a, wgt), dsc in zip(items, descs):
                    plan = O.make_plan(q, TOY_FORMULAS[q])
                    segs = [(plan["target_mode"], t), (plan["target_mode"], g)] + [(m, a[i]) for i, m in enumerate(plan["anchor_modes"])]
                    for mode, rows in segs:
                        feed = ps["idx"][off:off + len(rows)].cpu().numpy()
                        if mode in bag_modes:                          # bag ids pass through: nothing is fetched for them
                            assert np.array_equal(feed, rows), (step, q, mode)
                        else:
                            assert np.array_equal(fetched[feed], cur32[O.table_key(mode)][rows]), (step, q, mode)
                        off += len(rows)
            eng.run_margin(ps)
            with pytest.raises(GqeError):                              # one margin call per step in this mode
                eng.run_margin(ps)
            parallel.shard_exchange(eng, dist, ps)
            shard_losses.append(ps["losses"].clone())
            if eng is runs[0]:                                         # gradients at the owners == the oracle's, shard by shard
                got = read_arena(eng, eng.grads)                       # (folds the lists into the local dense gradient)
                for k in params_only:
                    want = parallel.shard_of(full[k], r, w) if k in tables else full[k]      # bag tables: the all-reduced whole
                    scale = max(1e-6, float(np.abs(full[k]).max()))
                    np.testing.assert_allclose(got[k], want, rtol=0, atol=2e-4 * scale, err_msg="step %d %s" % (step, k))
            eng.adam_step(keys, 0.01)
        single.adam_step(keys, 0.01)
        torch.cuda.synchronize()
        # per-rank mean losses recombine to the concatenated batch's: sum_r (n_r / n) * mean_r (the weights carry n_r / n)
        tot = shard_losses[0][-1:].clone().cpu()
        dist.all_reduce(tot)
        np.testing.assert_allclose(float(tot.item()), want_loss, rtol=2e-4)
        if step == 0:                                                  # identical parameters so far: the reference engine agrees too
            np.testing.assert_allclose(float(tot.item()), float(ref_losses[-1].item()), rtol=2e-4)
    # ---- the shards after three steps ----
    a0 = read_arena(runs[0], runs[0].params)
    want = read_arena(single, single.params)
    for name in ("params", "exp_avg", "exp_avg_sq"):                   # replicated tensors (and bag tables): the same bits on every rank
        mine = torch.cat([getattr(runs[0], name)[o:o + n] for o, n in runs[0].dense_spans()] +
                         [runs[0].layout.view(getattr(runs[0], name), k).reshape(-1) for k in bag_keys]).cpu()
        ref = mine.clone()
        dist.broadcast(ref, 0)
        assert torch.equal(mine, ref), "replicated tensors diverged: " + name
    worst, frac = 0.0, 0.0
    for k in params_only:
        w_k = parallel.shard_of(want[k], r, w) if k in tables else want[k]
        diff = np.abs(a0[k] - w_k)
        worst = max(worst, float(diff.max()))
        frac = max(frac, float((diff > 1e-4).mean()))
    # Adam turns summation-order noise of near-zero gradients into lr-sized steps (tests/test_oracle_golden.py::
    # test_adam_three_steps): everything else agrees to fp32 rounding
    assert worst < 0.04 and frac < 0.02, (worst, frac)
    # ---- forward on fetched rows == the single-rank forward (same kernel, same row values) ----
    sync = {k: torch.from_numpy(v) for k, v in gather_full(runs[0]).items() if k != O.BAGS_KEY}
    for k, v in sync.items():
        single.layout.view(single.params, k).copy_(v)
    for qtype in ("2-chain", "3-inter", "3-chain_inter"):
        t, g, a = toy_batch(rng, qtype, 40 + 7 * r, sizes=SIZES)
        descs, idx, n = pack_forward_batches([(plan_for(runs[0], qtype, TOY_FORMULAS[qtype]), t, a)])
        ps = parallel.shard_prepare(runs[0], dist, descs, idx, with_negatives=False)
        got = parallel.shard_forward(runs[0], dist, ps, n)
        descs, idx, n = pack_forward_batches([(plan_for(single, qtype, TOY_FORMULAS[qtype]), t, a)])
        ref = single.forward(descs, idx, n)
        assert torch.equal(got, ref), qtype
    with open(os.path.join(out_dir, "ok%d" % rank), "w") as f:
        f.write("ok")
    dist.barrier()
    dist.destroy_process_group()
    for e in runs + [single]:
        e.close()


@pytest.mark.parametrize("dec,inter,d,bag_modes", [("bilinear-diag", "min", 32, ()), ("bilinear", "mean", 32, ()), ("transe", "min-simple", 64, ()),
                                                   ("bilinear-diag", "min", 64, ("b",))])
def test_row_sharded_two_ranks(tmp_path, dec, inter, d, bag_modes):
    """bag_modes=("b",): mode b is an EmbeddingBag mode (Reddit posts) — its word table stays replicated, its indices pass
    through the plan as bag ids, its gradient is folded into the dense arena and all-reduced."""
    port = 29400 + os.getpid() % 150
    mp.spawn(_worker, args=(2, port, str(tmp_path), dec, inter, d, bag_modes), nprocs=2, join=True)
    assert os.path.exists(tmp_path / "ok0") and os.path.exists(tmp_path / "ok1")


@pytest.mark.parametrize("world,dec,inter,d,bag_modes", [(3, "bilinear-diag", "min", 32, ()), (4, "bilinear-diag", "min", 32, ()),
                                                         (3, "bilinear-diag", "min", 64, ("b",)), (4, "bilinear", "mean", 32, ())])
def test_row_sharded_more_ranks_with_remainders(tmp_path, world, dec, inter, d, bag_modes):
    """The same protocol on 3 and 4 ranks sharing the GPU, with tables of 91 / 71 / 53 rows: ceil(rows / W) does not divide,
    the last ranks' shards are short (their padding rows must never be named, served or stepped into the result), the owner
    counting sort has W buckets, a hub row lives on the LAST rank."""
    port = 29700 + os.getpid() % 150
    mp.spawn(_worker, args=(world, port, str(tmp_path), dec, inter, d, bag_modes, True), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / ("ok%d" % k)) for k in range(world))


def _trainer_worker(rank, world, port, out_dir):
    """TensorizedTrainer on row-sharded tables, 2 ranks: same formula draws, own query slices and negatives, rows fetched
    from / contributions sent to their owners every iteration -> the loss falls, the replicated tensors stay bit-identical,
    and the shards re-assemble to a model whose held-in scores improved."""
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from bench import build_layout, init_params
    from graphqembed_amd import parallel, synth
    from graphqembed_amd.data_utils import BIO_TINY_EDGES_PER_KIND, BIO_TINY_SIZES
    from graphqembed_amd.engine import Engine
    from graphqembed_amd.tensorize import FormulaPlan
    from graphqembed_amd.trainer import TensorizedTrainer
    r, w, _, dist = parallel.init_from_env("gloo")
    d, dec, inter, B = 32, "bilinear-diag", "min", 64
    g = synth.bio_synth(seed=1, sizes=BIO_TINY_SIZES, edges_per_kind=BIO_TINY_EDGES_PER_KIND)
    layout = build_layout(g, d, dec, inter, shard_world=w)
    eng = Engine(d, dec, inter, layout, max_queries=9 * B, max_batches=9, shard=(r, w))
    init_params(eng, d, 100 + r)                                   # shards: whatever; replicated tensors: equal
    gen = torch.Generator(device=eng.device)
    gen.manual_seed(0)
    for off, n in eng.dense_spans():
        eng.params[off:off + n].uniform_(-0.3, 0.3, generator=gen)
    types = ["1-chain", "2-chain", "3-chain", "2-inter", "3-inter", "3-inter_chain"]
    pools = synth.make_pools(g, types, formulas_per_type=3, pool_size=1000, seed=0)    # 1000 % 64 != 0: ragged slices

    class Shim(object):                       # plans + the optimiser on the bare engine
        lr, betas, eps = 0.01, (0.9, 0.999), 1e-8   # the trainer reads the hyper-parameters; gqe_shard_step does the stepping

        def __init__(self):
            self.plans, self.touched = {}, set()

        def plan(self, f):
            if f not in self.plans:
                self.plans[f] = FormulaPlan(f, layout, inter)
            return self.plans[f]

        def step(self):
            eng.adam_step(self.touched)
            self.touched = set()

    shim = Shim()
    all_rows = {m: np.arange(1, g.mode_sizes[m] + 1, dtype=np.int32) for m in g.modes}
    tr = TensorizedTrainer(shim, shim, pools, all_rows, batch_size=B, seed=0, dist=dist, rank=r, world=w, engine=eng, plan_of=shim.plan)
    assert tr.sharded

    def global_loss(losses):                   # the weights carry n_rank / n_all: the ranks' totals add up to the batch loss
        t = losses[-1:].clone().cpu()
        dist.all_reduce(t)
        return float(t.item())
    first = global_loss(tr.run(10, log_every=0))
    last = global_loss(tr.run(80, log_every=0))
    torch.cuda.synchronize()
    mine = torch.cat([eng.params[o:o + n] for o, n in eng.dense_spans()]).cpu()
    ref = mine.clone()
    dist.broadcast(ref, 0)
    assert torch.equal(mine, ref), "replicated tensors diverged"
    assert np.isfinite(last) and last < 0.9 * first, (first, last)
    # ---- the native feeder on the same row-sharded engine (gqe_feeder_run: sampling, packing, gqe_shard_post one iteration
    # ahead and gqe_shard_step from C++): per-rank slices it * W + rank of shared formula draws, loss weights n_rank / n_all
    from graphqembed_amd.tensorize import table_key
    plist = [(shim.plan(p.formula), p) for t in types for p in pools[t]]
    rows_by_key = {table_key(m): np.arange(1, g.mode_sizes[m] + 1, dtype=np.int32) for m in g.modes}
    feeder = eng.make_feeder(plist, rows_by_key, batch_size=B, seed=3)
    n_b = 1 + sum(2 if "inter" in t else 1 for t in types if t != "1-chain")
    l0 = global_loss(eng.feeder_run(feeder, 0, 5)[:n_b + 1])
    l1 = global_loss(eng.feeder_run(feeder, 5, 120)[:n_b + 1])
    torch.cuda.synchronize()
    assert getattr(tr._session, "error", None) is None, tr._session.error
    mine = torch.cat([eng.params[o:o + n] for o, n in eng.dense_spans()]).cpu()
    ref = mine.clone()
    dist.broadcast(ref, 0)
    assert torch.equal(mine, ref), "replicated tensors diverged under the native feeder"
    assert np.isfinite(l1) and l1 < l0, (l0, l1)
    eng.feeder_destroy(feeder)
    with open(os.path.join(out_dir, "tr_ok%d" % rank), "w") as f:
        f.write("%r %r" % (first, last))
    dist.barrier()
    dist.destroy_process_group()
    eng.close()


def test_row_sharded_trainer_two_ranks(tmp_path):
    port = 29250 + os.getpid() % 40
    mp.spawn(_trainer_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(tmp_path / "tr_ok0") and os.path.exists(tmp_path / "tr_ok1")


def _session_worker(rank, world, port, out_dir, dec, inter, d, odd=False):
    """gqe_shard_open / post / step (the row-sharded step as one library call, planning through the shared-memory plan board)
    on 2 gloo ranks sharing cuda:0, the transport being callbacks over torch.distributed:
      * == the phases driven by hand (plan / serve / link + Python collectives) BIT FOR BIT on shards, moments and the
        replicated tensors after three steps with unequal slices (same kernels, order-independent list sums);
      * lazy Adam on sharded tables == eager, bit for bit, once synchronised (rows owe deferred steps in between);
      * ranks that run DIFFERENT formulas in a step: the touched-tensor sets travel with the plans, the optimiser steps the
        union, the replicated tensors stay bit-identical, and the result matches a single-rank engine on the union batch;
      * forward through the session == the single-rank forward bit for bit."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from graphqembed_amd import parallel
    from graphqembed_amd.engine import ArenaLayout, Engine
    from graphqembed_amd.tensorize import pack_forward_batches, pack_margin_batches
    from gpu_utils import TOY_FORMULAS, TOY_KINDS, TOY_SIZES, TOY_SIZES_ODD, engine_from_params, plan_for, random_params, read_arena, toy_batch
    r, w, _, dist = parallel.init_from_env("gloo")
    rng = np.random.RandomState(33)
    SIZES = TOY_SIZES_ODD if odd else TOY_SIZES
    params = random_params(rng, d, dec, inter, SIZES, TOY_KINDS)
    tables = [k for k in params if k.startswith("enc.")]

    def sharded_engine(lazy=False):
        layout = ArenaLayout()
        for k, v in params.items():
            layout.add(k, (parallel.shard_rows(v.shape[0], w), d) if k in tables else v.shape)
        eng = Engine(d, dec, inter, layout, shard=(r, w), max_queries=1024, max_batches=8, lazy_adam=lazy,
                     ordered_sums=True)       # bit-reproducible list sums: the engines below are compared bit for bit
        for k, v in params.items():
            src = parallel.shard_of(v, r, w) if k in tables else v
            layout.view(eng._params, k).copy_(torch.from_numpy(np.ascontiguousarray(src)))
        return eng

    def gather_full(eng):
        out = {}
        for k in params:
            mine = eng.layout.view(eng.params, k).cpu()
            if k in tables:
                parts = [torch.zeros_like(mine) for _ in range(w)]
                dist.all_gather(parts, mine)
                full_t = torch.zeros(params[k].shape[0], d)
                for rr in range(w):
                    full_t[rr::w] = parts[rr][:len(full_t[rr::w])]
                out[k] = full_t.numpy()
            else:
                out[k] = mine.numpy().copy()
        return out

    by_hand, session, lazy = sharded_engine(), sharded_engine(), sharded_engine(lazy=True)
    # (session: the transport leaves the own block alone and the library keeps it in place, as on the RCCL path; lazy: the
    # transport moves every block)
    keep = [parallel.shard_session(session, dist, r, w), parallel.shard_session(lazy, dist, r, w, skip_own=False)]
    # One query type per step and <= 16 queries per rank (one tile): every floating-point reduction is then order-free (two
    # branch gradients into Pre, two ranks into the all-reduce, row lists summed order-independently) and two engines agree
    # BIT FOR BIT — or differ because of the protocol.  No 3-inter: its three branches add into the Pre gradient in atomic order.
    types = ["1-chain", "2-inter", "2-chain", "3-inter_chain", "3-chain", "3-chain_inter", "2-inter", "1-chain", "3-inter_chain"]
    sizes = [12, 9, 7, 5][:w]                                          # unequal slices
    for step, qtype in enumerate(types):
        t, g, a = toy_batch(rng, qtype, sum(sizes), hub=(step == 1), sizes=SIZES)   # hub rows: long lists at one owner
        hi = 25 + 12 * step                                            # most rows untouched at first: they lag in lazy mode
        t[:], g[:] = np.minimum(t, hi), np.minimum(g, hi)
        s0 = sum(sizes[:r])
        sl = slice(s0, s0 + sizes[r])
        items = [(qtype, t[sl], g[sl], a[:, sl], sizes[r] / float(sum(sizes)))]
        # the phases by hand
        packed = [(plan_for(by_hand, q, TOY_FORMULAS[q]), tt, gg, aa, wgt, 1.0) for (q, tt, gg, aa, wgt) in items]
        descs, idx, _ = pack_margin_batches(packed)
        keys = set().union(*[p[0].touched for p in packed])
        ps = parallel.shard_prepare(by_hand, dist, descs, idx)
        parallel.shard_margin_step(by_hand, dist, ps, adam=by_hand.prepare_adam(keys), lr=0.01)
        # one call (eager and lazy)
        for eng in (session, lazy):
            packed = [(plan_for(eng, q, TOY_FORMULAS[q]), tt, gg, aa, wgt, 1.0) for (q, tt, gg, aa, wgt) in items]
            descs, idx, _ = pack_margin_batches(packed)
            p1 = eng.prepare_shard(descs, idx, keys)
            eng.shard_post(p1)
            losses = eng.shard_step(p1, 0.01)
            assert torch.equal(losses, ps["losses"]), (step, eng is lazy)
        for k in keep:
            assert getattr(k, "error", None) is None, k.error
    torch.cuda.synchronize()
    assert not torch.equal(lazy._params, session._params)       # rows of the lazy engine do owe steps ...
    for name in ("params", "exp_avg", "exp_avg_sq"):            # ... and agree bit for bit once settled (the property syncs)
        assert torch.equal(getattr(session, name), getattr(by_hand, name)), "one call != phases: " + name
        assert torch.equal(getattr(lazy, name), getattr(session, name)), "lazy != eager: " + name
    # ---- different formulas on the two ranks (fresh engines: first Adam step on both sides) ----
    het = sharded_engine()
    keep.append(parallel.shard_session(het, dist, r, w))
    single = engine_from_params(params, d, dec, inter)
    # (with more than two ranks some rank names no row of table c at all: its requests to c's owners are empty)
    mixes = ([("2-chain", 0.5), ("1-chain", 1.0)], [("3-inter", 0.5), ("1-chain", 1.0)], [("1-chain", 1.0)], [("2-inter", 0.5), ("3-chain", 0.2)])[:w]
    per_rank = [[(q,) + toy_batch(np.random.RandomState(100 + 10 * rr + j), q, 40, sizes=SIZES) + (wgt,) for j, (q, wgt) in enumerate(lst)]
                for rr, lst in enumerate(mixes)]
    packed = [(plan_for(het, q, TOY_FORMULAS[q]), t, g, a, wgt, 1.0) for (q, t, g, a, wgt) in per_rank[r]]
    descs, idx, _ = pack_margin_batches(packed)
    p1 = het.prepare_shard(descs, idx, set().union(*[p[0].touched for p in packed]))
    het.shard_post(p1)
    het.shard_step(p1, 0.01)
    union = [x for lst in per_rank for x in lst]
    packed = [(plan_for(single, q, TOY_FORMULAS[q]), t, g, a, wgt, 1.0) for (q, t, g, a, wgt) in union]
    descs, idx, n_sc = pack_margin_batches(packed)
    single.margin_fwd_bwd(descs, idx, n_sc)
    single.adam_step(set().union(*[p[0].touched for p in packed]), 0.01)
    torch.cuda.synchronize()
    rep = torch.cat([het.params[o:o + n] for o, n in het.dense_spans()]).cpu()
    ref = rep.clone()
    dist.broadcast(ref, 0)
    assert torch.equal(rep, ref), "replicated tensors diverged when the ranks ran different formulas"
    got, want = gather_full(het), read_arena(single, single.params)
    changed = 0
    for k in params:
        diff = np.abs(got[k] - want[k])
        assert float(diff.max()) < 0.04 and float((diff > 1e-4).mean()) < 0.02, (k, float(diff.max()), float((diff > 1e-4).mean()))
        changed += int((np.abs(want[k] - params[k]) > 1e-3).sum())
    assert changed > 1000                              # the step did move the model: the comparison is not vacuous
    het.close()
    got = gather_full(session)
    # ---- forward through the session ----
    for k, v in got.items():
        single.layout.view(single.params, k).copy_(torch.from_numpy(v))
    for qtype in ("2-chain", "3-inter"):
        t, g, a = toy_batch(rng, qtype, 30 + 5 * r, sizes=SIZES)
        descs, idx, n = pack_forward_batches([(plan_for(session, qtype, TOY_FORMULAS[qtype]), t, a)])
        pf = session.prepare_shard(descs, idx, with_negatives=False)
        session.shard_post(pf)
        sc = session.shard_forward(n)
        descs, idx, n = pack_forward_batches([(plan_for(single, qtype, TOY_FORMULAS[qtype]), t, a)])
        assert torch.equal(sc, single.forward(descs, idx, n)), qtype
    # ---- candidate lists through the session (fused evaluation on row-sharded tables): the candidates are fetched from
    # their owners like any other row; ragged lists (one of them empty), a different number of queries per rank ----
    from graphqembed_amd.tensorize import pack_candidate_batches
    import oracle.netquery_numpy as O
    for qtype in ("1-chain", "3-chain", "2-inter", "3-inter_chain"):
        nq = 9 + 4 * r
        t, g, a = toy_batch(rng, qtype, nq, sizes=SIZES)
        mode = O.make_plan(qtype, TOY_FORMULAS[qtype])["target_mode"]
        lens = rng.randint(1, 40, nq)
        lens[nq // 2] = 0
        ptr = np.zeros(nq + 1, dtype=np.int32)
        ptr[1:] = np.cumsum(lens)
        rows = rng.randint(1, SIZES[mode] + 1, int(ptr[-1])).astype(np.int32)
        descs, idx, n = pack_candidate_batches([(plan_for(session, qtype, TOY_FORMULAS[qtype]), a, ptr, rows)])
        pf = session.prepare_shard(descs, idx, with_negatives=False)
        session.shard_post(pf)
        sc = session.shard_forward(n)
        descs, idx, n = pack_candidate_batches([(plan_for(single, qtype, TOY_FORMULAS[qtype]), a, ptr, rows)])
        assert torch.equal(sc, single.forward(descs, idx, n)), "candidate lists, " + qtype
    with open(os.path.join(out_dir, "s_ok%d" % rank), "w") as f:
        f.write("ok")
    dist.barrier()
    dist.destroy_process_group()
    for e in (by_hand, session, lazy, single):
        e.close()


@pytest.mark.parametrize("dec,inter,d", [("bilinear-diag", "min", 32), ("bilinear", "mean", 32), ("transe", "min-simple", 64)])
def test_shard_step_session_two_ranks(tmp_path, dec, inter, d):
    port = 29300 + os.getpid() % 90
    mp.spawn(_session_worker, args=(2, port, str(tmp_path), dec, inter, d), nprocs=2, join=True)
    assert os.path.exists(tmp_path / "s_ok0") and os.path.exists(tmp_path / "s_ok1")


@pytest.mark.parametrize("world,dec,inter,d", [(3, "bilinear-diag", "min", 32), (4, "bilinear-diag", "min", 32), (4, "transe", "min-simple", 64)])
def test_shard_step_session_more_ranks_with_remainders(tmp_path, world, dec, inter, d):
    """gqe_shard_open / post / step on 3 and 4 ranks (plan board with W slots, W-way owner sort, W blocks per all-to-all) over
    tables whose row counts W does not divide: one call == the phases by hand bit for bit, lazy == eager, ranks with different
    formulas (one of them names no row of a table), forward and candidate lists through the session."""
    port = 29850 + os.getpid() % 90
    mp.spawn(_session_worker, args=(world, port, str(tmp_path), dec, inter, d, True), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / ("s_ok%d" % k)) for k in range(world))
