"""-m gpu: the data-parallel step on SEVERAL GPUs over RCCL — one process per GPU, torch.distributed's ``nccl`` backend for the
hand-driven phases and the library's OWN RCCL binding (ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd + ncclAllReduce on a
communicator of its own: parallel.RcclComm -> gqe_shard_open(nccl_comm) / gqe_allreduce_grads) for the one-call step.

These tests skip themselves below 2 (4, 8) visible GPUs: the build loop's boxes have one.  So that the first multi-GPU box runs a
parity check instead of a first contact, the SAME worker also runs on one GPU with 2 and 3 gloo ranks sharing it (transport =
callbacks over torch.distributed) — what differs between the two is the backend name, the device of each rank and the transport
``parallel.shard_session`` picks, nothing in the checks:

  * row-sharded step, phases driven by hand: the gradients that arrive at the owners + the all-reduced relation / Pre / Post
    gradients == the single-rank ORACLE gradient of the concatenated batch, shard by shard;
  * row-sharded step as one library call, three steps with unequal slices: global loss == the oracle's on the concatenated batch,
    replicated tensors bit-identical on every rank, the re-assembled shards == a single-rank engine stepped on the concatenated
    batch (up to the summation-order noise Adam amplifies);
  * replicated tables, sparse exchange (slab all-gather) and dense exchange (all-reduce of the flat gradient arena — north_star's
    form; through gqe_allreduce_grads on the RCCL path): gradient == the oracle's on the concatenated batch, replicas
    bit-identical after the optimiser step;
  * the native feeder on the row-sharded engine: a few iterations, replicated tensors still bit-identical, loss finite."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _worker(rank, world, port, out_dir, backend):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    local = rank if backend == "nccl" else 0
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(local), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    from graphqembed_amd import parallel
    from graphqembed_amd.engine import ArenaLayout, Engine
    from graphqembed_amd.tensorize import pack_margin_batches
    from gpu_utils import TOY_FORMULAS, TOY_KINDS, TOY_SIZES_ODD, engine_from_params, plan_for, random_params, read_arena, toy_batch
    from oracle import netquery_numpy as O
    r, w, _, dist = parallel.init_from_env(backend)
    assert dist.get_backend() == backend and (backend != "nccl" or torch.cuda.current_device() == rank)
    dev = torch.device("cuda", torch.cuda.current_device())
    cdev = dev if backend == "nccl" else torch.device("cpu")            # where a collective's tensors have to live

    def same_everywhere(t):
        mine = t.detach().to(cdev).contiguous()
        ref = mine.clone()
        dist.broadcast(ref, 0)
        ok = torch.tensor([int(torch.equal(mine, ref))], device=cdev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        return int(ok.item()) == 1

    def global_sum(x):
        t = torch.tensor([float(x)], dtype=torch.float64, device=cdev)
        dist.all_reduce(t)
        return float(t.item())

    rng = np.random.RandomState(77)
    d, dec, inter = 32, "bilinear-diag", "min"
    SIZES = TOY_SIZES_ODD                                               # 91 / 71 / 53 rows: W does not divide them
    params = random_params(rng, d, dec, inter, SIZES, TOY_KINDS)
    tables = [k for k in params if k.startswith("enc.")]

    def sharded_engine():
        layout = ArenaLayout()
        for k, v in params.items():
            layout.add(k, (parallel.shard_rows(v.shape[0], w), d) if k in tables else v.shape)
        eng = Engine(d, dec, inter, layout, shard=(r, w), max_queries=1024, max_batches=8)
        for k, v in params.items():
            src = parallel.shard_of(v, r, w) if k in tables else v
            layout.view(eng.params, k).copy_(torch.from_numpy(np.ascontiguousarray(src)))
        return eng

    def gather_full(eng):
        out = {}
        for k in params:
            mine = eng.layout.view(eng.params, k).to(cdev).contiguous()
            if k in tables:
                parts = [torch.zeros_like(mine) for _ in range(w)]
                dist.all_gather(parts, mine)
                full_t = torch.zeros(params[k].shape[0], d)
                for rr in range(w):
                    full_t[rr::w] = parts[rr][:len(full_t[rr::w])].cpu()
                out[k] = full_t.numpy()
            else:
                out[k] = mine.cpu().numpy().copy()
        return out

    mix = [("1-chain", 1.0), ("2-chain", 0.3), ("2-inter", 0.5), ("3-inter", 0.5), ("3-inter_chain", 0.5), ("3-chain_inter", 0.2)]
    n_pool, B = 400, 48

    def step_batches(step, cur=None, full=None):
        """This rank's items, the concatenated batch's items, and (with ``cur``) the oracle's loss / gradient on the latter."""
        items, cat_items, want = [], [], 0.0
        for qtype, wgt in mix:
            t, g, a = toy_batch(rng, qtype, n_pool, hub=(qtype == "2-inter" and step == 0), sizes=SIZES)
            s, e = parallel.rank_slice(n_pool, B, step + 7, r, w)        # later steps wrap around the pool: unequal slices
            cat = np.concatenate([np.arange(*parallel.rank_slice(n_pool, B, step + 7, rr, w)) for rr in range(w)])
            items.append((qtype, t[s:e], g[s:e], a[:, s:e], wgt * (e - s) / float(len(cat))))
            cat_items.append((qtype, t[cat], g[cat], a[:, cat], wgt))
            if cur is not None:
                l, _, _, _ = O.margin_fwd_bwd(cur, O.make_plan(qtype, TOY_FORMULAS[qtype]), dec, inter, t[cat], g[cat], a[:, cat], weight=wgt, grads=full)
                want += wgt * l
        return items, cat_items, want

    def pack(eng, items):
        packed = [(plan_for(eng, q, TOY_FORMULAS[q]), t, g, a, wgt, 1.0) for (q, t, g, a, wgt) in items]
        descs, idx, n_sc = pack_margin_batches(packed)
        return descs, idx, n_sc, set().union(*[p[0].touched for p in packed])

    # ---- 1. the phases by hand: gradients at the owners == the oracle's gradient of the concatenated batch -----------------------
    hand = sharded_engine()
    full = O.zero_grads_like(params)
    items, cat_items, want_loss = step_batches(0, {k: v.astype(np.float64) for k, v in params.items()}, full)
    descs, idx, _, keys = pack(hand, items)
    ps = parallel.shard_prepare(hand, dist, descs, idx)
    parallel.shard_fetch(hand, dist, ps)
    hand.run_margin(ps)
    parallel.shard_exchange(hand, dist, ps)
    np.testing.assert_allclose(global_sum(ps["losses"][-1].item()), want_loss, rtol=2e-4)
    got = read_arena(hand, hand.grads)
    for k in params:
        want = parallel.shard_of(full[k], r, w) if k in tables else full[k]
        scale = max(1e-6, float(np.abs(full[k]).max()))
        np.testing.assert_allclose(got[k], want, rtol=0, atol=2e-4 * scale, err_msg="gradient at the owner: " + k)
    hand.close()

    # ---- 2. the step as ONE library call (RCCL path: the library's own ncclSend / ncclRecv groups + ncclAllReduce) ----------------
    one = sharded_engine()
    keep = parallel.shard_session(one, dist, r, w)
    assert (backend == "nccl") == isinstance(keep, parallel.RcclComm)
    single = engine_from_params(params, d, dec, inter)
    rng = np.random.RandomState(78)
    for step in range(3):
        cur = gather_full(one)
        items, cat_items, want_loss = step_batches(step, {k: v.astype(np.float64) for k, v in cur.items()}, O.zero_grads_like(params))
        descs, idx, _, keys = pack(one, items)
        p1 = one.prepare_shard(descs, idx, keys)
        one.shard_post(p1)
        losses = one.shard_step(p1, 0.01)
        assert getattr(keep, "error", None) is None, keep.error
        np.testing.assert_allclose(global_sum(losses[-1].item()), want_loss, rtol=2e-4, err_msg="step %d" % step)
        descs, idx, n_sc, keys1 = pack(single, cat_items)
        ref_losses, _, _ = single.margin_fwd_bwd(descs, idx, n_sc)
        single.adam_step(keys1, 0.01)
        if step == 0:
            np.testing.assert_allclose(float(ref_losses[-1].item()), want_loss, rtol=2e-4)
    torch.cuda.synchronize()
    for name in ("params", "exp_avg", "exp_avg_sq"):
        rep = torch.cat([getattr(one, name)[o:o + n] for o, n in one.dense_spans()])
        assert same_everywhere(rep), "replicated tensors diverged: " + name
    got, want = gather_full(one), read_arena(single, single.params)
    worst = frac = 0.0
    moved = 0
    for k in params:
        diff = np.abs(got[k] - want[k])
        worst, frac = max(worst, float(diff.max())), max(frac, float((diff > 1e-4).mean()))
        moved += int((np.abs(want[k] - params[k]) > 1e-3).sum())
    assert worst < 0.04 and frac < 0.02 and moved > 1000, (worst, frac, moved)

    # ---- 3. the native feeder on the row-sharded engine ----------------------------------------------------------------------------
    from graphqembed_amd import synth
    from graphqembed_amd.tensorize import FormulaPlan, table_key
    # (pools over the toy schema: one formula per type, rows drawn like toy_batch)
    class Pool(object):
        pass
    plist, prng = [], np.random.RandomState(5)
    from graphqembed_amd.graph import Formula
    for qtype, _ in mix:
        p = Pool()
        p.target, p.neg, anchors = toy_batch(prng, qtype, 300, sizes=SIZES)
        p.anchors, p.hard = anchors, (p.neg.copy() if "inter" in qtype else None)
        plist.append((FormulaPlan(Formula(qtype, TOY_FORMULAS[qtype]), one.layout, inter), p))
    rows_by_key = {table_key(m): np.arange(1, SIZES[m] + 1, dtype=np.int32) for m in SIZES}
    feeder = one.make_feeder(plist, rows_by_key, batch_size=32, seed=3)
    n_b = 1 + sum(2 if "inter" in q else 1 for q, _ in mix if q != "1-chain")
    l0 = global_sum(one.feeder_run(feeder, 0, 4)[n_b].item())
    l1 = global_sum(one.feeder_run(feeder, 4, 40)[n_b].item())
    torch.cuda.synchronize()
    assert getattr(keep, "error", None) is None, keep.error
    assert np.isfinite(l0) and np.isfinite(l1) and l1 < l0, (l0, l1)
    assert same_everywhere(torch.cat([one.params[o:o + n] for o, n in one.dense_spans()])), "replicated tensors diverged under the native feeder"
    one.feeder_destroy(feeder)
    one.close()
    if hasattr(keep, "close"):
        keep.close()

    # ---- 4. replicated tables: sparse (slab all-gather) and dense (all-reduce of the arena) exchange -------------------------------
    rng = np.random.RandomState(79)
    sparse = engine_from_params(params, d, dec, inter, rank=r, world=w)
    dense = engine_from_params(params, d, dec, inter)
    full = O.zero_grads_like(params)
    items, cat_items, want_loss = step_batches(1, {k: v.astype(np.float64) for k, v in params.items()}, full)
    slab = sum((2 + len(O.make_plan(q, TOY_FORMULAS[q])["anchor_modes"])) * B for q, _ in mix)
    sparse.exchange_reserve(slab)
    comm = parallel.RcclComm(r, w, dist, device=dev) if backend == "nccl" else None
    for eng in (sparse, dense):
        descs, idx, n_sc, keys = pack(eng, items)
        eng.margin_fwd_bwd(descs, idx, n_sc)
    parallel.exchange_sparse(sparse, dist)
    if comm is not None:
        dense.allreduce_grads(comm.handle)                              # the library's own ncclAllReduce (gqe_allreduce_grads)
    else:
        parallel.exchange_gradients(dense.grads, dist, engine=dense)
    g_dense = read_arena(dense, dense.grads)
    for k in params:
        scale = max(1e-6, float(np.abs(full[k]).max()))
        np.testing.assert_allclose(g_dense[k], full[k], rtol=0, atol=2e-4 * scale, err_msg="dense exchange vs oracle " + k)
    sparse.adam_step(keys, 0.01)
    dense.adam_step(keys, 0.01)
    torch.cuda.synchronize()
    assert same_everywhere(sparse.params) and same_everywhere(dense.params), "replicas diverged"
    diff = (sparse.params - dense.params).abs()
    assert float(diff.max()) < 0.011 and float((diff > 1e-4).float().mean()) < 0.02, (float(diff.max()), float((diff > 1e-4).float().mean()))
    if comm is not None:
        comm.close()
    with open(os.path.join(out_dir, "ok%d" % rank), "w") as f:
        f.write("ok")
    dist.barrier()
    dist.destroy_process_group()
    for e in (single, sparse, dense):
        e.close()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_data_parallel_step_over_rccl(tmp_path, world):
    """One process per GPU over RCCL; skipped when fewer than ``world`` GPUs are visible."""
    if torch.cuda.device_count() < world:
        pytest.skip("%d GPUs visible, %d needed" % (torch.cuda.device_count(), world))
    port = 29050 + os.getpid() % 40 + world
    mp.spawn(_worker, args=(world, port, str(tmp_path), "nccl"), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / ("ok%d" % k)) for k in range(world))


@pytest.mark.parametrize("world", [2, 3])
def test_the_same_worker_on_one_gpu_over_gloo(tmp_path, world):
    """The worker of the RCCL test with gloo ranks sharing cuda:0 (transport: callbacks over torch.distributed): its checks are
    exercised on every box, so that a multi-GPU box tests the RCCL transport and not the test."""
    port = 29100 + os.getpid() % 40 + world
    mp.spawn(_worker, args=(world, port, str(tmp_path), "gloo"), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / ("ok%d" % k)) for k in range(world))
