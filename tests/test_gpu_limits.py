"""-m gpu: capacities and state corners of libgqe that the BASELINE workloads do not reach by themselves.

  * optimiser passes over more tensors / more distinct per-tensor Adam step counts than the kernel-argument form
    holds (a Bio-scale schema: dozens of relation types, each with its own step counter);
  * the device formula-descriptor cache re-uses least-recently-used slots (the reference draws a Formula per
    batch from thousands of distinct ones, train_helpers.py:96-100);
  * lazy Adam: gqe_optimizer_sync must not consume a materialised dense gradient; a pending margin call's staged
    index feed that later host-fed forwards overwrite makes the step fall back to the full pass;
  * nn.Module.load_state_dict settles deferred steps first.
"""
import numpy as np
import pytest

from oracle import netquery_numpy as O

pytestmark = pytest.mark.gpu


def test_optimizer_pass_over_many_tensors_with_diverging_step_counts():
    """130 relation vectors + 2 tables, every tensor at its own Adam step count (> 96 tensors and > 32 distinct
    counts: the pass is described by an uploaded list, not by kernel arguments), three passes over changing
    subsets, against the fp64 restatement of torch.optim.Adam."""
    import torch
    from gpu_utils import engine_from_params, read_arena
    rng = np.random.RandomState(3)
    d = 16
    params = {"enc.feat-a.weight": rng.randn(301, d).astype(np.float32),
              "enc.feat-b.weight": rng.randn(77, d).astype(np.float32)}
    for k in range(130):
        params["path_dec.a_r%d_b" % k] = rng.randn(d).astype(np.float32)
    eng = engine_from_params(params, d, "bilinear-diag", "min-simple")
    ref = {k: v.astype(np.float64) for k, v in params.items()}
    state = {}
    names = list(params)
    for k_i, k in enumerate(names):                      # diverged counters, as after a long run on a skewed type mix
        state[k] = {"step": 3 * k_i + (k_i % 7), "m": rng.randn(*params[k].shape) * 0.01,
                    "v": np.abs(rng.randn(*params[k].shape)) * 1e-3}
        eng.steps[k] = state[k]["step"]
        eng.layout.view(eng.exp_avg, k).copy_(torch.from_numpy(state[k]["m"].astype(np.float32)))
        eng.layout.view(eng.exp_avg_sq, k).copy_(torch.from_numpy(state[k]["v"].astype(np.float32)))
        state[k]["m"] = state[k]["m"].astype(np.float32).astype(np.float64)
        state[k]["v"] = state[k]["v"].astype(np.float32).astype(np.float64)
    for step in range(3):
        keys = [k for i, k in enumerate(names) if step == 0 or (i + step) % 3]
        grads = {}
        eng.materialize()                                 # the dense gradient is authoritative: written by hand below
        for k in keys:
            g = (rng.randn(*params[k].shape) * 10 ** rng.uniform(-4, 0, size=params[k].shape)).astype(np.float32)
            grads[k] = g
            eng.layout.view(eng.grads, k).copy_(torch.from_numpy(g))
        eng.adam_step(keys)
        O.adam_step(ref, {k: v.astype(np.float64) for k, v in grads.items()}, state, keys)
        got = read_arena(eng, eng.params)
        for k in names:
            np.testing.assert_allclose(got[k], ref[k], rtol=0, atol=3e-6, err_msg="%s pass %d" % (k, step))
        assert float(eng.grads.abs().max()) == 0.0
    # the kernel-argument form and the list form are the same arithmetic: a second engine that only ever sees 40
    # tensors at 3 distinct counts (argument form) must agree bit for bit on those tensors
    small = {k: params[k] for k in names[:40]}
    e1 = engine_from_params(params, d, "bilinear-diag", "min-simple")
    e2 = engine_from_params(small, d, "bilinear-diag", "min-simple")
    for e in (e1, e2):
        e.materialize()
    for i, k in enumerate(names):
        g = torch.from_numpy((rng.randn(*params[k].shape)).astype(np.float32))
        e1.steps[k] = 100 + i                              # 132 distinct counts -> list form
        e1.layout.view(e1.grads, k).copy_(g)
        if k in small:
            e2.steps[k] = 100 + i
            e2.layout.view(e2.grads, k).copy_(g)
    e1.adam_step(names)
    e2.adam_step(list(small))
    a, b = read_arena(e1, e1.params), read_arena(e2, e2.params)
    for k in small:
        assert np.array_equal(a[k], b[k]), k
    for e in (eng, e1, e2):
        e.close()


def test_more_tensors_than_the_declared_limit_is_an_error():
    from gpu_utils import engine_from_params
    from graphqembed_amd.engine import GqeError
    import ctypes as C
    from graphqembed_amd.engine import gqe_segment
    rng = np.random.RandomState(0)
    params = {"enc.feat-a.weight": rng.randn(9, 16).astype(np.float32), "path_dec.a_r_a": rng.randn(16).astype(np.float32)}
    eng = engine_from_params(params, 16, "bilinear-diag", "min-simple")     # limit = 2 tensors
    arr = (gqe_segment * 3)()
    for i, (off, n) in enumerate([(0, 9 * 16), (eng.layout.offset("path_dec.a_r_a"), 16), (4, 4)]):
        arr[i].offset, arr[i].numel, arr[i].step = off, n, 1
    rc = eng.lib.gqe_adam_step(eng.ctx, arr, 3, 0.01, 0.9, 0.999, 1e-8, eng._stream())
    assert rc != 0 and b"gqe_set_limits" in eng.lib.gqe_last_error(eng.ctx)
    eng.close()


def test_formula_cache_replaces_least_recently_used_descriptors():
    """200 distinct formulas through a 64-slot descriptor cache, revisited in two different orders, 5 per call:
    scores and gradients equal those of an engine whose cache holds them all."""
    import torch
    from gpu_utils import engine_from_params, random_params, read_arena
    from graphqembed_amd.graph import Formula
    from graphqembed_amd.tensorize import FormulaPlan, pack_forward_batches, pack_margin_batches
    rng = np.random.RandomState(9)
    d = 32
    sizes = {"a": 40, "b": 30}
    kinds = tuple(("a", "r%d" % k, "b") for k in range(12)) + (("a", "s", "a"),)
    params = random_params(rng, d, "bilinear-diag", "min", sizes, kinds)
    small = engine_from_params(params, d, "bilinear-diag", "min", max_formulas=64)
    big = engine_from_params(params, d, "bilinear-diag", "min")
    formulas = []
    for i in range(12):
        for j in range(12):
            formulas.append(Formula("2-chain", (("a", "r%d" % i, "b"), ("b", "r%d" % j, "a"))))
            if i < j:
                formulas.append(Formula("2-inter", (("a", "r%d" % i, "b"), ("a", "r%d" % j, "b"))))
    assert len(formulas) > 3 * 64
    B = 5
    for order in (np.arange(len(formulas)), rng.permutation(len(formulas)), rng.permutation(len(formulas))):
        for c0 in range(0, len(order), 5):
            fs = [formulas[k] for k in order[c0:c0 + 5]]
            feeds = []
            for f in fs:
                na = len(f.anchor_modes)
                t = rng.randint(1, sizes["a"] + 1, B).astype(np.int32)
                g = rng.randint(1, sizes["a"] + 1, B).astype(np.int32)
                a = np.stack([rng.randint(1, sizes[m] + 1, B) for m in f.anchor_modes]).astype(np.int32)
                feeds.append((f, t, g, a))
            outs = []
            for eng in (small, big):
                fw = [(FormulaPlan(f, eng.layout, "min"), t, a) for (f, t, g, a) in feeds]
                descs, idx, n = pack_forward_batches(fw)
                sc = eng.forward(descs, idx, n).clone()
                mg = [(FormulaPlan(f, eng.layout, "min"), t, g, a, 1.0, 1.0) for (f, t, g, a) in feeds]
                descs, idx, n = pack_margin_batches(mg)
                losses, pos, neg = eng.margin_fwd_bwd(descs, idx, n, want_scores=True)
                outs.append((sc, losses.clone(), pos.clone(), neg.clone()))
                eng.zero_grads(list(eng.layout.entries))
            for x, y in zip(*outs):
                assert torch.allclose(x, y, rtol=1e-6, atol=1e-7)
    # a call may name as many distinct formulas as the cache holds (GQE_MAX_BATCHES = 64 batches per call)
    tiny = [(FormulaPlan(f, small.layout, "min"), np.ones(1, np.int32), np.ones((len(f.anchor_modes), 1), np.int32)) for f in formulas[:64]]
    descs, idx, n = pack_forward_batches(tiny)
    small.forward(descs, idx, n)
    for eng in (small, big):
        eng.close()


def test_lazy_sync_keeps_a_materialised_dense_gradient():
    """Lazy Adam, rows lagging (a sparse step), then backward + gqe_materialize_grads (torch.optim compatibility,
    Engine.reserve) + gqe_optimizer_sync, then the step: the synchronisation only replays deferred steps, the
    gradient is still there for the step — parameters equal the eager engine's bit for bit."""
    import torch
    from gpu_utils import TOY_FORMULAS, TOY_KINDS, TOY_SIZES, engine_from_params, plan_for, random_params
    from graphqembed_amd.tensorize import pack_margin_batches
    from test_gpu_parity import _disjoint_batch
    rng = np.random.RandomState(4)
    d = 32
    params = random_params(rng, d, "bilinear-diag", "min-simple", TOY_SIZES, TOY_KINDS)
    lazy = engine_from_params(params, d, "bilinear-diag", "min-simple", lazy_adam=True)
    eager = engine_from_params(params, d, "bilinear-diag", "min-simple")
    pl = {e: plan_for(e, "2-inter", TOY_FORMULAS["2-inter"]) for e in (lazy, eager)}
    for step in range(4):
        t, g, a = _disjoint_batch(rng, "2-inter", 8, 0.0, 0.5)
        for e in (lazy, eager):
            descs, idx, n = pack_margin_batches([(pl[e], t, g, a, 1.0, 1.0)])
            e.margin_fwd_bwd(descs, idx, n)
            e.adam_step(pl[e].touched)
    assert not torch.equal(lazy._params, eager._params)            # rows do lag
    t, g, a = _disjoint_batch(rng, "2-inter", 8, 0.3, 1.0)
    for e in (lazy, eager):
        descs, idx, n = pack_margin_batches([(pl[e], t, g, a, 1.0, 1.0)])
        e.margin_fwd_bwd(descs, idx, n)
        e.materialize()
        e.sync()                                                   # what Engine.reserve / state_dict() do
        assert float(e.grads.abs().max()) > 0.0
        e.adam_step(pl[e].touched)
    assert torch.equal(lazy.params, eager.params)
    assert torch.equal(lazy.exp_avg, eager.exp_avg) and torch.equal(lazy.exp_avg_sq, eager.exp_avg_sq)
    assert float(lazy.grads.abs().max()) == 0.0
    lazy.close()
    eager.close()


def test_lazy_step_after_the_staged_feed_was_overwritten():
    """Lazy Adam with HOST index feeds: margin call, then two host-fed forwards (which cycle through both staging
    buffers and overwrite the margin call's feed), then the optimiser step — the step must not walk the overwritten
    feed.  Compared with the eager engine bit for bit, and a following step still works."""
    import torch
    from gpu_utils import TOY_FORMULAS, TOY_KINDS, TOY_SIZES, engine_from_params, plan_for, random_params
    from graphqembed_amd.tensorize import pack_forward_batches, pack_margin_batches
    from test_gpu_parity import _disjoint_batch
    rng = np.random.RandomState(8)
    d = 32
    params = random_params(rng, d, "transe", "min-simple", TOY_SIZES, TOY_KINDS)
    lazy = engine_from_params(params, d, "transe", "min-simple", lazy_adam=True)
    eager = engine_from_params(params, d, "transe", "min-simple")
    pl = {e: plan_for(e, "2-inter", TOY_FORMULAS["2-inter"]) for e in (lazy, eager)}
    pf = {e: plan_for(e, "2-chain", TOY_FORMULAS["2-chain"]) for e in (lazy, eager)}
    for step in range(6):
        t, g, a = _disjoint_batch(rng, "2-inter", 8, 0.0, 1.0)
        ft = [_disjoint_batch(rng, "2-chain", 8, 0.0, 1.0) for _ in range(2)]
        scores = []
        for e in (lazy, eager):
            descs, idx, n = pack_margin_batches([(pl[e], t, g, a, 1.0, 1.0)])
            e.margin_fwd_bwd(descs, idx, n)                        # numpy idx: host feed through the staging ring
            for (tt, _, aa) in ft:
                descs, idx, n = pack_forward_batches([(pf[e], tt, aa)])
                scores.append(e.forward(descs, idx, n).clone())
            e.adam_step(pl[e].touched)
        assert torch.equal(scores[0], scores[2]) and torch.equal(scores[1], scores[3])
    assert torch.equal(lazy.params, eager.params)
    assert torch.equal(lazy.exp_avg_sq, eager.exp_avg_sq)
    lazy.close()
    eager.close()


def test_load_state_dict_settles_deferred_steps_first():
    """A lazy-Adam model whose rows owe steps is given a new state_dict: the loaded values must be what the next
    forward reads (the deferred steps belong to the old values and are settled before the copy)."""
    import torch
    from test_gpu_api import build_world, rebuild_queries
    from graphqembed_amd.model import FusedAdam
    model, _ = build_world("bilinear-diag", "min", 32, "train_bilinear-diag_min_d32.npz")
    ref, _ = build_world("bilinear-diag", "min", 32, "train_bilinear-diag_min_d32.npz")
    model.engine._check(model.engine.lib.gqe_set_lazy_adam(model.engine.ctx, 1))
    model.engine.lazy_adam = True
    train, _ = rebuild_queries()
    opt = FusedAdam(model, lr=0.01)
    f2 = next(iter(train["2-inter"]))
    f1 = next(iter(train["1-chain"]))
    for it in range(3):
        qs = train["2-inter"][f2][:4]
        t, a = model._rows(f2, qs, [q.target_node for q in qs])
        neg = model.enc.rows([q.neg_samples[0] for q in qs], f2.target_mode)
        model.margin_step([(f2, t, neg, a, 1.0, 1.0)])
        opt.step()
    sd = {k: v.clone() for k, v in ref.state_dict().items()}
    model.load_state_dict(sd)
    qs = train["1-chain"][f1][:16]
    nodes = [q.target_node for q in qs]
    assert torch.equal(model.forward(f1, qs, nodes), ref.forward(f1, qs, nodes))
    model.sync()
    for k, v in model.state_dict().items():
        assert torch.equal(v, sd[k]), k


def test_allreduce_grads_entry_point_over_rccl():
    """gqe_allreduce_grads (the dense exchange the north star names, as one library call): lists folded into the dense
    arena + ncclAllReduce over an RCCL communicator.  One GPU here, so the communicator has one rank (the sum is the
    identity) — what is checked is the entry point itself: librccl bound at run time, the call enqueued on the caller's
    stream, the gradient afterwards equal to the materialised gradient, and the optimiser step still consuming it."""
    import torch
    from gpu_utils import TOY_FORMULAS, TOY_KINDS, TOY_SIZES, engine_from_params, plan_for, random_params, read_arena, toy_batch
    from graphqembed_amd import parallel
    from graphqembed_amd.tensorize import pack_margin_batches
    rng = np.random.RandomState(6)
    d = 64
    params = random_params(rng, d, "bilinear-diag", "min", TOY_SIZES, TOY_KINDS)
    a, b = engine_from_params(params, d, "bilinear-diag", "min"), engine_from_params(params, d, "bilinear-diag", "min")
    comm = parallel.RcclComm(0, 1)
    t, g, anc = toy_batch(rng, "3-inter", 100)
    for eng in (a, b):
        plan = plan_for(eng, "3-inter", TOY_FORMULAS["3-inter"])
        descs, idx, n = pack_margin_batches([(plan, t, g, anc, 1.0, 1.0)])
        eng.margin_fwd_bwd(descs, idx, n)
    a.allreduce_grads(comm.handle)
    ga, gb = read_arena(a, a.grads), read_arena(b, b.grads)
    for k in ga:
        scale = max(1e-12, float(np.abs(gb[k]).max()))
        np.testing.assert_allclose(ga[k], gb[k], rtol=0, atol=2e-5 * scale, err_msg=k)      # (atomics order differs between engines)
    a.adam_step(plan.touched)
    a.materialize()
    assert float(a.grads.abs().max()) == 0.0
    comm.close()
    a.close()
    b.close()


def test_rank_candidates_against_scipy_including_nan():
    """gqe_rank_candidates == scipy.stats.percentileofscore(kind 'rank') list by list — ties, single-candidate lists, and NaN:
    a NaN target score or a NaN among the candidates gives nan (scipy's nan_policy 'propagate'), not a percentile of 0 that
    would silently lower the mean (utils.py:26-33, 91)."""
    import torch
    from scipy import stats
    from gpu_utils import TOY_KINDS, TOY_SIZES, engine_from_params, random_params
    from graphqembed_amd.utils import _percentile_of_score
    rng = np.random.RandomState(5)
    eng = engine_from_params(random_params(rng, 16, "bilinear-diag", "min", TOY_SIZES, TOY_KINDS), 16, "bilinear-diag", "min")
    lists = [np.round(rng.randn(rng.randint(2, 200)), 1).astype(np.float32) for _ in range(40)]     # rounded: ties
    lists[3][0] = np.nan                   # the target's score
    lists[7][5 % len(lists[7])] = np.nan   # a candidate's score
    lists[9] = lists[9][:2]
    ptr = np.concatenate([[0], np.cumsum([len(l) for l in lists])]).astype(np.int32)
    got = eng.rank_candidates(torch.from_numpy(np.concatenate(lists)).cuda(), ptr).cpu().numpy()
    for i, l in enumerate(lists):
        want = stats.percentileofscore(l[1:], l[0])
        assert (np.isnan(want) and np.isnan(got[i])) or abs(got[i] - want) < 1e-9, (i, got[i], want)
        mine = _percentile_of_score(l[1:], l[0])
        assert (np.isnan(want) and np.isnan(mine)) or abs(mine - want) < 1e-9, (i, mine, want)
    assert np.isnan(got[3]) and np.isnan(got[7]) and not np.isnan(got[9])
    eng.close()


@pytest.mark.parametrize("env", [{"GQE_DEBUG_GEMM_KMUL": "8"}, {"GQE_DEBUG_GEMM_KMUL": "1"}, {"GQE_DEBUG_FW8_MIN_TILES": "0"},
                                 {"GQE_DEBUG_FW8_MIN_TILES": "1000000"}])
def test_tuning_switches_do_not_change_results(env):
    """The launch-shape heuristics (chunks per pair-GEMM unit, 8- vs 16-wave tiles at d = 128) pick between kernels that must
    agree: the d = 128 many-tile parity cases run again in a process of their own with the heuristic forced either way
    (the switches are read once per process)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider", os.path.join(root, "tests", "test_gpu_parity.py"),
                        "-k", "eight_wave and 128 and bilinear-diag"], cwd=root, env=dict(os.environ, **env), stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, universal_newlines=True, timeout=900)
    assert p.returncode == 0 and " passed" in p.stdout, p.stdout[-2000:]


@pytest.mark.parametrize("dec,inter,d", [("bilinear-diag", "min", 128), ("transe", "mean-simple", 48)])
def test_hot_rows_state_machine(dec, inter, d):
    """Hot rows (include/gqe.h, gqe_hot_rows): on a toy world with 50-90 rows per mode and hundreds of queries per batch, EVERY
    row's gradient list is long, so the first optimiser pass promotes them and every later call goes through the dense
    accumulators instead of the lists — plain rows and, for the bag mode, word rows.  Every consumer of the lists has to consume
    the accumulators too: materialize (gradients vs the fp64 oracle), zero_grads (nothing left), SGD and Adam steps (against a
    hot-rows-free engine built from the same parameters: GQE_HOT has no per-engine switch, so the reference run is the
    ordered-sums engine, whose kernels are compiled without the accumulator path), lazy Adam, and gqe_set_ordered_sums after
    promotion (slots cleared: lists again)."""
    import torch
    from gpu_utils import (TOY_FORMULAS, TOY_KINDS, TOY_SIZES, engine_from_params, plan_for, random_params, read_arena, toy_batch)
    from graphqembed_amd.tensorize import pack_margin_batches
    from test_gpu_parity import assert_grads_close
    rng = np.random.RandomState(11 + d)
    params = random_params(rng, d, dec, inter, TOY_SIZES, TOY_KINDS, bag_modes=("b",))
    B = 300

    def batches(eng, seed):
        r = np.random.RandomState(seed)
        items = []
        for j, qtype in enumerate(("1-chain", "2-inter", "3-inter_chain", "3-chain")):
            t, g, a = toy_batch(r, qtype, B - 7 * j, hub=(j == 1))
            items.append((plan_for(eng, qtype, TOY_FORMULAS[qtype]), t, g, a, [1.0, 0.5, 0.25, 0.1][j], 1.0))
        return items

    def oracle_grads(p, items_spec):
        grads = O.zero_grads_like(p)
        for (qtype, t, g, a, w) in items_spec:
            O.margin_fwd_bwd(p, O.make_plan(qtype, TOY_FORMULAS[qtype]), dec, inter, t, g, a, weight=w, grads=grads)
        grads.pop(O.BAGS_KEY, None)
        return grads

    hot = engine_from_params(params, d, dec, inter, max_queries=4 * B, max_batches=4)
    ref = engine_from_params(params, d, dec, inter, max_queries=4 * B, max_batches=4, ordered_sums=True)   # never promotes
    lazy = engine_from_params(params, d, dec, inter, max_queries=4 * B, max_batches=4, lazy_adam=True)
    keys = list(hot.layout.entries)
    assert hot.hot_rows() == 0
    for step in range(4):
        for eng in (hot, ref, lazy):
            descs, idx, n = pack_margin_batches(batches(eng, 100 + step))
            eng.margin_fwd_bwd(descs, idx, n)
            eng.adam_step(keys)
        if step == 0:
            assert hot.hot_rows() >= 20 and ref.hot_rows() == 0, (hot.hot_rows(), ref.hot_rows())
    torch.cuda.synchronize()
    scale = float(ref.params.abs().max())
    # accumulators sum in arrival order: agreement to rounding (amplified by Adam's g / (|g| + eps) on noise-sized gradients)
    assert float((hot.params - ref.params).abs().max()) <= 2e-3 * scale + 0.021, float((hot.params - ref.params).abs().max())
    assert float((hot.params - ref.params).abs().mean()) <= 2e-4, float((hot.params - ref.params).abs().mean())
    assert float((lazy.params - ref.params).abs().mean()) <= 2e-4 and lazy.hot_rows() >= 20
    # materialize folds accumulators and lists alike: gradients against the fp64 oracle on the parameters of the moment
    cur = read_arena(hot, hot.params)
    if O.BAGS_KEY in params:
        cur[O.BAGS_KEY] = params[O.BAGS_KEY]
    items = batches(hot, 777)
    spec = [(qt, it[1], it[2], it[3], it[4]) for qt, it in zip(("1-chain", "2-inter", "3-inter_chain", "3-chain"), items)]
    descs, idx, n = pack_margin_batches(items)
    hot.margin_fwd_bwd(descs, idx, n)
    assert_grads_close(read_arena(hot, hot.grads), oracle_grads(cur, spec), "hot rows, materialised")
    hot.zero_grads(keys)
    # zero_grads drops what sits in the accumulators
    hot.margin_fwd_bwd(descs, idx, n)
    hot.zero_grads(keys)
    hot.materialize()
    assert float(hot.grads.abs().max()) == 0.0
    # an SGD step fed by the accumulators
    before = read_arena(hot, hot.params)
    want = oracle_grads(cur, spec)
    hot.margin_fwd_bwd(descs, idx, n)
    hot.sgd_step(keys, lr=0.5)
    after = read_arena(hot, hot.params)
    for k in keys:
        np.testing.assert_allclose(after[k], before[k] - 0.5 * want[k], rtol=2e-3, atol=2e-5 * max(1.0, float(np.abs(want[k]).max())), err_msg=k)
    # ordered sums after promotion: the slots are cleared, gradients come from the lists again
    hot._check(hot.lib.gqe_set_ordered_sums(hot.ctx, 1))
    cur = read_arena(hot, hot.params)
    if O.BAGS_KEY in params:
        cur[O.BAGS_KEY] = params[O.BAGS_KEY]
    hot.margin_fwd_bwd(descs, idx, n)
    assert_grads_close(read_arena(hot, hot.grads), oracle_grads(cur, spec), "ordered sums after promotion")
    for eng in (hot, ref, lazy):
        eng.close()


@pytest.mark.parametrize("env", [{"GQE_HOT_FEW_LEN": "0", "GQE_HOT_SUB": "0"}, {"GQE_HOT_FEW_LEN": "0"}, {"GQE_HOT_SUB": "0"}])
def test_hot_row_switches_do_not_change_results(env):
    """The hot rows' forms have to agree: all 32 accumulators for every promoted row (GQE_HOT_FEW_LEN=0: rows promoted on short lists
    otherwise keep to 8), word rows without sub-lists (GQE_HOT_SUB=0: one atomic row per word and bag, as before round 6), both.  The
    state-machine test and the sub-list test run again in a process of their own under each setting (the switches are read once per
    process; the sub-list test's own child process inherits them — without sub-lists it is skipped)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    k = "hot_rows_state_machine" + ("" if env.get("GQE_HOT_SUB") == "0" else " or sub_lists_and_their_overflow")
    p = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider", os.path.join(root, "tests", "test_gpu_limits.py"), "-k", k],
                       cwd=root, env=dict(os.environ, **env), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, timeout=900)
    assert p.returncode == 0 and " passed" in p.stdout, p.stdout[-2000:]


def test_hot_word_sub_lists_and_their_overflow_chains():
    """Hot WORD rows (include/gqe.h, gqe_hot_sub_lists): a promoted row of a bag table gets sub-lists sized from the list length that
    promoted it — arrays of 128 entries behind a counter, an overflow chain of link nodes behind the array — and a gather launch
    behind the fused kernel sums them into the row's accumulators.  A child process promotes EVERY word row of the toy world on a
    short list (GQE_HOT_MIN_LEN=2, batches of 40 queries: one sub-list each) and then sends hub batches of 300 queries through
    them: the busiest word gets more than 128 entries in a step, i.e. the array AND the chain.  Gradients against the fp64 oracle
    (materialize folds the accumulators), zero_grads leaves nothing behind, and Adam steps agree with the ordered-sums engine
    (which never promotes)."""
    import os
    import subprocess
    import sys
    code = r'''
import sys, numpy as np, torch
sys.path.insert(0, %r); sys.path.insert(0, %r)
from oracle import netquery_numpy as O
from gpu_utils import (TOY_FORMULAS, TOY_KINDS, TOY_SIZES, engine_from_params, plan_for, random_params, read_arena, toy_batch)
from graphqembed_amd.tensorize import pack_margin_batches
from test_gpu_parity import assert_grads_close
d, dec, inter = 128, "bilinear-diag", "min"
rng = np.random.RandomState(3)
params = random_params(rng, d, dec, inter, TOY_SIZES, TOY_KINDS, bag_modes=("b",))
ptr, ids = params[O.BAGS_KEY]["b"]
QT = ("1-chain", "2-inter", "3-inter_chain", "3-chain")
def batches(eng, seed, B, hub):
    r = np.random.RandomState(seed)
    out = []
    for j, qtype in enumerate(QT):
        t, g, a = toy_batch(r, qtype, B - j, hub=hub)
        out.append((plan_for(eng, qtype, TOY_FORMULAS[qtype]), t, g, a, [1.0, 0.5, 0.25, 0.1][j], 1.0))
    return out
def sent_to_words(items):     # bag entries a step sends to every word row (an upper bound: inactive hinges send nothing)
    sent = np.zeros(params[O.table_key("b")].shape[0], dtype=np.int64)
    for (pl, t, g, a, w, m), qtype in zip(items, QT):
        op = O.make_plan(qtype, TOY_FORMULAS[qtype])
        rows = [t, g] if op["target_mode"] == "b" else []
        rows += [a[i] for i, am in enumerate(op["anchor_modes"]) if am == "b"]
        for r in rows:
            for b in np.asarray(r).ravel():
                np.add.at(sent, ids[ptr[b]:ptr[b + 1]], 1)
    return sent
hot = engine_from_params(params, d, dec, inter, max_queries=4 * 300, max_batches=4)
ref = engine_from_params(params, d, dec, inter, max_queries=4 * 300, max_batches=4, ordered_sums=True)
keys = list(hot.layout.entries)
items777 = batches(hot, 777, 300, hub=True)
sent = sent_to_words(items777)
busiest = int(np.argmax(sent))
assert sent[busiest] > 128 + 64, sent[busiest]                # more than the one array of its sub-list holds
for step, B in enumerate((40, 300, 300, 300)):
    for eng in (hot, ref):
        items = batches(eng, 50 + step, B, hub=step > 0)
        descs, idx, n = pack_margin_batches(items)
        eng.margin_fwd_bwd(descs, idx, n)
        eng.adam_step(keys)
    if step == 0:
        sent0 = sent_to_words(items)
        # the word that will be the busiest below is promoted here, on a list of <= 32 entries like every other row: one sub-list
        assert sent0[busiest] >= 8 and sent0.max() <= 32, (sent0[busiest], sent0.max())
        heads, on = hot.hot_sub_lists()
        rows0 = hot.hot_rows()
        assert rows0 >= int((sent0 >= 8).sum()) and heads == rows0 and on, (rows0, heads, on)
torch.cuda.synchronize()
assert float((hot.params - ref.params).abs().mean()) <= 2e-4, float((hot.params - ref.params).abs().mean())
assert float((hot.params - ref.params).abs().max()) <= 2e-3 * float(ref.params.abs().max()) + 0.021
cur = read_arena(hot, hot.params)
cur[O.BAGS_KEY] = params[O.BAGS_KEY]
items = items777
grads = O.zero_grads_like(cur)
for (pl, t, g, a, w, m), qtype in zip(items, QT):
    O.margin_fwd_bwd(cur, O.make_plan(qtype, TOY_FORMULAS[qtype]), dec, inter, t, g, a, weight=w, grads=grads)
grads.pop(O.BAGS_KEY, None)
descs, idx, n = pack_margin_batches(items)
hot.margin_fwd_bwd(descs, idx, n)
assert_grads_close(read_arena(hot, hot.grads), grads, "hot word rows: arrays + overflow chains")
hot.zero_grads(keys)
hot.margin_fwd_bwd(descs, idx, n)
hot.zero_grads(keys)
hot.materialize()
assert float(hot.grads.abs().max()) == 0.0
print("sub-lists ok", rows0, hot.hot_rows(), hot.hot_sub_lists(), int(sent0[busiest]), int(sent0.max()), int(sent[busiest]))
''' % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GQE_HOT_MIN_LEN="2")
    env.pop("GQE_HOT_SUB", None)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "sub-lists ok" in out.stdout, out.stdout[-1500:] + out.stderr[-3000:]


@pytest.mark.parametrize("dec,inter,d", [("bilinear", "min", 64), ("bilinear-diag", "mean", 256), ("transe", "min", 80)])
def test_operand_copies_follow_every_parameter_write(dec, inter, d):
    """The fused kernels read the d x d matrices from operand-ordered copies in the workspace (include/gqe.h,
    gqe_params_changed).  Whoever writes the parameters, the copies have to follow: the library's Adam and SGD passes (copies
    rewritten in the pass), a caller's own write announced by gqe_params_changed (rebuilt by a launch in front of the next call),
    a re-bound workspace.  After each of them the scores of an intersection batch and a chain batch have to equal those of a
    FRESH engine created from the current parameter values."""
    import torch
    from gpu_utils import (TOY_FORMULAS, TOY_KINDS, TOY_SIZES, engine_from_params, plan_for, random_params, read_arena, toy_batch)
    from graphqembed_amd.tensorize import pack_forward_batches, pack_margin_batches
    rng = np.random.RandomState(5 + d)
    params = random_params(rng, d, dec, inter, TOY_SIZES, TOY_KINDS)
    eng = engine_from_params(params, d, dec, inter)
    r = np.random.RandomState(1)
    spec = [(q,) + toy_batch(r, q, 40) for q in ("3-inter_chain", "3-chain", "2-inter", "3-chain_inter")]

    def scores(e):
        packed = [(plan_for(e, q, TOY_FORMULAS[q]), t, a) for (q, t, g, a) in spec]
        descs, idx, n = pack_forward_batches(packed)
        return e.forward(descs, idx, n).cpu().numpy()

    def check(what):
        fresh = engine_from_params(read_arena(eng, eng._params), d, dec, inter)
        want = scores(fresh)
        fresh.close()
        got = scores(eng)
        assert np.isfinite(want).all()
        np.testing.assert_array_equal(got, want, err_msg=what)

    check("after creation")
    keys = None
    for step, opt in enumerate(("adam", "adam", "sgd", "adam")):
        items = [(plan_for(eng, q, TOY_FORMULAS[q]), t, g, a, 1.0, 1.0) for (q, t, g, a) in spec]
        descs, idx, n_scores = pack_margin_batches(items)
        eng.margin_fwd_bwd(descs, idx, n_scores=n_scores)
        keys = sorted(set().union(*[it[0].touched for it in items]))
        (eng.adam_step if opt == "adam" else eng.sgd_step)(keys)
        check("after %s step %d" % (opt, step))
    # a write behind the library's back: every matrix of the layout scaled in place through a tensor the test kept
    flat = eng._params
    mats = [k for k, (off, shape) in eng.layout.entries.items() if len(shape) == 2 and not k.startswith("enc.")]
    assert mats
    for k in mats:
        eng.layout.view(flat, k).mul_(0.5).add_(0.01)
    eng.params_changed()
    check("after gqe_params_changed")
    eng.reserve(4 * eng.max_queries, eng.max_batches)          # a larger workspace: the copies live in it
    check("after the workspace was re-bound")
    eng.close()


@pytest.mark.parametrize("dec,inter,d", [("bilinear-diag", "min", 128), ("bilinear", "mean", 64), ("transe", "min", 192)])
def test_deferred_pair_gemm_is_the_same_training_run(dec, inter, d):
    """gqe_set_deferred_gemm (include/gqe.h): the matrix-gradient units and the losses' finalize block of a margin call run in
    front of the next Adam pass's chunks, the matrices are stepped by a second launch.  Same batches through an engine with and
    one without the switch: losses (read behind the step), every parameter after every step, and the Adam moments agree to
    float-atomic reordering; and everything that needs the gradients BEFORE an Adam step — materialize, a second margin call,
    an SGD step, zero_grads, a forward call — finds them complete (the deferred launch is flushed on its own)."""
    import torch
    from gpu_utils import (TOY_FORMULAS, TOY_KINDS, TOY_SIZES, engine_from_params, plan_for, random_params, read_arena, toy_batch)
    from graphqembed_amd.tensorize import pack_forward_batches, pack_margin_batches
    rng = np.random.RandomState(17 + d)
    params = random_params(rng, d, dec, inter, TOY_SIZES, TOY_KINDS)
    plain = engine_from_params(params, d, dec, inter)
    lazyg = engine_from_params(params, d, dec, inter)
    lazyg.set_deferred_gemm(True)
    r = np.random.RandomState(2)
    qtypes = ("3-inter_chain", "2-chain", "2-inter", "3-chain_inter", "3-inter")

    def margin(e, spec):
        items = [(plan_for(e, q, TOY_FORMULAS[q]), t, g, a, 1.0 / (1 + j), 1.0) for j, (q, t, g, a) in enumerate(spec)]
        descs, idx, n_scores = pack_margin_batches(items)
        losses, _, _ = e.margin_fwd_bwd(descs, idx, n_scores=n_scores)
        return losses, sorted(set().union(*[it[0].touched for it in items]))

    def close(a, b, what, rtol=1e-4, atol=1e-5):
        # (two runs of ONE engine differ as much: the order of a row's list and of the float atomics is not fixed, and Adam
        # turns a last-bit difference of a small gradient into a visible one of the step)
        # ... and an element whose gradient is of the order of Adam's eps moves by a visibly different amount: 1 in 2000 may
        for k in a:
            bad = np.abs(b[k] - a[k]) > atol + rtol * np.abs(a[k])
            assert bad.mean() <= 5e-4, "%s %s: %d of %d elements differ, worst %.3g" % (what, k, bad.sum(), bad.size, np.abs(b[k] - a[k]).max())

    for step in range(4):
        spec = [(q,) + toy_batch(r, q, 150 + 16 * j) for j, q in enumerate(qtypes)]
        l0, keys = margin(plain, spec)
        l1, _ = margin(lazyg, spec)
        plain.adam_step(keys)
        lazyg.adam_step(keys)
        np.testing.assert_allclose(l1.cpu().numpy(), l0.cpu().numpy(), rtol=1e-6, err_msg="losses behind step %d" % step)
        close(read_arena(plain, plain._params), read_arena(lazyg, lazyg._params), "params after step %d" % step)
        close(read_arena(plain, plain._exp_avg_sq), read_arena(lazyg, lazyg._exp_avg_sq), "second moments after step %d" % step, rtol=1e-3, atol=1e-9)
    # consumers other than an Adam step
    spec = [(q,) + toy_batch(r, q, 96) for q in qtypes[:3]]
    for e in (plain, lazyg):
        margin(e, spec)
        e.materialize()
    close(read_arena(plain, plain.grads), read_arena(lazyg, lazyg.grads), "materialized gradient", rtol=1e-4, atol=2e-6)   # (a row's list is summed in whatever order it was linked)
    for e in (plain, lazyg):
        e.zero_grads(list(e.layout.entries))
    assert float(lazyg.grads.abs().max()) == 0.0
    for e in (plain, lazyg):
        _, keys = margin(e, spec)
        packed = [(plan_for(e, q, TOY_FORMULAS[q]), t, a) for (q, t, g, a) in spec]
        descs, idx, n = pack_forward_batches(packed)
        e.forward(descs, idx, n)                            # overwrites nothing the pending units need: they were launched first
        e.sgd_step(keys)
    close(read_arena(plain, plain._params), read_arena(lazyg, lazyg._params), "params after margin + forward + sgd")
    plain.close()
    lazyg.close()


def test_check_tiles_switch_catches_an_unannounced_parameter_write():
    """GQE_CHECK_TILES=1 (debug switch, read once per process): a forward call after a parameter write that was NOT announced with
    gqe_params_changed fails with an error that says so, instead of scoring with the old matrices; the announced write passes."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
from gpu_utils import TOY_FORMULAS, TOY_KINDS, TOY_SIZES, engine_from_params, plan_for, random_params, toy_batch
from graphqembed_amd.engine import GqeError
from graphqembed_amd.tensorize import pack_forward_batches
rng = np.random.RandomState(0)
d = 64
eng = engine_from_params(random_params(rng, d, "bilinear", "min", TOY_SIZES, TOY_KINDS), d, "bilinear", "min")
t, g, a = toy_batch(rng, "2-inter", 32)
def fwd():
    descs, idx, n = pack_forward_batches([(plan_for(eng, "2-inter", TOY_FORMULAS["2-inter"]), t, a)])
    return eng.forward(descs, idx, n)
fwd()
flat = eng._params                       # (not through the Engine.params property, which announces)
mats = [k for k, (off, shape) in eng.layout.entries.items() if len(shape) == 2 and not k.startswith("enc.")]
for k in mats:                           # (only the matrices a registered formula names are watched: touch them all)
    eng.layout.view(flat, k).add_(1.0)
try:
    fwd()
    print("NOT CAUGHT")
except GqeError as e:
    print("CAUGHT" if "gqe_params_changed" in str(e) else "OTHER %%s" %% e)
eng.params_changed()
fwd()
print("ANNOUNCED OK")
""" % (root, os.path.join(root, "tests"))
    p = subprocess.run([sys.executable, "-c", code], cwd=root, env=dict(os.environ, GQE_CHECK_TILES="1"), stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, universal_newlines=True, timeout=600)
    assert "CAUGHT" in p.stdout and "NOT CAUGHT" not in p.stdout and "ANNOUNCED OK" in p.stdout, p.stdout[-2000:]


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_deferred_pair_gemm_under_random_call_sequences(seed):
    """The same random sequence of calls — margin, Adam step, SGD step, materialize, zero_grads, forward, two margins in a row,
    a lazy-Adam stretch — through an engine with gqe_set_deferred_gemm and one without: whenever the sequence looks at the
    state (parameters after a step, the dense gradient after materialize, scores), the two agree.  What rides and what is
    flushed is the library's business; the caller only ever reads what the contract defines."""
    import torch
    from gpu_utils import (TOY_FORMULAS, TOY_KINDS, TOY_SIZES, engine_from_params, plan_for, random_params, read_arena, toy_batch)
    from graphqembed_amd.tensorize import pack_forward_batches, pack_margin_batches
    d, dec, inter = 64, ("bilinear", "bilinear-diag", "transe")[seed % 3], ("min", "mean", "min")[seed % 3]
    rng = np.random.RandomState(100 + seed)
    params = random_params(rng, d, dec, inter, TOY_SIZES, TOY_KINDS)
    engines = [engine_from_params(params, d, dec, inter), engine_from_params(params, d, dec, inter)]
    engines[1].set_deferred_gemm(True)
    qtypes = ("3-inter_chain", "2-chain", "2-inter", "3-chain_inter", "3-inter", "1-chain")
    pending = [set(), set()]          # tensors with an un-stepped gradient

    def same(a, b, what, rtol=1e-4, atol=1e-5):
        for k in a:
            bad = np.abs(b[k] - a[k]) > atol + rtol * np.abs(a[k])
            assert bad.mean() <= 1e-3, "%s %s: %d of %d differ, worst %.3g" % (what, k, bad.sum(), bad.size, np.abs(b[k] - a[k]).max())

    r = np.random.RandomState(7 + seed)
    rides_seen = 0
    for op_i in range(40):
        op = r.choice(["margin", "margin", "adam", "adam", "sgd", "materialize", "zero", "forward"])
        spec = [(q,) + toy_batch(r, q, 48 + 16 * j) for j, q in enumerate(r.choice(qtypes, size=3, replace=False))]
        outs = []
        for ei, e in enumerate(engines):
            if op == "margin":
                items = [(plan_for(e, q, TOY_FORMULAS[q]), t, g, a, 1.0, 1.0) for (q, t, g, a) in spec]
                descs, idx, n_scores = pack_margin_batches(items)
                e.margin_fwd_bwd(descs, idx, n_scores=n_scores)
                pending[ei] |= set().union(*[it[0].touched for it in items])
            elif op in ("adam", "sgd") and pending[ei]:
                keys = sorted(pending[ei])
                (e.adam_step if op == "adam" else e.sgd_step)(keys)
                pending[ei] = set()
                outs.append(read_arena(e, e._params))
            elif op == "materialize":
                e.materialize()
                outs.append(read_arena(e, e.grads))
            elif op == "zero" and pending[ei]:
                e.zero_grads(sorted(pending[ei]))
                pending[ei] = set()
                outs.append(read_arena(e, e.grads))
            elif op == "forward":
                packed = [(plan_for(e, q, TOY_FORMULAS[q]), t, a) for (q, t, g, a) in spec]
                descs, idx, n = pack_forward_batches(packed)
                outs.append({"scores": e.forward(descs, idx, n).cpu().numpy()})
        if len(outs) == 2:
            same(outs[0], outs[1], "op %d (%s)" % (op_i, op), atol=1e-5 if op != "forward" else 2e-5)
    rides_seen = engines[1].gemm_rides()
    assert engines[0].gemm_rides() == 0
    if dec != "transe" or inter in ("min", "mean"):
        assert rides_seen > 0, "the sequence never let a pair GEMM ride"
    for e in engines:
        e.close()


def test_merged_dense_segment_over_matrices_keeps_copies_and_rides_consistent():
    """The C ABI accepts ANY dense segment: one flat segment over all relation / Pre / Post parameters moves the matrices
    without being their own d x d universe entries.  (1) the operand-ordered copies must follow (GQE_CHECK_TILES=1: every
    forward / backward call verifies them): the pass marks them dirty and the next launch rebuilds them; (2) with
    gqe_set_deferred_gemm such a pass must NOT carry the riding GEMM units (its chunks would read and zero a matrix gradient the
    units are still adding to): the deferred launch is flushed first; (3) gqe_train_step takes the two-call sequence.  Results ==
    an engine stepped tensor by tensor."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import sys, ctypes as C, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import torch
from gpu_utils import TOY_FORMULAS, TOY_KINDS, TOY_SIZES, engine_from_params, plan_for, random_params, read_arena, toy_batch
from graphqembed_amd.engine import gqe_segment
from graphqembed_amd.tensorize import pack_margin_batches
rng = np.random.RandomState(0)
d, dec, inter = 64, "bilinear", "min"
params = random_params(rng, d, dec, inter, TOY_SIZES, TOY_KINDS)
ref, eng = engine_from_params(params, d, dec, inter), engine_from_params(params, d, dec, inter)
eng.set_deferred_gemm(True)
lay = eng.layout
tables = [k for k in lay.entries if k.startswith("enc.")]
dense = [k for k in lay.entries if not k.startswith("enc.")]
lo = min(lay.offset(k) for k in dense)
hi = max(lay.offset(k) + lay.numel(k) for k in dense)

def merged_segments(step):
    arr = (gqe_segment * (len(tables) + 1))()
    for i, k in enumerate(tables):
        arr[i].offset, arr[i].numel, arr[i].step = lay.offset(k), lay.numel(k), step
    arr[len(tables)].offset, arr[len(tables)].numel, arr[len(tables)].step = lo, hi - lo, step
    return arr

for step in range(1, 4):
    items = [(q,) + toy_batch(rng, q, 96) for q in ("3-inter", "2-inter", "2-chain", "3-inter_chain", "1-chain", "3-chain_inter")]
    keys = list(lay.entries)                      # both engines step EVERY tensor (a flat segment cannot leave one out): same step counts
    for e in (ref, eng):
        packed = [(plan_for(e, q, TOY_FORMULAS[q]), t, g, a, 1.0, 1.0) for (q, t, g, a) in items]
        descs, idx, n = pack_margin_batches(packed)
        if e is ref:
            e.margin_fwd_bwd(descs, idx, n)
            e.adam_step(keys)
        elif step < 3:
            e.margin_fwd_bwd(descs, idx, n)       # the pair GEMM is deferred ...
            segs = merged_segments(step)          # ... and the pass that follows covers the matrices inside ONE flat segment
            e._check(e.lib.gqe_adam_step(e.ctx, segs, len(segs), 0.01, 0.9, 0.999, 1e-8, e._stream()))
        else:                                     # the same through gqe_train_step
            arr = e.make_batches(descs)
            keep, ptr, n_idx, on_dev = e._idx_arg(idx)
            losses = torch.empty(len(descs) + 1, device=e.device)
            segs = merged_segments(step)
            e._check(e.lib.gqe_train_step(e.ctx, arr, len(descs), ptr, n_idx, on_dev, segs, len(segs), 0.01, 0.9, 0.999, 1e-8, losses.data_ptr(), e._stream()))
    a, b = read_arena(ref, ref.params), read_arena(eng, eng.params)
    for k in a:
        diff = np.abs(a[k].astype(np.float64) - b[k])
        assert (diff > 1e-4 * max(float(np.abs(a[k]).max()), 1e-30) + 1e-5).mean() <= 2e-3, (step, k, float(diff.max()))
assert eng.gemm_rides() == 0 and eng.split_steps() == 0, (eng.gemm_rides(), eng.split_steps())
print("MERGED OK")
""" % (root, os.path.join(root, "tests"))
    p = subprocess.run([sys.executable, "-c", code], cwd=root, env=dict(os.environ, GQE_CHECK_TILES="1"), stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, universal_newlines=True, timeout=600)
    assert "MERGED OK" in p.stdout, p.stdout[-3000:]
