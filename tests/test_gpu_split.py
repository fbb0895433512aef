"""-m gpu: gqe_train_step (include/gqe.h) — the whole iteration as one library call, run as a "split step" where that applies
(Adam over the rows the batches do not name rides in the fused launch: csrc/gqe_split.h).  Against the two-call step
(gqe_margin_fwd_bwd + gqe_adam_step) and against the fp64 oracle's gradient + Adam step."""
import numpy as np
import pytest

from oracle import netquery_numpy as O

pytestmark = pytest.mark.gpu

SPLIT_DIMS = [64, 128, 256]          # the straight-line kernels carry riders; every other dim runs the two-call sequence


def _world(rng, d, dec, inter, sizes=None):
    from gpu_utils import TOY_KINDS, TOY_SIZES, engine_from_params, random_params
    sizes = sizes or TOY_SIZES
    params = random_params(rng, d, dec, inter, sizes, TOY_KINDS)
    return params, (lambda **kw: engine_from_params(params, d, dec, inter, **kw))


def _batches(eng, rng, types, B, hub=False):
    from gpu_utils import TOY_FORMULAS, plan_for, toy_batch
    items = []
    for k, qt in enumerate(types):
        t, ng, a = toy_batch(rng, qt, B + 7 * k, hub=hub and k == 0)
        items.append((plan_for(eng, qt, TOY_FORMULAS[qt]), t, ng, a, 1.0 if qt == "1-chain" else 0.01, 1.0))
    return items


def _named_rows(items, layout):
    """{table key: set of rows the step's feed names}"""
    from graphqembed_amd.tensorize import table_key
    named = {}
    for (plan, t, ng, a, w, m) in items:
        f = plan.formula
        named.setdefault(table_key(f.target_mode), set()).update(int(x) for x in np.concatenate([t, ng]))
        for i, am in enumerate(f.anchor_modes):
            named.setdefault(table_key(am), set()).update(int(x) for x in a[i])
    return named


@pytest.mark.parametrize("d", SPLIT_DIMS + [48])
@pytest.mark.parametrize("dec,inter", [("bilinear-diag", "min"), ("bilinear", "mean"), ("transe", "min-simple"), ("bilinear-diag", "mean-simple")])
def test_train_step_is_the_two_call_step(dec, inter, d):
    """Six iterations with changing formulas: one engine steps with margin_fwd_bwd + adam_step, the other with train_step.
    After EVERY iteration: the same losses; the rows the iteration's feed does not name are BIT-equal in p, m and v (the riders
    execute the eager pass's arithmetic on a zero gradient); named rows, vectors and matrices agree to what float atomics in
    another order allow.  d = 48 (a guarded kernel: no riders) must take the two-call sequence inside the library."""
    import torch
    from gpu_utils import read_arena
    from graphqembed_amd.tensorize import pack_margin_batches
    rng = np.random.RandomState(5)
    params, make = _world(rng, d, dec, inter)
    a, b = make(), make()
    mixes = [["1-chain"], ["1-chain", "2-chain", "2-inter"], ["3-inter", "3-chain", "3-inter_chain", "2-inter"], ["2-chain"],
             ["1-chain", "3-chain_inter", "3-inter"], ["2-inter", "3-inter_chain"]]
    for it, types in enumerate(mixes):
        items = _batches(a, rng, types, 33 + 5 * it, hub=it == 2)
        descs, idx, n = pack_margin_batches(items)
        keys = set().union(*[p[0].touched for p in items])
        # every iteration starts from the SAME state (list sums in arrival order differ in the last bit between two runs:
        # a row named in an earlier iteration would carry that bit into the bit-equality check of this one)
        b.params.copy_(a.params); b.exp_avg.copy_(a.exp_avg); b.exp_avg_sq.copy_(a.exp_avg_sq)
        la, _, _ = a.margin_fwd_bwd(descs, idx, n)
        a.adam_step(keys)
        lb = b.train_step(descs, idx, keys)
        np.testing.assert_allclose(lb.cpu().numpy(), la.cpu().numpy(), rtol=2e-5, atol=1e-7, err_msg="iteration %d" % it)
        named = _named_rows(items, a.layout)
        for name, fa, fb in (("p", a.params, b.params), ("m", a.exp_avg, b.exp_avg), ("v", a.exp_avg_sq, b.exp_avg_sq)):
            xa, xb = read_arena(a, fa), read_arena(b, fb)
            for k in xa:
                if k.startswith("enc."):
                    rows = np.ones(xa[k].shape[0], dtype=bool)
                    rows[sorted(named.get(k, ()))] = False
                    assert np.array_equal(xa[k][rows], xb[k][rows]), (it, name, k, "rows the feed does not name must be bit-equal")
                    rows = ~rows
                else:
                    rows = slice(None)
                # (Adam turns rounding noise on an exactly-cancelling gradient into lr-sized moves: compare where the step is defined)
                diff = np.abs(xa[k][rows].astype(np.float64) - xb[k][rows])
                scale = max(float(np.abs(xa[k][rows]).max()), 1e-30) if np.size(xa[k][rows]) else 1.0
                bad = diff > 1e-4 * scale + (1e-5 if name == "p" else 1e-9)
                assert bad.mean() <= 2e-3 if bad.size else True, (it, name, k, int(bad.sum()), bad.size, float(diff.max()))
        assert float(b.grads.abs().max()) == 0.0
    assert b.split_steps() == (len(mixes) if d in SPLIT_DIMS else 0), b.split_steps()
    a.close()
    b.close()


@pytest.mark.parametrize("mode", ["two-call", "deferred", "train-step"])
@pytest.mark.parametrize("dec,inter,d", [("bilinear-diag", "min", 128), ("bilinear", "mean", 64), ("transe", "mean", 128)])
def test_step_modes_vs_oracle_adam(dec, inter, d, mode):
    """Three iterations, every way of stepping the library has (gqe_margin_fwd_bwd + gqe_adam_step; the same with
    gqe_set_deferred_gemm; gqe_train_step), against the fp64 oracle: loss, then the PARAMETERS AFTER THE STEP against
    O.adam_step on the oracle's gradient — elements whose gradient is signal (|g| > 1e-4 max|g| or exactly 0) within 2e-3 of the
    lr-sized move; at most 0.2 % of them may sit on the other side of a relu / arg-min decision."""
    import torch
    from gpu_utils import TOY_FORMULAS, read_arena
    from graphqembed_amd.tensorize import pack_margin_batches
    rng = np.random.RandomState(11)
    params, make = _world(rng, d, dec, inter)
    eng = make()
    if mode == "deferred":
        eng.set_deferred_gemm(True)
    oparams = {k: v.astype(np.float64) for k, v in params.items()}
    ostate = {}
    for it, types in enumerate([["1-chain", "2-inter", "3-inter"], ["2-chain", "3-inter_chain", "3-chain_inter", "3-chain"], ["1-chain", "2-inter"]]):
        items = _batches(eng, rng, types, 300 if it == 0 else 64)     # (300 queries: more than one 128-query chunk per pair-GEMM unit)
        descs, idx, n = pack_margin_batches(items)
        keys = set().union(*[p[0].touched for p in items])
        before = read_arena(eng, eng.params)
        ograds = O.zero_grads_like(oparams)
        want_l = []
        for (plan, t, ng, a, w, m) in items:
            f = plan.formula
            l, _, _, _ = O.margin_fwd_bwd(oparams, O.make_plan(f.query_type, f.rels), dec, inter, t, ng, a, margin=m, weight=w, grads=ograds)
            want_l.append(l)
        if mode == "train-step":
            losses = eng.train_step(descs, idx, keys)
        else:
            losses, _, _ = eng.margin_fwd_bwd(descs, idx, n)
            eng.adam_step(keys)                       # (deferred: the losses are defined behind the step)
        np.testing.assert_allclose(losses.cpu().numpy()[:-1], want_l, rtol=1e-4, err_msg="%s iteration %d" % (mode, it))
        O.adam_step(oparams, ograds, ostate, keys)
        after = read_arena(eng, eng.params)
        for k in sorted(keys):
            g = np.abs(ograds[k])
            signal = (g == 0) | (g > 1e-4 * g.max())
            move = np.abs(oparams[k] - before[k])
            diff = np.abs(after[k].astype(np.float64) - oparams[k])
            bad = signal & (diff > 2e-3 * move + 2e-7)
            assert bad.sum() <= max(2, 2e-3 * signal.sum()), (mode, it, k, int(bad.sum()), int(signal.sum()), float(diff[signal].max()))
        for k in set(after) - keys:
            assert np.array_equal(after[k], before[k]), (mode, it, k)
        # the oracle continues from the DEVICE's parameters: trajectories are compared step by step, not compounded
        oparams = {k: v.astype(np.float64) for k, v in after.items()}
        host_m, host_v = read_arena(eng, eng.exp_avg), read_arena(eng, eng.exp_avg_sq)
        for k in keys:
            ostate[k]["m"], ostate[k]["v"] = host_m[k].astype(np.float64), host_v[k].astype(np.float64)
    if mode == "deferred":
        assert eng.gemm_rides() > 0
    if mode == "train-step":
        assert eng.split_steps() == 3
    eng.close()


def test_train_step_state_machine():
    """Whatever follows a split step finds the matrices stepped: forward, another kind of step, zero_grads, a workspace that grows,
    parameters read through the engine; gradients pending from an earlier margin call send the step down the two-call path."""
    import torch
    from gpu_utils import TOY_FORMULAS, plan_for, read_arena
    from graphqembed_amd.tensorize import pack_forward_batches, pack_margin_batches
    rng = np.random.RandomState(3)
    d, dec, inter = 128, "bilinear-diag", "min"
    params, make = _world(rng, d, dec, inter)
    a, b = make(max_queries=512), make(max_queries=512)

    def both(fn):
        return fn(a, False), fn(b, True)

    def step(eng, split, types, B):
        items = _batches(eng, np.random.RandomState(B), types, B)
        descs, idx, n = pack_margin_batches(items)
        keys = set().union(*[p[0].touched for p in items])
        if split:
            return eng.train_step(descs, idx, keys)
        l, _, _ = eng.margin_fwd_bwd(descs, idx, n)
        eng.adam_step(keys)
        return l

    def scores(eng, split):
        rs = np.random.RandomState(77)
        from gpu_utils import toy_batch
        t, ng, anch = toy_batch(rs, "3-inter", 40)
        descs, idx, n = pack_forward_batches([(plan_for(eng, "3-inter", TOY_FORMULAS["3-inter"]), t, anch)])
        return eng.forward(descs, idx, n).cpu().numpy()

    def close(x, y, what):
        np.testing.assert_allclose(x, y, rtol=1e-3, atol=2e-5, err_msg=what)

    la, lb = both(lambda e, s: step(e, s, ["2-inter", "3-inter", "1-chain"], 60))
    close(la.cpu().numpy(), lb.cpu().numpy(), "losses")
    # a forward right behind a split step: contracts with the matrices that step moved (its matrix step was still pending)
    close(*both(scores), "forward behind a split step")
    both(lambda e, s: step(e, s, ["3-inter_chain", "2-chain"], 90))
    # the engine's accessors settle the pending matrix step
    pa, pb = read_arena(a, a.params), read_arena(b, b.params)
    for k in pa:
        if not k.startswith("enc."):
            close(pa[k], pb[k], k)
    # a batch larger than the bound workspace: the engine re-binds it between two split steps
    both(lambda e, s: step(e, s, ["3-inter", "2-inter", "3-chain_inter"], 400))
    both(lambda e, s: step(e, s, ["1-chain"], 20))            # a step without matrix jobs
    close(*both(scores), "forward after growth")
    # gradients pending from a plain margin call: the next train_step steps them too, through the two-call path
    n_split = b.split_steps()
    for eng in (a, b):
        items = _batches(eng, np.random.RandomState(1), ["2-inter"], 50)
        descs, idx, n = pack_margin_batches(items)
        eng.margin_fwd_bwd(descs, idx, n)
    both(lambda e, s: step(e, s, ["2-inter", "2-chain"], 70))
    assert b.split_steps() == n_split            # (not split: lists were pending)
    both(lambda e, s: step(e, s, ["2-inter", "2-chain"], 70))
    assert b.split_steps() == n_split + 1
    # zero_grads / sgd / materialize behind a split step
    for eng in (a, b):
        eng.zero_grads(list(eng.layout.entries))
        eng.materialize()
        assert float(eng.grads.abs().max()) == 0.0
    close(*both(scores), "forward at the end")
    pa, pb = read_arena(a, a.params), read_arena(b, b.params)
    for k in pa:
        diff = np.abs(pa[k].astype(np.float64) - pb[k])
        assert np.median(diff) < 1e-6 and (diff > 1e-3).mean() < 2e-3, (k, float(diff.max()))
    a.close()
    b.close()


def test_train_step_through_the_reference_api():
    """train_helpers.run_train with FusedAdam goes through model.train_step: the run of tests/test_gpu_api.py's d = 128 fixture,
    batch by batch, with the one-call step — same losses as the reference recorded, same parameters after five iterations."""
    import json
    import torch
    from golden_utils import to_rels
    from test_gpu_api import build_world
    from graphqembed_amd.graph import Formula
    from graphqembed_amd.model import FusedAdam
    model, z = build_world("bilinear-diag", "min", 128, "train_bilinear-diag_min_d128.npz")
    p0 = {k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()}
    opt = FusedAdam(model, lr=0.01)
    i = 0
    while "it%d/n" % i in z.files:
        items = []
        for j in range(int(z["it%d/n" % i])):
            meta = json.loads(str(z["it%d/b%d/meta" % (i, j)]))
            w = 1.0 if meta["type"] == "1-chain" else (0.005 if "inter" in meta["type"] else 0.01)
            items.append((Formula(meta["type"], to_rels(meta["rels"])), z["it%d/b%d/target" % (i, j)], z["it%d/b%d/neg" % (i, j)],
                          z["it%d/b%d/anchors" % (i, j)], w, float(meta["margin"])))
        opt.zero_grad()
        l = model.train_step(items, opt).cpu().numpy()
        for j in range(len(items)):
            np.testing.assert_allclose(l[j], float(z["it%d/b%d/loss" % (i, j)]), rtol=1e-4 if i == 0 else 5e-2, atol=1e-5, err_msg="it %d batch %d" % (i, j))
        np.testing.assert_allclose(l[-1], float(z["it%d/loss" % i]), rtol=1e-4 if i == 0 else 3e-2)
        opt.step()                         # (nothing left to do: the step was part of train_step)
        i += 1
    assert i == 5 and model.engine.split_steps() == 5
    got = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    for k in got:
        diff = np.abs(got[k].astype(np.float64) - p0[k] - z["delta/" + k])
        assert diff.max() < 6e-2 and np.median(diff) < 1e-3, (k, diff.max(), np.median(diff))


@pytest.mark.parametrize("dec,inter,d", [("bilinear-diag", "min", 128), ("bilinear", "mean", 64)])
def test_lazy_adam_row_launch_carries_the_deferred_pair_gemm(dec, inter, d):
    """Lazy (deferred, bit-exact) Adam with gqe_set_deferred_gemm: the matrix-gradient units and the loss finalize ride in the step's
    ROW launch (gqe_rows_ride_kernel), the d x d matrices are stepped by gqe_matstep_kernel behind it.  Against an eager engine
    without the switch on the same batches — with and without the next feed declared (gqe_lazy_prefetch: one launch covers
    rows(t) and rows(t+1)) — losses, parameters and moments agree to float-atomic reordering once the lazy engine is synchronised."""
    import torch
    from gpu_utils import read_arena
    from graphqembed_amd.tensorize import pack_margin_batches
    rng = np.random.RandomState(8)
    params, make = _world(rng, d, dec, inter)
    eager, lazy = make(), make(lazy_adam=True)
    lazy.set_deferred_gemm(True)
    mixes = [["2-inter", "1-chain"], ["3-inter", "2-chain"], ["3-inter_chain", "2-inter", "3-chain"], ["2-inter"], ["3-chain_inter", "1-chain"], ["3-inter", "2-inter"]]
    batches = [_batches(eager, rng, types, 200 + 16 * it) for it, types in enumerate(mixes)]
    prepared = []
    for items in batches:
        descs, idx, n = pack_margin_batches(items)
        prepared.append((descs, idx, n, set().union(*[p[0].touched for p in items]),
                         lazy.prepare_margin(descs, torch.from_numpy(idx).to(lazy.device))))
    for it, (descs, idx, n, keys, ps) in enumerate(prepared):
        le, _, _ = eager.margin_fwd_bwd(descs, idx, n)
        eager.adam_step(keys)
        lazy.run_margin(ps)
        if it % 2 == 0 and it + 1 < len(prepared):
            lazy.lazy_prefetch(prepared[it + 1][4])                     # rows(t) and rows(t+1) in one launch
        lazy.adam_step(keys)
        torch.cuda.synchronize()
        np.testing.assert_allclose(ps["losses"].cpu().numpy(), le.cpu().numpy(), rtol=2e-5, atol=1e-7, err_msg="iteration %d" % it)
    assert lazy.gemm_rides() == len(prepared), lazy.gemm_rides()
    for name, fa, fb in (("p", eager.params, lazy.params), ("m", eager.exp_avg, lazy.exp_avg), ("v", eager.exp_avg_sq, lazy.exp_avg_sq)):
        xa, xb = read_arena(eager, fa), read_arena(lazy, fb)
        for k in xa:
            diff = np.abs(xa[k].astype(np.float64) - xb[k])
            scale = max(float(np.abs(xa[k]).max()), 1e-30)
            bad = diff > 1e-4 * scale + (1e-5 if name == "p" else 1e-9)
            assert bad.mean() <= 2e-3, (name, k, int(bad.sum()), bad.size, float(diff.max()))
    assert float(lazy.grads.abs().max()) == 0.0
    eager.close()
    lazy.close()


def test_train_step_host_feeds_without_synchronisation():
    """Host index feeds are staged through two device buffers by the library's upload stream.  A split step reads its feed THREE
    times (stamps, tiles, named rows): the staging buffer may only be reused once the step's second launch has read it — a loop that
    runs many iterations ahead of the device, each with a different feed, must give what the same loop gives with device-resident
    feeds."""
    import torch
    from gpu_utils import read_arena
    from graphqembed_amd.tensorize import pack_margin_batches
    rng = np.random.RandomState(12)
    d, dec, inter = 128, "bilinear-diag", "min"
    params, make = _world(rng, d, dec, inter)
    host, dev = make(max_queries=2048), make(max_queries=2048)
    steps = []
    for it in range(24):
        items = _batches(host, rng, ["1-chain", "2-inter", "3-inter", "2-chain"], 300 + (it % 5) * 16)
        descs, idx, n = pack_margin_batches(items)
        steps.append((descs, idx, set().union(*[p[0].touched for p in items])))
    staged = [torch.from_numpy(idx).to(dev.device) for (_, idx, _) in steps]
    torch.cuda.synchronize()
    for (descs, idx, keys) in steps:                     # no synchronisation: the host runs ahead of the device
        host.train_step(descs, idx, keys)
    for (descs, idx, keys), didx in zip(steps, staged):
        dev.train_step(descs, didx, keys)
    torch.cuda.synchronize()
    assert host.split_steps() == len(steps) == dev.split_steps()
    # a second launch that read ANOTHER step's feed would have left the lists of its own named rows linked: nothing may be pending
    host.materialize()
    assert float(host.grads.abs().max()) == 0.0
    a, b = read_arena(dev, dev.params), read_arena(host, host.params)
    # (Two runs of 24 steps on this tiny, fully coupled world are not expected to agree closely: ONE run against itself lands on a
    # few discrete trajectories — median differences of 1e-8, 2e-5 or 1e-4 — whichever way it is stepped or fed, two calls
    # included (tools/probes/race_probe.py): a hinge / arg-min decision that float-atomic order flips early on.  A feed race would
    # step rows with another iteration's stamps: lr-sized errors everywhere.)
    for k in a:
        diff = np.abs(a[k].astype(np.float64) - b[k])
        assert np.isfinite(b[k]).all() and float(diff.max()) < 0.1, (k, float(diff.max()))
        if k.startswith("enc."):      # (a relation vector whose gradient is rounding noise moves by +- lr per step either way)
            assert np.median(diff) < 1e-3 and (diff > 2e-2).mean() <= 1e-3, (k, float(diff.max()), float(np.median(diff)), float((diff > 2e-2).mean()))
    host.close()
    dev.close()


@pytest.mark.parametrize("d", [64, 128])
def test_train_step_with_a_bag_table(d):
    """A world with an nn.EmbeddingBag mode (Reddit posts: mode b = bags over a word table): the word table's gradient lists hang
    on word rows no feed names, so a split step leaves that table to its second launch's ordinary chunk loop (lists, link nodes,
    hot accumulators) while the riders cover the plain tables.  Against the two-call step, iteration by iteration from the same
    state: losses, plain tables' unnamed rows bit-equal, everything else to float-atomic reordering."""
    import torch
    from gpu_utils import TOY_KINDS, TOY_SIZES, engine_from_params, random_params, read_arena
    from graphqembed_amd.tensorize import pack_margin_batches
    rng = np.random.RandomState(31)
    dec, inter = "bilinear-diag", "min"
    params = random_params(rng, d, dec, inter, TOY_SIZES, TOY_KINDS, bag_modes=("b",))
    a, b = engine_from_params(params, d, dec, inter), engine_from_params(params, d, dec, inter)
    bag_key = O.table_key("b")
    for it, types in enumerate([["1-chain", "2-inter"], ["3-inter", "2-chain", "3-inter_chain"], ["2-inter", "3-chain_inter"], ["1-chain"]]):
        items = _batches(a, rng, types, 60 + 11 * it)
        descs, idx, n = pack_margin_batches(items)
        keys = set().union(*[p[0].touched for p in items])
        b.params.copy_(a.params); b.exp_avg.copy_(a.exp_avg); b.exp_avg_sq.copy_(a.exp_avg_sq)
        la, _, _ = a.margin_fwd_bwd(descs, idx, n)
        a.adam_step(keys)
        lb = b.train_step(descs, idx, keys)
        np.testing.assert_allclose(lb.cpu().numpy(), la.cpu().numpy(), rtol=2e-5, atol=1e-7, err_msg="iteration %d" % it)
        named = _named_rows(items, a.layout)
        xa, xb = read_arena(a, a.params), read_arena(b, b.params)
        for k in xa:
            if k.startswith("enc.") and k != bag_key:
                rows = np.ones(xa[k].shape[0], dtype=bool)
                rows[sorted(named.get(k, ()))] = False
                assert np.array_equal(xa[k][rows], xb[k][rows]), (it, k)
            diff = np.abs(xa[k].astype(np.float64) - xb[k])
            scale = max(float(np.abs(xa[k]).max()), 1e-30)
            assert (diff > 1e-4 * scale + 1e-5).mean() <= 2e-3, (it, k, float(diff.max()))
    assert b.split_steps() == 4
    b.materialize()                                   # nothing left on a list, a link node or in an accumulator
    assert float(b.grads.abs().max()) == 0.0
    a.close()
    b.close()


def test_split_step_epoch_wraps(tmp_path):
    """The stamps carry a 15-bit epoch (gqe_split.h): every 32 767 split steps the array is cleared and the epoch starts over.  A
    child process starts its epochs at 32 760 (GQE_SPLIT_DEBUG_EPOCH0) and runs 20 iterations across the wrap: every iteration
    against the two-call step on the same state — same losses, unnamed rows bit-equal in p, m, v, named rows stepped (a stamp
    misread on either side of the wrap would leave a named row to the riders, or an unnamed one to nobody)."""
    import os
    import subprocess
    import sys
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import test_gpu_split as T
from gpu_utils import read_arena
from graphqembed_amd.tensorize import pack_margin_batches
rng = np.random.RandomState(9)
params, make = T._world(rng, 64, "bilinear-diag", "min")
a, b = make(), make()
for it in range(20):
    items = T._batches(a, rng, ["1-chain", "2-inter", "3-chain"], 40 + it)
    descs, idx, n = pack_margin_batches(items)
    keys = set().union(*[p[0].touched for p in items])
    b.params.copy_(a.params); b.exp_avg.copy_(a.exp_avg); b.exp_avg_sq.copy_(a.exp_avg_sq)
    before = read_arena(b, b.params)
    la, _, _ = a.margin_fwd_bwd(descs, idx, n)
    a.adam_step(keys)
    lb = b.train_step(descs, idx, keys)
    np.testing.assert_allclose(lb.cpu().numpy(), la.cpu().numpy(), rtol=2e-5, atol=1e-7)
    named = T._named_rows(items, a.layout)
    for fa, fb in ((a.params, b.params), (a.exp_avg, b.exp_avg), (a.exp_avg_sq, b.exp_avg_sq)):
        xa, xb = read_arena(a, fa), read_arena(b, fb)
        for k in xa:
            if k.startswith("enc."):
                rows = np.ones(xa[k].shape[0], dtype=bool)
                rows[sorted(named.get(k, ()))] = False
                assert np.array_equal(xa[k][rows], xb[k][rows]), (it, k)
                assert np.abs(xa[k][~rows].astype(np.float64) - xb[k][~rows]).max() < 1.1e-2, (it, k)
    after = read_arena(b, b.params)
    for k, rows in named.items():
        moved = np.abs(after[k][sorted(rows)] - before[k][sorted(rows)]).max(axis=1)
        assert (moved > 0).mean() > 0.5, (it, k)          # the named rows were stepped by somebody
assert b.split_steps() == 20
print("wrapped ok")
''' % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GQE_SPLIT_DEBUG_EPOCH0="32760")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "wrapped ok" in out.stdout, out.stdout[-1500:] + out.stderr[-3000:]
