"""-m gpu: the HIP path (through libgqe.so's C ABI) against the golden vectors of the
reference and against the fp64 oracle.  Tolerances (fp32 device arithmetic, atomics in
arbitrary order): scores atol 2e-5, loss rtol 1e-4, gradients rtol 2e-3 + atol 1e-6*scale."""
import os
import numpy as np
import pytest

from golden_utils import GOLDEN, adam1_files, case_names, load_case, load_params, load_reddit_params, model_files, reddit_files
from oracle import netquery_numpy as O

pytestmark = pytest.mark.gpu

SCORE_ATOL = 2e-5
LOSS_RTOL = 1e-4


def assert_grads_close(got, want, what, want32=None):
    """``want32``: the same gradients from the oracle run in fp32.  relu / first-arg-min / hinge decisions
    that sit within fp32 rounding of their threshold legitimately go either way in fp32 (device and
    fp32 oracle alike); where the fp32 and fp64 oracles disagree, that disagreement is allowed as slack."""
    for k in want:
        scale = max(float(np.abs(want[k]).max()), 1e-12)
        tol = 2e-3 * np.abs(want[k]) + 2e-6 * scale + 1e-9
        if want32 is not None:
            tol = tol + 2.0 * np.abs(want32[k].astype(np.float64) - want[k])
        diff = np.abs(got[k] - want[k])
        if (diff > tol).any():
            i = np.unravel_index(np.argmax(diff - tol), diff.shape)
            raise AssertionError("%s %s: %d/%d elements off; worst at %s: got %r want %r (tol %.3g, scale %.3g)"
                                 % (what, k, int((diff > tol).sum()), diff.size, i, got[k][i], want[k][i], tol[i], scale))


def _ids(v):
    import os
    return os.path.basename(v) if isinstance(v, str) and v.endswith(".npz") else None


@pytest.mark.parametrize("path,dec,inter,d", model_files(), ids=_ids)
def test_golden_scores_loss_grads(path, dec, inter, d):
    import torch
    from gpu_utils import engine_from_params, plan_for, read_arena
    from graphqembed_amd.tensorize import pack_forward_batches, pack_margin_batches
    z = np.load(path)
    params = load_params(z, d)
    eng = engine_from_params(params, d, dec, inter)
    for case in case_names(z):
        c = load_case(z, case)
        plan = plan_for(eng, c["type"], c["rels"])
        # forward only (gqe_forward): positives and negatives as two target lists
        descs, idx, n = pack_forward_batches([(plan, c["target"], c["anchors"]), (plan, c["neg"], c["anchors"])])
        s = eng.forward(descs, idx, n).cpu().numpy()
        B = len(c["target"])
        np.testing.assert_allclose(s[:B], c["pos"], atol=SCORE_ATOL, rtol=1e-4, err_msg=case)
        np.testing.assert_allclose(s[B:], c["negscore"], atol=SCORE_ATOL, rtol=1e-4, err_msg=case)
        # fused forward + backward (gqe_margin_fwd_bwd)
        eng.grads.zero_()
        descs, idx, n = pack_margin_batches([(plan, c["target"], c["neg"], c["anchors"], 1.0, c["margin"])])
        losses, pos, neg = eng.margin_fwd_bwd(descs, idx, n, want_scores=True)
        np.testing.assert_allclose(pos.cpu().numpy(), c["pos"], atol=SCORE_ATOL, rtol=1e-4, err_msg=case)
        np.testing.assert_allclose(neg.cpu().numpy(), c["negscore"], atol=SCORE_ATOL, rtol=1e-4, err_msg=case)
        l = losses.cpu().numpy()
        np.testing.assert_allclose(l[0], c["loss"], rtol=LOSS_RTOL, err_msg=case)
        np.testing.assert_allclose(l[1], c["loss"], rtol=LOSS_RTOL, err_msg=case)
        got = read_arena(eng, eng.grads)
        assert_grads_close(got, c["grads"], case)
        for k in set(got) - set(c["grads"]):
            assert not got[k].any(), (case, k)
        assert plan.touched == set(c["grads"].keys()), case
    eng.close()


@pytest.mark.parametrize("path,dec,inter,d", model_files(32), ids=_ids)
def test_golden_adam_three_steps(path, dec, inter, d):
    """Three margin + Adam steps against the trajectory the reference recorded (torch.optim.Adam, lr 0.01).
    Step 1 is pinned tightly elsewhere (test_one_adam_step_from_the_golden_gradient).  After it, Adam's sign-like first
    steps (dp = lr g / (|g| + 1e-8)) turn gradients that are rounding noise around an exact 0 into lr-sized moves in the
    reference itself, so the comparison is made where it is well defined: on the SIGNAL elements — those whose gradient
    in every step is either exactly 0 or above 1e-4 of the tensor's largest (classified with the fp64 oracle run on the
    same batches) — at rtol 1e-3 (+ 2e-6) for all but 5 % of them (2 elements of a small tensor; an arg-min that flips for a few
    elements moves those by a fraction of lr without showing in the loss) and 2e-3 absolute for all (the loose bound is 4e-2), whenever the three losses show that no discrete decision (arg-min / relu /
    hinge) flipped on the way (loss of steps 2-3 within 1e-4 of the reference's).  A trajectory that did flip — the fp32
    numpy oracle does so on 2 of the 132 recorded cases — is held to the per-case bound only (3 x the deviation of the fp32 oracle's
    own trajectory on that case and tensor from the recorded one, + 2e-3; losses: 3 x the fp32 oracle's loss deviation), and at most 3
    cases per model may take that route."""
    from gpu_utils import engine_from_params, load_params as put, plan_for, read_arena
    from graphqembed_amd.tensorize import pack_margin_batches
    z = np.load(path)
    p0 = load_params(z, d)
    eng = engine_from_params(p0, d, dec, inter)
    diverged, tight = [], 0
    for case in case_names(z):
        c = load_case(z, case)
        if "adam_neg" not in c:
            continue
        put(eng, p0)
        eng.grads.zero_(); eng.exp_avg.zero_(); eng.exp_avg_sq.zero_()
        eng.steps = {k: 0 for k in eng.steps}
        plan = plan_for(eng, c["type"], c["rels"])
        oplan = O.make_plan(c["type"], c["rels"])
        oparams, ostate = {k: v.astype(np.float64) for k, v in p0.items()}, {}
        # the SAME trajectory through the oracle in fp32: how far fp32 arithmetic alone lands from what the reference recorded
        # on THIS case is the yardstick for the device (x 3), instead of one blanket bound for all cases
        o32, s32, loss32 = {k: v.astype(np.float32) for k, v in p0.items()}, {}, []
        signal = {k: np.ones(p0[k].shape, dtype=bool) for k in plan.touched}
        loss_err = 0.0
        for step in range(3):
            l32, _, _, g32 = O.margin_fwd_bwd(o32, oplan, dec, inter, c["target"], c["adam_neg"][step], c["anchors"], margin=c["margin"])
            O.adam_step(o32, g32, s32, plan.touched)
            loss32.append(abs(float(l32) - float(c["adam_loss"][step])))
            descs, idx, n = pack_margin_batches([(plan, c["target"], c["adam_neg"][step], c["anchors"], 1.0, c["margin"])])
            losses, _, _ = eng.margin_fwd_bwd(descs, idx, n)
            got_loss = float(losses.cpu().numpy()[0])
            np.testing.assert_allclose(got_loss, c["adam_loss"][step], rtol=LOSS_RTOL, atol=3.0 * loss32[-1] + (0.0 if step == 0 else 2e-4 * abs(float(c["adam_loss"][step]))),
                                       err_msg="%s step %d (fp32 oracle off by %.3g)" % (case, step, loss32[-1]))
            loss_err = max(loss_err, abs(got_loss - c["adam_loss"][step]) / max(abs(c["adam_loss"][step]), 1e-12))
            eng.adam_step(plan.touched)
            _, _, _, og = O.margin_fwd_bwd(oparams, oplan, dec, inter, c["target"], c["adam_neg"][step], c["anchors"], margin=c["margin"])
            for k in plan.touched:
                g = np.abs(og[k])
                signal[k] &= (g == 0) | (g > 1e-4 * g.max())
            O.adam_step(oparams, og, ostate, plan.touched)
        got = read_arena(eng, eng.params)
        assert float(eng.grads.abs().max()) == 0.0
        flipped = loss_err > 1e-4
        for k, delta in c["adam_delta"].items():
            diff = np.abs(got[k].astype(np.float64) - p0[k] - delta)
            # the bound every trajectory has to meet: three times what the fp32 oracle itself is off by on this case and tensor
            # (Adam amplifies rounding noise on ~zero gradients: tests/test_oracle_golden.py), never more than the old blanket 4e-2
            dev32 = np.abs(o32[k].astype(np.float64) - p0[k] - delta)
            allow = min(4e-2, 3.0 * float(dev32.max()) + 2e-3)
            assert diff.max() < allow and np.median(diff) < max(5e-5, 3.0 * float(np.median(dev32))), (case, k, diff.max(), allow, np.median(diff), float(np.median(dev32)))
            if not flipped and signal[k].any():
                sg = signal[k]
                bad = int((diff[sg] > 1e-3 * np.abs(delta[sg]) + 2e-6).sum())
                assert bad <= max(2, 0.05 * sg.sum()) and diff[sg].max() < 2e-3, (case, k, bad, int(sg.sum()), float(diff[sg].max()))
        if flipped:
            diverged.append(case)
        else:
            tight += 1
        for k in set(got) - set(c["adam_delta"]):
            assert np.array_equal(got[k], p0[k]), (case, k)
    assert len(diverged) <= 3 and tight >= 6, (diverged, tight)
    eng.close()


@pytest.mark.parametrize("path,dec,inter,d", reddit_files(), ids=_ids)
def test_golden_reddit_embedding_bag(path, dec, inter, d):
    """Bag modes (Reddit posts = nn.EmbeddingBag mean over word rows): scores, loss, gradients of the word
    table and everything else, then the 3-step Adam trajectory, against the reference's own outputs."""
    from gpu_utils import engine_from_params, load_params as put, plan_for, read_arena
    from graphqembed_amd.tensorize import pack_forward_batches, pack_margin_batches
    z = np.load(path)
    params = load_reddit_params(z)
    eng = engine_from_params(params, d, dec, inter)
    for case in case_names(z):
        c = load_case(z, case)
        plan = plan_for(eng, c["type"], c["rels"])
        put(eng, params)
        eng.exp_avg.zero_(); eng.exp_avg_sq.zero_(); eng.zero_grads(list(eng.layout.entries))
        eng.steps = {k: 0 for k in eng.steps}
        descs, idx, n = pack_forward_batches([(plan, c["target"], c["anchors"]), (plan, c["neg"], c["anchors"])])
        s = eng.forward(descs, idx, n).cpu().numpy()
        B = len(c["target"])
        np.testing.assert_allclose(s[:B], c["pos"], atol=SCORE_ATOL, rtol=1e-4, err_msg=case)
        np.testing.assert_allclose(s[B:], c["negscore"], atol=SCORE_ATOL, rtol=1e-4, err_msg=case)
        descs, idx, n = pack_margin_batches([(plan, c["target"], c["neg"], c["anchors"], 1.0, c["margin"])])
        losses, pos, neg = eng.margin_fwd_bwd(descs, idx, n, want_scores=True)
        np.testing.assert_allclose(losses.cpu().numpy()[0], c["loss"], rtol=LOSS_RTOL, err_msg=case)
        got = read_arena(eng, eng.grads)
        assert_grads_close(got, c["grads"], case)
        for k in set(got) - set(c["grads"]):
            assert not got[k].any(), (case, k)
        if "adam_neg" not in c:       # the d=128 fixture (hard negatives for every intersection type) carries no trajectory
            continue
        # Adam: lists (incl. bag link nodes) consumed directly by the optimiser pass
        eng.zero_grads(list(eng.layout.entries))
        for step in range(3):
            descs, idx, n = pack_margin_batches([(plan, c["target"], c["adam_neg"][step], c["anchors"], 1.0, c["margin"])])
            losses, _, _ = eng.margin_fwd_bwd(descs, idx, n)
            np.testing.assert_allclose(losses.cpu().numpy()[0], c["adam_loss"][step], rtol=LOSS_RTOL if step == 0 else 6e-2, err_msg=case)
            eng.adam_step(plan.touched)
        now = read_arena(eng, eng.params)
        for k, delta in c["adam_delta"].items():
            diff = np.abs(now[k].astype(np.float64) - params[k] - delta)
            assert diff.max() < 4e-2 and np.median(diff) < 5e-4, (case, k, diff.max(), np.median(diff))
    eng.close()


@pytest.mark.parametrize("path,dec,inter,d", adam1_files(), ids=_ids)
def test_one_adam_step_from_the_golden_gradient(path, dec, inter, d):
    """The reference's own gradient written into the dense gradient arena, ONE gqe_adam_step, against the reference's
    parameters after one torch.optim.Adam step: atol 1e-6 (gqe_adam1 uses v_sqrt_f32 / v_rcp_f32, 1 ulp, not IEEE —
    csrc/gqe_adam.h; the step is lr-sized, so that is ~1e-9 here), and every other tensor must not move."""
    import os
    import torch
    from gpu_utils import engine_from_params, load_params as put, read_arena
    z = np.load(path)
    p0 = load_params(np.load(os.path.join(GOLDEN, "model_%s_%s_d%d.npz" % (dec, inter, d))), d)
    eng = engine_from_params(p0, d, dec, inter)
    for case in sorted(set(k.split("/")[0] for k in z.files)):
        put(eng, p0)
        eng.exp_avg.zero_(); eng.exp_avg_sq.zero_()
        eng.steps = {k: 0 for k in eng.steps}
        eng.zero_grads(list(eng.layout.entries))
        eng.materialize()                              # the dense gradient is authoritative: it is written by hand
        keys = []
        for k in z.files:
            if k.startswith(case + "/grad/"):
                name = k[len(case) + 6:]
                eng.layout.view(eng.grads, name).copy_(torch.from_numpy(z[k]))
                keys.append(name)
        eng.adam_step(keys)
        got = read_arena(eng, eng.params)
        for name in keys:
            np.testing.assert_allclose(got[name], z[case + "/after/" + name], rtol=0, atol=1e-6, err_msg="%s %s" % (case, name))
        for name in set(got) - set(keys):
            assert np.array_equal(got[name], p0[name]), (case, name)
        assert float(eng.grads.abs().max()) == 0.0
    eng.close()


def test_adam_kernel_matches_formula():
    """gqe_adam_step on random state, 4 steps, two tensors with different step counters and a
    ragged tail, against the fp64 restatement of torch.optim.Adam."""
    import torch
    from gpu_utils import engine_from_params, read_arena
    rng = np.random.RandomState(1)
    params = {"enc.feat-a.weight": rng.randn(37, 16).astype(np.float32),
              "path_dec.a_r_a": rng.randn(16).astype(np.float32),
              "enc.feat-b.weight": rng.randn(1031, 16).astype(np.float32)}
    eng = engine_from_params(params, 16, "bilinear-diag", "min-simple")
    ref = {k: v.astype(np.float64) for k, v in params.items()}
    state = {}
    for step in range(4):
        keys = list(params) if step % 2 == 0 else ["enc.feat-b.weight", "path_dec.a_r_a"]
        grads = {}
        for k in keys:
            g = (rng.randn(*params[k].shape) * 10 ** rng.uniform(-6, 0, size=params[k].shape)).astype(np.float32)
            g[rng.rand(*g.shape) < 0.3] = 0
            grads[k] = g
            eng.materialize()        # declare the dense gradient authoritative before writing it by hand
            eng.layout.view(eng.grads, k).copy_(torch.from_numpy(g))
        eng.adam_step(keys)
        O.adam_step(ref, {k: v.astype(np.float64) for k, v in grads.items()}, state, keys)
        got = read_arena(eng, eng.params)
        for k in params:
            np.testing.assert_allclose(got[k], ref[k], rtol=0, atol=3e-6, err_msg="%s step %d" % (k, step))
        assert float(eng.grads.abs().max()) == 0.0
    # SGD + zero
    k = "enc.feat-b.weight"
    g = rng.randn(*params[k].shape).astype(np.float32)
    eng.materialize()
    eng.layout.view(eng.grads, k).copy_(torch.from_numpy(g))
    before = read_arena(eng, eng.params)[k]
    eng.sgd_step([k], lr=0.05)
    np.testing.assert_allclose(read_arena(eng, eng.params)[k], before - 0.05 * g, atol=1e-6)
    eng.materialize()
    eng.layout.view(eng.grads, k).fill_(3.0)
    eng.zero_grads([k])
    assert float(eng.grads.abs().max()) == 0.0
    eng.close()


CONFIGS = [("bilinear-diag", "min"), ("bilinear-diag", "mean-simple"), ("transe", "mean"),
           ("transe", "min-simple"), ("bilinear", "min"), ("bilinear", "mean-simple"),
           ("bilinear-diag", "mean"), ("transe", "min")]   # the staged-matrix path (d = 64 / 128) with both aggregations per decoder


@pytest.mark.parametrize("d", list(range(16, 257, 16)))   # every dim the library accepts: all guarded variants x decoders
@pytest.mark.parametrize("dec,inter", CONFIGS)
def test_random_schema_vs_oracle(dec, inter, d):
    """Every query type, ragged / tiny / hub-heavy batches, all in ONE grouped launch, against
    the fp64 oracle; then the same batches launched one by one must give the same gradients."""
    from gpu_utils import (TOY_FORMULAS, TOY_KINDS, TOY_SIZES, engine_from_params, plan_for,
                           random_params, read_arena, toy_batch)
    from graphqembed_amd.tensorize import pack_margin_batches
    from graphqembed_amd.engine import DECODERS, INTER_DECODERS, GqeError, load_library
    rng = np.random.RandomState(d * 7 + len(dec) + len(inter))
    params = random_params(rng, d, dec, inter, TOY_SIZES, TOY_KINDS)
    if not load_library().gqe_dim_supported(DECODERS[dec], INTER_DECODERS[inter], d):
        # a configuration whose fused kernel would spill registers: refused, not trusted (include/gqe.h, gqe_dim_supported)
        with pytest.raises(GqeError, match="not supported"):
            engine_from_params(params, d, dec, inter)
        return
    eng = engine_from_params(params, d, dec, inter)
    sizes = {"1-chain": 1, "2-chain": 17, "3-chain": 64, "2-inter": 33, "3-inter": 16, "3-inter_chain": 5,
             "3-chain_inter": 48}
    items, want_l, grads = [], [], O.zero_grads_like(params)
    grads32 = O.zero_grads_like(params, np.float32)
    want_p, want_n = [], []
    for j, (qtype, B) in enumerate(sizes.items()):
        t, g, a = toy_batch(rng, qtype, B, hub=(j % 2 == 0))
        w = [1.0, 0.01, 0.01, 0.005, 0.005, 0.5, 2.0][j]
        m = 1.0 if j != 3 else 0.3
        items.append((plan_for(eng, qtype, TOY_FORMULAS[qtype]), t, g, a, w, m))
        l, sp, sn, _ = O.margin_fwd_bwd(params, O.make_plan(qtype, TOY_FORMULAS[qtype]), dec, inter, t, g, a,
                                        margin=m, weight=w, grads=grads)
        O.margin_fwd_bwd(params, O.make_plan(qtype, TOY_FORMULAS[qtype]), dec, inter, t, g, a,
                         margin=m, weight=w, grads=grads32, dtype=np.float32)
        want_l.append(l); want_p.append(sp); want_n.append(sn)
    descs, idx, n = pack_margin_batches(items)
    losses, pos, neg = eng.margin_fwd_bwd(descs, idx, n, want_scores=True)
    np.testing.assert_allclose(pos.cpu().numpy(), np.concatenate(want_p), atol=SCORE_ATOL, rtol=1e-4)
    np.testing.assert_allclose(neg.cpu().numpy(), np.concatenate(want_n), atol=SCORE_ATOL, rtol=1e-4)
    l = losses.cpu().numpy()
    np.testing.assert_allclose(l[:-1], want_l, rtol=LOSS_RTOL, atol=1e-6)
    np.testing.assert_allclose(l[-1], sum(w * x for (_, _, _, _, w, _), x in zip(items, want_l)), rtol=LOSS_RTOL)
    grouped = read_arena(eng, eng.grads)
    assert_grads_close(grouped, grads, "%s/%s d=%d" % (dec, inter, d), want32=grads32)
    # one launch per batch
    eng.grads.zero_()
    for it in items:
        descs, idx, n = pack_margin_batches([it])
        eng.margin_fwd_bwd(descs, idx, n)
    single = read_arena(eng, eng.grads)
    assert_grads_close(single, grouped, "single vs grouped")
    eng.close()


@pytest.mark.parametrize("dec,inter,d,B", [("bilinear-diag", "min", 128, 1200), ("bilinear", "mean", 128, 1200), ("transe", "min-simple", 128, 1200),
                                           # thousands of pair-GEMM units: a unit walks 2 / 4 chunks of 128 queries (ragged last ones)
                                           ("bilinear-diag", "mean", 128, 4100), ("bilinear", "min", 128, 4100),
                                           ("bilinear-diag", "mean", 144, 40), ("bilinear", "min", 48, 40),
                                           ("bilinear-diag", "min", 176, 40), ("transe", "mean-simple", 240, 40), ("transe", "min", 112, 40)])
def test_eight_wave_workgroups_vs_oracle(dec, inter, d, B):
    _eight_wave_case(dec, inter, d, B)


def _eight_wave_case(dec, inter, d, B, many_tiles=True):
    """The 8-wave shape of the fused kernel (two query rows per wave; csrc/gqe_fused.h): d = 128 launches with more than
    512 tiles (two workgroups per CU; here 7 x 1200 queries = 525 tiles, ragged last tiles), the guarded d in (64, 256)
    variants and the full-Bilinear guarded d < 64 variant — every query type in one grouped launch against the fp64 oracle,
    like the 16-wave shape."""
    from gpu_utils import (TOY_FORMULAS, TOY_KINDS, TOY_SIZES, engine_from_params, plan_for, random_params, read_arena,
                           toy_batch)
    from graphqembed_amd.tensorize import pack_margin_batches
    rng = np.random.RandomState(d + B)
    params = random_params(rng, d, dec, inter, TOY_SIZES, TOY_KINDS)
    eng = engine_from_params(params, d, dec, inter, max_queries=7 * B + 64, max_batches=8)
    items, want_l, want_p, want_n = [], [], [], []
    grads, grads32 = O.zero_grads_like(params), O.zero_grads_like(params, np.float32)
    for j, qtype in enumerate(sorted(TOY_FORMULAS)):
        n = B - 3 * j
        t, g, a = toy_batch(rng, qtype, n, hub=(j == 2))
        w = [1.0, 0.01, 0.5, 0.005, 0.005, 0.3, 2.0][j]
        items.append((plan_for(eng, qtype, TOY_FORMULAS[qtype]), t, g, a, w, 1.0))
        l, sp, sn, _ = O.margin_fwd_bwd(params, O.make_plan(qtype, TOY_FORMULAS[qtype]), dec, inter, t, g, a, weight=w, grads=grads)
        O.margin_fwd_bwd(params, O.make_plan(qtype, TOY_FORMULAS[qtype]), dec, inter, t, g, a, weight=w, grads=grads32, dtype=np.float32)
        want_l.append(l); want_p.append(sp); want_n.append(sn)
    descs, idx, n = pack_margin_batches(items)
    if d == 128 and many_tiles:
        assert sum((len(it[1]) + 15) // 16 for it in items) > 512       # what selects the 8-wave shape at d = 128
    losses, pos, neg = eng.margin_fwd_bwd(descs, idx, n, want_scores=True)
    np.testing.assert_allclose(pos.cpu().numpy(), np.concatenate(want_p), atol=SCORE_ATOL, rtol=1e-4)
    np.testing.assert_allclose(neg.cpu().numpy(), np.concatenate(want_n), atol=SCORE_ATOL, rtol=1e-4)
    np.testing.assert_allclose(losses.cpu().numpy()[:-1], want_l, rtol=LOSS_RTOL, atol=1e-6)
    assert_grads_close(read_arena(eng, eng.grads), grads, "%s/%s d=%d B=%d" % (dec, inter, d, B), want32=grads32)
    # the forward-only kernels of the same shape (their own instantiations, and at d = 128 their own LDS layout)
    from graphqembed_amd.tensorize import pack_forward_batches
    descs, idx, n = pack_forward_batches([(it[0], it[1], it[3]) for it in items])
    np.testing.assert_allclose(eng.forward(descs, idx, n).cpu().numpy(), np.concatenate(want_p), atol=SCORE_ATOL, rtol=1e-4)
    eng.close()


FULL_MIX = [("1-chain", 1.0), ("2-chain", 0.01), ("3-chain", 0.01), ("2-inter", 0.005), ("2-inter", 0.005),
            ("3-inter", 0.005), ("3-inter", 0.005), ("3-inter_chain", 0.005), ("3-inter_chain", 0.005)]
BASELINE_CONFIGS = {
    # BASELINE.json configs at their full batch size (B=512 per batch); the toy schema keeps the oracle fast
    "config1_bio_2chain_2inter_diag_d128": ("bilinear-diag", "min", 128, (), [("1-chain", 1.0), ("2-chain", 0.01), ("2-inter", 0.005), ("2-inter", 0.005)]),
    "config2_bio_full_mix_diag_d128": ("bilinear-diag", "min", 128, (), FULL_MIX),
    "config3_bio_full_mix_bilinear_d128": ("bilinear", "mean", 128, (), FULL_MIX),
    "config4_reddit_full_mix_d256_bags": ("bilinear-diag", "min", 256, ("b",), FULL_MIX + [("3-chain_inter", 0.005)]),
}


@pytest.mark.parametrize("name", sorted(BASELINE_CONFIGS))
def test_baseline_configs_full_batches(name):
    """Every BASELINE.json configuration at B=512 per batch, one grouped launch with the index feed already in
    HBM, against the fp64 oracle (config 4: d=256 with an EmbeddingBag mode, as Reddit posts)."""
    import torch
    from gpu_utils import (TOY_FORMULAS, TOY_KINDS, TOY_SIZES, engine_from_params, plan_for,
                           random_params, read_arena, toy_batch)
    from graphqembed_amd.tensorize import pack_margin_batches
    dec, inter, d, bag_modes, mix = BASELINE_CONFIGS[name]
    rng = np.random.RandomState(len(name))
    params = random_params(rng, d, dec, inter, TOY_SIZES, TOY_KINDS, bag_modes=bag_modes)
    eng = engine_from_params(params, d, dec, inter, max_queries=512 * len(mix), max_batches=len(mix))
    items, grads, want_l = [], O.zero_grads_like(params), []
    grads32 = O.zero_grads_like(params, np.float32)
    for qtype, w in mix:
        t, g, a = toy_batch(rng, qtype, 512)
        items.append((plan_for(eng, qtype, TOY_FORMULAS[qtype]), t, g, a, w, 1.0))
        l, _, _, _ = O.margin_fwd_bwd(params, O.make_plan(qtype, TOY_FORMULAS[qtype]), dec, inter, t, g, a, weight=w, grads=grads)
        O.margin_fwd_bwd(params, O.make_plan(qtype, TOY_FORMULAS[qtype]), dec, inter, t, g, a, weight=w, grads=grads32, dtype=np.float32)
        want_l.append(l)
    descs, idx, n = pack_margin_batches(items)
    didx = torch.from_numpy(idx).cuda()
    losses, _, _ = eng.margin_fwd_bwd(descs, didx, n)
    np.testing.assert_allclose(losses.cpu().numpy()[:-1], want_l, rtol=LOSS_RTOL)
    keys = [k for k in grads if k != O.BAGS_KEY]
    assert_grads_close(read_arena(eng, eng.grads), {k: grads[k] for k in keys}, name, want32={k: grads32[k] for k in keys})
    # size-independent property: a fused Adam step right after must leave no gradient anywhere
    losses, _, _ = eng.margin_fwd_bwd(descs, didx, n)
    eng.adam_step(set().union(*[it[0].touched for it in items]))
    eng.materialize()
    assert float(eng.grads.abs().max()) == 0.0
    eng.close()


@pytest.mark.parametrize("dec,inter,d,bag_modes", [("bilinear-diag", "min", 128, ()), ("transe", "mean", 64, ()),
                                                    ("bilinear", "mean-simple", 32, ()), ("bilinear-diag", "min", 256, ("a", "c"))])
def test_candidate_list_evaluation_matches_expanded_forward(dec, inter, d, bag_modes):
    """gqe_forward with candidate lists (fused evaluation) == scoring every (query, candidate)
    pair as its own forward query, for every query type, ragged lists (0..37 candidates), several batches."""
    from gpu_utils import TOY_FORMULAS, TOY_KINDS, TOY_SIZES, engine_from_params, plan_for, random_params, toy_batch
    from graphqembed_amd.engine import GqeError
    from graphqembed_amd.tensorize import pack_candidate_batches, pack_forward_batches
    rng = np.random.RandomState(d)
    params = random_params(rng, d, dec, inter, TOY_SIZES, TOY_KINDS, bag_modes=bag_modes)
    eng = engine_from_params(params, d, dec, inter)
    cand_items, fwd_items = [], []
    for j, qtype in enumerate(TOY_FORMULAS):
        B = [1, 16, 21, 40, 7, 33, 18][j]
        plan = plan_for(eng, qtype, TOY_FORMULAS[qtype])
        _, _, a = toy_batch(rng, qtype, B)
        nt = TOY_SIZES[O.make_plan(qtype, TOY_FORMULAS[qtype])["target_mode"]]
        lens = rng.randint(0, 38, size=B)
        lens[0] = 37
        ptr = np.zeros(B + 1, dtype=np.int32)
        ptr[1:] = np.cumsum(lens)
        rows = rng.randint(1, nt + 1, size=int(ptr[-1])).astype(np.int32)
        cand_items.append((plan, a, ptr, rows))
        rep = np.repeat(np.arange(B), lens)
        fwd_items.append((plan, rows, a[:, rep]))
    descs, idx, n = pack_forward_batches(fwd_items)
    want = eng.forward(descs, idx, n).cpu().numpy()
    # (full-Bilinear chains project the candidate: their tiles cover candidates, 16 per tile, anchors looked up per candidate)
    descs, idx, n = pack_candidate_batches(cand_items)
    got = eng.forward(descs, idx, n).cpu().numpy()
    np.testing.assert_allclose(got, want, atol=2e-6, rtol=1e-5)
    eng.close()


def test_error_paths():
    from gpu_utils import TOY_FORMULAS, TOY_KINDS, TOY_SIZES, engine_from_params, plan_for, random_params, toy_batch
    from graphqembed_amd.engine import GqeError
    from graphqembed_amd.tensorize import pack_margin_batches
    rng = np.random.RandomState(0)
    params = random_params(rng, 16, "bilinear-diag", "min", TOY_SIZES, TOY_KINDS)
    eng = engine_from_params(params, 16, "bilinear-diag", "min")
    plan = plan_for(eng, "2-inter", TOY_FORMULAS["2-inter"])
    t, g, a = toy_batch(rng, "2-inter", 8)
    descs, idx, n = pack_margin_batches([(plan, t, g, a, 1.0, 1.0)])
    with pytest.raises(GqeError):                 # index buffer too short
        eng.margin_fwd_bwd(descs, idx[:-3], n)
    bad = dict(descs[0]); bad["n_anchors"] = 3
    with pytest.raises(GqeError):
        eng.margin_fwd_bwd([bad], idx, n)
    bad = dict(descs[0]); bad["target_table"] = 10 ** 9
    with pytest.raises(GqeError):
        eng.margin_fwd_bwd([bad], idx, n)
    with pytest.raises(GqeError):
        eng.adam_step([])  if False else eng._check(eng.lib.gqe_adam_step(eng.ctx, None, 0, 0.01, 0.9, 0.999, 1e-8, None))
    eng.close()


@pytest.mark.parametrize("feed", ["zero-copy", "copy"])
@pytest.mark.parametrize("lazy", [False, True])
def test_native_feeder_matches_python_driven_iterations(lazy, feed):
    """(lazy=True: the feeder engine runs in lazy-Adam mode, the Python-driven one eagerly.)  gqe_feeder_run (C++ sampling + packing + launch + step) against the same iterations driven
    from Python: one pool per query type (so the formula draw is forced), 1-chain negatives drawn from a
    single-row list (so the RNG cannot matter), batch size that wraps around the pools."""
    import torch
    from gpu_utils import TOY_FORMULAS, TOY_KINDS, TOY_SIZES, engine_from_params, plan_for, random_params, read_arena, toy_batch

    class Pool(object):
        pass
    rng = np.random.RandomState(11)
    d, dec, inter, B = 64, "bilinear-diag", "min", 48
    params = random_params(rng, d, dec, inter, TOY_SIZES, TOY_KINDS)
    engs = [engine_from_params(params, d, dec, inter, lazy_adam=lazy), engine_from_params(params, d, dec, inter)]
    pools = []
    for qtype in ("1-chain", "2-chain", "2-inter", "3-inter_chain", "3-chain_inter"):
        t, g, a = toy_batch(rng, qtype, 100)
        p = Pool()
        p.target, p.anchors, p.neg, p.hard = t, a, g, (np.roll(g, 1) if "inter" in qtype else None)
        pools.append((qtype, p))
    one_row = {O.table_key("a"): np.array([7], dtype=np.int32)}
    plans0 = [(plan_for(engs[0], qt, TOY_FORMULAS[qt]), p) for qt, p in pools]
    feeder = engs[0].make_feeder(plans0, one_row, batch_size=B, path_weight=0.01, inter_weight=0.005, seed=1, feed=feed)
    losses_f = engs[0].feeder_run(feeder, 0, 3, burn_in=1).cpu().numpy()
    engs[0].feeder_destroy(feeder)
    # the same three iterations from Python on the second engine
    from graphqembed_amd.tensorize import pack_margin_batches
    e1 = engs[1]
    for it in range(3):
        items = []
        for qt, p in pools:
            if it < 1 and qt != "1-chain":
                continue
            plan = plan_for(e1, qt, TOY_FORMULAS[qt])
            n = 100
            s = (it * B) % n
            e = min(((it + 1) * B) % n, n)
            e = n if e <= s else e
            for hard in ((False, True) if "inter" in qt else (False,)):
                neg = np.full(e - s, 7, dtype=np.int32) if qt == "1-chain" else (p.hard if hard else p.neg)[s:e]
                w = 1.0 if qt == "1-chain" else (0.005 if "inter" in qt else 0.01)
                items.append((plan, p.target[s:e], neg, p.anchors[:, s:e], w, 1.0))
        descs, idx, n_s = pack_margin_batches(items)
        losses_p, _, _ = e1.margin_fwd_bwd(descs, idx, n_s)
        e1.adam_step(set().union(*[i[0].touched for i in items]))
    lp = losses_p.cpu().numpy()
    np.testing.assert_allclose(losses_f[:len(lp) - 1], lp[:-1], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(losses_f[len(lp) - 1], lp[-1], rtol=1e-5)
    pf, pp = read_arena(engs[0], engs[0].params), read_arena(e1, e1.params)
    for k in pf:
        np.testing.assert_allclose(pf[k], pp[k], rtol=0, atol=5e-6, err_msg=k)
    for e in engs:
        e.close()


def test_native_feeder_feed_modes_over_many_iterations():
    """60 feeder iterations (the 8 pinned slots are re-used seven times, the guard event fires every 4 iterations; the
    staging ring of the copy mode wraps as often): both feed modes train the same model — same sampled batches (same
    seed), so the loss curves agree up to the atomics noise Adam amplifies — and the loss falls."""
    import torch
    from bench import build_layout, init_params
    from graphqembed_amd import synth
    from graphqembed_amd.data_utils import BIO_TINY_EDGES_PER_KIND, BIO_TINY_SIZES
    from graphqembed_amd.engine import Engine
    from graphqembed_amd.tensorize import FormulaPlan, table_key
    d, dec, inter, B = 32, "bilinear-diag", "min", 64
    g = synth.bio_synth(seed=1, sizes=BIO_TINY_SIZES, edges_per_kind=BIO_TINY_EDGES_PER_KIND)
    layout = build_layout(g, d, dec, inter)
    types = ["1-chain", "2-chain", "3-chain", "2-inter", "3-inter", "3-inter_chain"]
    pools = synth.make_pools(g, types, formulas_per_type=2, pool_size=700, seed=0)
    all_rows = {table_key(m): np.arange(1, g.mode_sizes[m] + 1, dtype=np.int32) for m in g.modes}
    curves = {}
    for feed in ("zero-copy", "copy"):
        eng = Engine(d, dec, inter, layout, max_queries=9 * B, max_batches=9)
        init_params(eng, d, 0)
        plist = [(FormulaPlan(p.formula, layout, inter), p) for t in types for p in pools[t]]
        feeder = eng.make_feeder(plist, all_rows, batch_size=B, seed=5, feed=feed)
        curve = []
        for it in range(0, 60, 3):
            losses = eng.feeder_run(feeder, it, 3)
            curve.append(float(losses[9].item()))
        eng.feeder_destroy(feeder)
        assert bool(torch.isfinite(eng.params).all())
        eng.close()
        curves[feed] = np.asarray(curve)
    a, b = curves["zero-copy"], curves["copy"]
    assert np.isfinite(a).all() and np.isfinite(b).all()
    np.testing.assert_allclose(a[0], b[0], rtol=1e-3)                 # same batches, nearly the same parameters early on
    np.testing.assert_allclose(a, b, rtol=0.15)
    assert a[-3:].mean() < 0.9 * a[:3].mean(), a


def test_gradient_bookkeeping_state_machine():
    """Lists + dense arena bookkeeping: accumulate over calls, materialise in the middle, zero_grads drops pending
    lists, more than 16 batches in one call (several launches), long bags (> 64 words), contribution-buffer
    overflow is an error, not corruption."""
    import torch
    from gpu_utils import TOY_FORMULAS, TOY_KINDS, TOY_SIZES, engine_from_params, plan_for, random_params, read_arena, toy_batch
    from graphqembed_amd.engine import GqeError
    from graphqembed_amd.tensorize import pack_margin_batches
    rng = np.random.RandomState(23)
    d, dec, inter = 32, "bilinear-diag", "mean"
    params = random_params(rng, d, dec, inter, TOY_SIZES, TOY_KINDS, bag_modes=("c",))
    ptr, ids = params[O.BAGS_KEY]["c"]                       # make two bags long: 70 and 150 words
    lens = np.diff(ptr).copy(); lens[3] = 70; lens[5] = 150
    ptr2 = np.zeros_like(ptr); ptr2[1:] = np.cumsum(lens)
    params[O.BAGS_KEY]["c"] = (ptr2, rng.randint(0, 64, size=int(ptr2[-1])).astype(np.int32))
    eng = engine_from_params(params, d, dec, inter, max_queries=64, max_batches=4)
    types = list(TOY_FORMULAS)
    batches = []
    for j in range(20):                                       # 20 batches -> 2 launches inside one call
        qt = types[j % len(types)]
        t, g, a = toy_batch(rng, qt, 3 + j)
        plan_o = O.make_plan(qt, TOY_FORMULAS[qt])
        for arr, mode in [(t, plan_o["target_mode"]), (g, plan_o["target_mode"])] + [(a[i], m) for i, m in enumerate(plan_o["anchor_modes"])]:
            if mode == "c":
                arr[:2] = (3, 5)                              # hit the long bags
        batches.append((qt, t, g, a, 0.1 * (j + 1)))
    want = O.zero_grads_like(params)
    want_l = []
    for qt, t, g, a, w in batches:
        l, _, _, _ = O.margin_fwd_bwd(params, O.make_plan(qt, TOY_FORMULAS[qt]), dec, inter, t, g, a, weight=w, grads=want)
        want_l.append(l)
    keys = [k for k in want if k != O.BAGS_KEY]
    items = [(plan_for(eng, qt, TOY_FORMULAS[qt]), t, g, a, w, 1.0) for qt, t, g, a, w in batches]

    def check(what):
        got = read_arena(eng, eng.grads)
        for k in keys:
            scale = max(np.abs(want[k]).max(), 1e-12)
            np.testing.assert_allclose(got[k], want[k], rtol=2e-3, atol=3e-6 * scale + 1e-9, err_msg="%s %s" % (what, k))
        eng.zero_grads(list(eng.layout.entries))
        assert float(eng.grads.abs().max()) == 0.0

    # (1) all 20 batches in ONE call
    descs, idx, n = pack_margin_batches(items)
    losses, _, _ = eng.margin_fwd_bwd(descs, idx, n)
    l = losses.cpu().numpy()
    np.testing.assert_allclose(l[:20], want_l, rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(l[20], sum(w * x for (_, _, _, _, w), x in zip(batches, want_l)), rtol=1e-4)
    check("one call")
    # (2) one call per batch, a materialise in the middle, the optimiser-free read at the end
    for j, it in enumerate(items):
        descs, idx, n = pack_margin_batches([it])
        eng.margin_fwd_bwd(descs, idx, n)
        if j == 9:
            eng.materialize()
    check("accumulated")
    # (3) zero_grads drops what is pending in the lists
    descs, idx, n = pack_margin_batches(items[:5])
    eng.margin_fwd_bwd(descs, idx, n)
    eng.zero_grads(list(eng.layout.entries))
    got = read_arena(eng, eng.grads)
    assert all(not got[k].any() for k in keys)
    # (4) lists + dense gradient consumed together by one Adam step == everything at once
    p0 = read_arena(eng, eng.params)
    descs, idx, n = pack_margin_batches(items[:10])
    eng.margin_fwd_bwd(descs, idx, n)
    eng.materialize()
    descs, idx, n = pack_margin_batches(items[10:])
    eng.margin_fwd_bwd(descs, idx, n)
    touched = set().union(*[it[0].touched for it in items])
    eng.adam_step(touched)
    ref = {k: v.astype(np.float64) for k, v in params.items() if k != O.BAGS_KEY}
    O.adam_step(ref, {k: want[k] for k in keys}, {}, [k for k in keys if k in touched])
    got = read_arena(eng, eng.params)
    for k in keys:
        diff = np.abs(got[k] - ref[k])
        # first Adam step: |dp| = lr for every non-zero gradient; rounding-noise gradients may differ (see oracle tests)
        assert np.median(diff) < 1e-5 and (diff > 1e-3).mean() < 0.02, (k, np.median(diff), (diff > 1e-3).mean())
    assert float(eng.grads.abs().max()) == 0.0
    # (5) the contribution buffer is bounded: overflow is reported, nothing is corrupted
    small = engine_from_params(params, d, dec, inter, max_queries=32, max_batches=2)
    it5 = (plan_for(small, "3-inter", TOY_FORMULAS["3-inter"]),) + tuple(toy_batch(rng, "3-inter", 30)) + (1.0, 1.0)
    descs, idx, n = pack_margin_batches([it5])
    with pytest.raises(GqeError, match="contribution buffer full"):
        for _ in range(10):
            small.lib.gqe_margin_fwd_bwd  # noqa: B018  (the call below goes through the binding)
            arr = small.make_batches(descs)
            keep, ptr_, n_idx, on_dev = small._idx_arg(idx)
            losses_ = torch.empty(2, device="cuda")
            small._check(small.lib.gqe_margin_fwd_bwd(small.ctx, arr, 1, ptr_, n_idx, on_dev, losses_.data_ptr(), None, None, small._stream()))
    small.materialize()
    small.close()
    eng.close()


def _disjoint_batch(rng, qtype, B, lo_frac, hi_frac):
    from gpu_utils import TOY_FORMULAS, TOY_SIZES
    """A batch in which no table row occurs twice (so gradient lists hold one entry and every sum is
    order-free), drawn from the [lo_frac, hi_frac) part of each table."""
    plan = O.make_plan(qtype, TOY_FORMULAS[qtype])
    pools = {}

    def take(mode, n):
        if mode not in pools:
            size = TOY_SIZES[mode]
            lo, hi = 1 + int(lo_frac * size), 1 + int(hi_frac * size)
            pools[mode] = list(rng.permutation(np.arange(lo, hi)))
        out = [pools[mode].pop() for _ in range(n)]
        return np.asarray(out, dtype=np.int32)
    t = take(plan["target_mode"], B)
    g = take(plan["target_mode"], B)
    a = np.stack([take(m, B) for m in plan["anchor_modes"]])
    return t, g, a


@pytest.mark.parametrize("dec,inter,d", [("bilinear-diag", "min", 32), ("bilinear", "mean", 32), ("transe", "min-simple", 32),
                                         ("bilinear-diag", "min", 64), ("bilinear", "mean", 64), ("transe", "mean", 128),
                                         # d / 4 does not divide 64: a table row's threads must not straddle two waves of the
                                         # optimiser pass (round 4: lazy full passes lost elements at d = 48, 80, 96, ...)
                                         ("bilinear-diag", "min", 48), ("transe", "mean", 80), ("bilinear", "min-simple", 144)])
def test_lazy_adam_is_bit_identical_to_the_eager_schedule(dec, inter, d):
    """gqe_set_lazy_adam: rows without a gradient are not streamed every step; their zero-gradient Adam steps are
    replayed when the row is next read or stepped.  150 iterations on two engines (eager / lazy) with batches built
    so that every reduction is order-free: scores read along the way and the final (p, m, v) arenas must agree BIT
    FOR BIT.  The row ranges move in phases, so rows lag for more steps than the 64-entry coefficient ring holds
    (forcing the interleaved full pass) and come back later."""
    import torch
    from gpu_utils import TOY_FORMULAS, TOY_KINDS, TOY_SIZES, engine_from_params, plan_for, random_params
    from graphqembed_amd.tensorize import pack_forward_batches, pack_margin_batches
    rng = np.random.RandomState(5)
    params = random_params(rng, d, dec, inter, TOY_SIZES, TOY_KINDS)
    eager = engine_from_params(params, d, dec, inter)
    lazy = engine_from_params(params, d, dec, inter, lazy_adam=True)
    assert lazy.lazy_adam and not eager.lazy_adam
    # a second lazy engine that names every next batch before the step (gqe_lazy_prefetch): the step's row launch also
    # catches up the next batch's rows and the next call skips its catch-up launch — same bits
    ahead = engine_from_params(params, d, dec, inter, lazy_adam=True)
    schedule = []
    # no 3-inter: its three branches add into the same Pre gradient with atomics, the only order-dependent sum
    # left at one tile per launch — a run-to-run effect that has nothing to do with the optimiser mode
    types = ["1-chain", "2-inter", "2-chain", "3-inter_chain", "3-chain", "3-chain_inter"]
    B = 6
    n_steps = 150
    for step in range(n_steps):
        qt = types[step % len(types)]
        # phase A: low third of every table; phase B (after 80 steps): upper two thirds; phase C: everything
        lo, hi = (0.0, 0.4) if step < 80 else ((0.35, 1.0) if step < 120 else (0.0, 1.0))
        t, g, a = _disjoint_batch(rng, qt, B, lo, hi)
        schedule.append((qt, t, g, a))
        outs = []
        for eng in (eager, lazy):
            plan = plan_for(eng, qt, TOY_FORMULAS[qt])
            descs, idx, n_scores = pack_margin_batches([(plan, t, g, a, 1.0, 1.0)])
            losses, pos, neg = eng.margin_fwd_bwd(descs, idx, n_scores, want_scores=True)
            eng.adam_step(plan.touched, 0.01)
            outs.append((losses.clone(), pos.clone(), neg.clone()))
        for x, y in zip(*outs):
            assert torch.equal(x, y), "step %d (%s): forward differs between eager and lazy" % (step, qt)
        if step % 37 == 5:      # a forward-only read of rows that may lag (replayed by the catch-up launch)
            tt, _, aa = _disjoint_batch(rng, "2-inter", B, 0.0, 1.0)
            sc = []
            for eng in (eager, lazy):
                descs, idx, n = pack_forward_batches([(plan_for(eng, "2-inter", TOY_FORMULAS["2-inter"]), tt, aa)])
                sc.append(eng.forward(descs, idx, n).clone())
            assert torch.equal(sc[0], sc[1])
    # the same schedule on the prefetching engine, batches device-resident (what gqe_lazy_prefetch names is a device feed)
    prepared = []
    for (qt, t, g, a) in schedule:
        plan = plan_for(ahead, qt, TOY_FORMULAS[qt])
        descs, idx, _ = pack_margin_batches([(plan, t, g, a, 1.0, 1.0)])
        prepared.append((ahead.prepare_margin(descs, torch.from_numpy(idx).to(ahead.device)), ahead.prepare_adam(plan.touched)))
    for step, (ps, pa) in enumerate(prepared):
        ahead.run_margin(ps)
        if step + 1 < len(prepared) and step % 11 != 7:          # (now and then no declaration: the catch-up launch runs)
            ahead.lazy_prefetch(prepared[step + 1][0])
        ahead.run_adam(pa, 0.01)
    torch.cuda.synchronize()
    assert torch.equal(ahead.params, eager.params) and torch.equal(ahead.exp_avg, eager.exp_avg) and torch.equal(ahead.exp_avg_sq, eager.exp_avg_sq)
    ahead.close()
    # lazy really was lazy: before the sync its arena differs from the eager one ... (where the sparse row launch exists: d / 4
    # threads per row must divide a wave; elsewhere every lazy step is a full pass and nothing ever lags)
    if 64 % (d // 4) == 0:
        assert not torch.equal(lazy._params, eager._params)
    # ... and after it (the properties synchronise) everything is bit-identical
    assert torch.equal(lazy.params, eager.params)
    assert torch.equal(lazy.exp_avg, eager.exp_avg)
    assert torch.equal(lazy.exp_avg_sq, eager.exp_avg_sq)
    for e in (eager, lazy):
        e.close()


def test_lazy_adam_state_machine():
    import torch
    from gpu_utils import TOY_FORMULAS, TOY_KINDS, TOY_SIZES, engine_from_params, plan_for, random_params
    from graphqembed_amd.engine import GqeError
    from graphqembed_amd.tensorize import pack_margin_batches
    rng = np.random.RandomState(2)      # min-simple: two margin calls accumulating into a Pre gradient would add in atomic order
    d = 32
    params = random_params(rng, d, "bilinear-diag", "min-simple", TOY_SIZES, TOY_KINDS)
    eng = engine_from_params(params, d, "bilinear-diag", "min-simple", lazy_adam=True)
    ref = engine_from_params(params, d, "bilinear-diag", "min-simple")
    plan_l, plan_r = plan_for(eng, "2-inter", TOY_FORMULAS["2-inter"]), plan_for(ref, "2-inter", TOY_FORMULAS["2-inter"])
    for step in range(5):
        t, g, a = _disjoint_batch(rng, "2-inter", 8, 0.0, 0.5)
        for e, pl in ((eng, plan_l), (ref, plan_r)):
            descs, idx, n = pack_margin_batches([(pl, t, g, a, 1.0, 1.0)])
            e.margin_fwd_bwd(descs, idx, n)
            # a different learning rate half way: the deferred steps are settled with the old one first
            e.adam_step(pl.touched, 0.01 if step < 3 else 0.003)
    # two margin calls before one step: the lists of the first are not named by the second call's feed -> full pass
    for _ in range(2):
        t, g, a = _disjoint_batch(rng, "2-inter", 8, 0.5, 1.0)
        for e, pl in ((eng, plan_l), (ref, plan_r)):
            descs, idx, n = pack_margin_batches([(pl, t, g, a, 1.0, 1.0)])
            e.margin_fwd_bwd(descs, idx, n)
    for e, pl in ((eng, plan_l), (ref, plan_r)):
        e.adam_step(pl.touched, 0.003)
    # SGD after Adam: rows settle their Adam debt first
    t, g, a = _disjoint_batch(rng, "2-inter", 8, 0.0, 1.0)
    for e, pl in ((eng, plan_l), (ref, plan_r)):
        descs, idx, n = pack_margin_batches([(pl, t, g, a, 1.0, 1.0)])
        e.margin_fwd_bwd(descs, idx, n)
        e.sgd_step(pl.touched, 0.05)
    assert torch.equal(eng.params, ref.params) and torch.equal(eng.exp_avg_sq, ref.exp_avg_sq)
    # a resumed run hands in its own step counts: accepted when every row is current (the row counters are re-based)
    eng.sync()
    eng.steps = {k: v + 1000 for k, v in eng.steps.items()}
    ref.steps = dict(eng.steps)
    t, g, a = _disjoint_batch(rng, "2-inter", 8, 0.0, 1.0)
    for e, pl in ((eng, plan_l), (ref, plan_r)):
        descs, idx, n = pack_margin_batches([(pl, t, g, a, 1.0, 1.0)])
        e.margin_fwd_bwd(descs, idx, n)
        e.adam_step(pl.touched, 0.003)
    assert torch.equal(eng.params, ref.params) and torch.equal(eng.exp_avg, ref.exp_avg)
    # leaving lazy mode needs a sync
    t, g, a = _disjoint_batch(rng, "2-inter", 8, 0.0, 0.5)
    descs, idx, n = pack_margin_batches([(plan_l, t, g, a, 1.0, 1.0)])
    eng.margin_fwd_bwd(descs, idx, n)
    eng.adam_step(plan_l.touched, 0.003)
    assert eng.lib.gqe_set_lazy_adam(eng.ctx, 0) != 0
    eng.sync()
    assert eng.lib.gqe_set_lazy_adam(eng.ctx, 0) == 0
    eng.close()
    ref.close()


@pytest.mark.parametrize("dec,inter", [("bilinear-diag", "min"), ("bilinear", "mean")])
def test_full_size_properties(dec, inter):
    """Size-independent properties at the BASELINE sizes (full mix, B=512, d=128), no oracle involved:
      * linearity: loss weights x2 -> every gradient x2 (a power of two: only summation order can differ);
      * batching: the same batches twice in one grouped launch == one launch called twice == 2 x the gradient;
      * permutation: shuffling the queries of a batch permutes its scores and leaves loss and gradients alone;
      * a forward of the same queries reproduces the margin call's positive scores."""
    import torch
    from gpu_utils import (TOY_FORMULAS, TOY_KINDS, TOY_SIZES, engine_from_params, plan_for, random_params, read_arena,
                           toy_batch)
    from graphqembed_amd.tensorize import pack_forward_batches, pack_margin_batches
    d, B = 128, 512
    mix = [("1-chain", 1.0), ("2-chain", 0.01), ("3-chain", 0.01), ("2-inter", 0.005), ("3-inter", 0.005), ("3-inter_chain", 0.005),
           ("3-chain_inter", 0.005)]
    rng = np.random.RandomState(17)
    params = random_params(rng, d, dec, inter, TOY_SIZES, TOY_KINDS)
    eng = engine_from_params(params, d, dec, inter, max_queries=2 * B * len(mix), max_batches=2 * len(mix))
    base = []
    for qt, w in mix:
        t, g, a = toy_batch(rng, qt, B)
        base.append((plan_for(eng, qt, TOY_FORMULAS[qt]), t, g, a, w, 1.0))

    def run(items):
        eng.zero_grads(list(eng.layout.entries))
        descs, idx, n = pack_margin_batches(items)
        losses, pos, neg = eng.margin_fwd_bwd(descs, torch.from_numpy(idx).cuda(), n, want_scores=True)
        grads = read_arena(eng, eng.grads)
        return losses.cpu().numpy(), pos.cpu().numpy(), neg.cpu().numpy(), grads

    def close(a, b, what, factor=1.0):
        for k in a:
            scale = max(1e-12, float(np.abs(b[k]).max()))
            np.testing.assert_allclose(a[k], factor * b[k], rtol=0, atol=3e-5 * factor * scale, err_msg="%s %s" % (what, k))

    l1, p1, n1, g1 = run(base)
    # linearity in the loss weight
    l2, p2, n2, g2 = run([(pl, t, g, a, 2.0 * w, m) for (pl, t, g, a, w, m) in base])
    assert np.array_equal(p1, p2) and np.array_equal(n1, n2)
    np.testing.assert_allclose(l2[:-1], l1[:-1], rtol=1e-6)          # per-batch mean losses do not carry the weight
    np.testing.assert_allclose(l2[-1], 2.0 * l1[-1], rtol=1e-5)
    close(g2, g1, "weights x2", 2.0)
    # the batches twice in one grouped launch
    l3, p3, n3, g3 = run(base + base)
    assert np.array_equal(p3[: len(p1)], p1) and np.array_equal(p3[len(p1):], p1)
    close(g3, g1, "batches twice", 2.0)
    # permutation of the queries inside every batch
    perms = [rng.permutation(B) for _ in base]
    l4, p4, n4, g4 = run([(pl, t[pm], g[pm], a[:, pm], w, m) for (pl, t, g, a, w, m), pm in zip(base, perms)])
    off = 0
    for pm in perms:
        assert np.array_equal(p4[off:off + B], p1[off:off + B][pm]) and np.array_equal(n4[off:off + B], n1[off:off + B][pm])
        off += B
    np.testing.assert_allclose(l4, l1, rtol=2e-6)
    close(g4, g1, "permuted queries")
    # forward() of the same (target, anchors) = the positive scores of the margin call
    descs, idx, n = pack_forward_batches([(pl, t, a) for (pl, t, g, a, w, m) in base])
    np.testing.assert_allclose(eng.forward(descs, idx, n).cpu().numpy(), p1, rtol=0, atol=1e-6)   # (the training variant
    eng.close()                                                             # of the Bilinear chain contracts in another order)


def test_lazy_adam_with_a_bag_mode():
    """An EmbeddingBag mode next to ordinary tables (the Reddit shape): the bag table is stepped in full every
    iteration (its gradient lists hang on word rows no index feed names), the other tables lazily, in the same
    step.  With bags no run is reproducible bit for bit (word rows sum many contributions in atomic order and Adam
    turns that noise into lr-sized steps — two EAGER engines drift apart by ~0.2 over these 60 steps), so the yardstick
    is a second eager engine: the lazy engine must stay as close to eager #1 as eager #2 does."""
    import torch
    from gpu_utils import TOY_FORMULAS, TOY_KINDS, TOY_SIZES, engine_from_params, plan_for, random_params, toy_batch
    from graphqembed_amd.tensorize import pack_margin_batches
    rng = np.random.RandomState(8)
    d, dec, inter = 64, "bilinear-diag", "min"
    params = random_params(rng, d, dec, inter, TOY_SIZES, TOY_KINDS, bag_modes=("b",))
    eager = engine_from_params(params, d, dec, inter)
    eager2 = engine_from_params(params, d, dec, inter)
    lazy = engine_from_params(params, d, dec, inter, lazy_adam=True)
    types = ["1-chain", "2-inter", "2-chain", "3-inter_chain", "3-chain_inter"]
    for step in range(60):
        qt = types[step % len(types)]
        t, g, a = toy_batch(rng, qt, 12)
        t[:], g[:] = np.minimum(t, 30), np.minimum(g, 30)      # keep most rows of the big tables untouched: they must lag
        outs = []
        for eng in (eager, eager2, lazy):
            plan = plan_for(eng, qt, TOY_FORMULAS[qt])
            descs, idx, n = pack_margin_batches([(plan, t, g, a, 1.0, 1.0)])
            losses, _, _ = eng.margin_fwd_bwd(descs, idx, n)
            eng.adam_step(plan.touched, 0.01)
            outs.append(float(losses[-1].item()))
        if step == 1:      # before the noise has had time to grow, everything still agrees tightly
            assert float((lazy.params - eager.params).abs().max()) < 1e-5
        assert abs(outs[2] - outs[0]) <= 0.25 * abs(outs[0]) + 1e-3, (step, outs)      # same loss curve, loosely
    torch.cuda.synchronize()
    assert not torch.equal(lazy._params, eager._params)            # rows of the ordinary tables do lag
    noise = (eager2.params - eager.params).abs()
    diff = (lazy.params - eager.params).abs()
    assert float(diff.max()) <= 3.0 * float(noise.max()) + 0.02, (float(diff.max()), float(noise.max()))
    # (one realisation of the noise against another: 3.2 x was seen once; exactness of the bag gradients themselves is pinned
    # by test_golden_reddit_embedding_bag and test_reddit_synth_config5_full_size)
    assert float(diff.mean()) <= 5.0 * float(noise.mean()) + 1e-4, (float(diff.mean()), float(noise.mean()))
    for e in (eager, eager2, lazy):
        e.close()


def _second_step_vs_oracle(eng, wl, items, descs, didx, n, keys, dec, inter, mode):
    """One more iteration on the same batches, stepped the way ``mode`` says ("two-call": gqe_margin_fwd_bwd + gqe_adam_step;
    "deferred": the same with gqe_set_deferred_gemm — the pair-GEMM units ride in the Adam pass; "train-step": gqe_train_step —
    the split step where it applies), and the PARAMETERS AFTER IT against the oracle: O.adam_step (fp64) on the oracle's gradient at
    the device's parameters / moments of the moment.  Rows the iteration names, vectors, matrices: elements whose gradient is
    signal (|g| > 1e-4 max|g|, or exactly 0) within 2e-3 of the lr-sized move, at most 0.2 % of them (or two rows of a d x d matrix) on
    the other side of a relu / arg-min decision; rows it does not name: the zero-gradient Adam formula to 1e-4 of the move (v_sqrt_f32 / v_rcp_f32 are 1 ulp)."""
    import torch
    entries = eng.layout.entries
    # (the caller folded the lists with gqe_materialize_grads: until a pass consumes them the tables' DENSE gradients are
    # authoritative, and a step that has to read them neither splits nor carries riding units — consume them)
    eng.zero_grads(list(entries))
    torch.cuda.synchronize()

    def host(flat):
        h = flat.cpu().numpy()
        return {k: h[off:off + int(np.prod(shape))].reshape(shape).astype(np.float64) for k, (off, shape) in entries.items()}
    before, m0, v0 = host(eng.params), host(eng.exp_avg), host(eng.exp_avg_sq)
    oparams = dict(before)
    oparams[O.BAGS_KEY] = {m: csr for m, csr in wl.g.bags.items()}
    ograds = {k: np.zeros(v.shape, dtype=np.float64) for k, v in before.items()}
    ograds[O.BAGS_KEY] = oparams[O.BAGS_KEY]
    want_l = []
    for (f, t, ng, a, w, m) in items:
        l, _, _, _ = O.margin_fwd_bwd(oparams, O.make_plan(f.query_type, f.rels), dec, inter, t, ng, a, margin=m, weight=w, grads=ograds)
        want_l.append(l)
    steps = {k: int(eng.steps[k]) for k in keys}
    ostate = {k: {"step": steps[k], "m": m0[k].copy(), "v": v0[k].copy()} for k in keys}
    expect = {k: before[k].copy() for k in keys}
    O.adam_step(expect, ograds, ostate, keys)
    rides0, splits0 = eng.gemm_rides(), eng.split_steps()
    if mode == "deferred":
        eng.set_deferred_gemm(True)
    if mode == "train-step":
        losses = eng.train_step(descs, didx, keys)
    else:
        losses, _, _ = eng.margin_fwd_bwd(descs, didx, n)
        eng.adam_step(keys)
    np.testing.assert_allclose(losses.cpu().numpy()[:-1], want_l, rtol=LOSS_RTOL, err_msg=mode)
    after = host(eng.params)
    for k in sorted(keys):
        g = np.abs(ograds[k])
        move = np.abs(expect[k] - before[k])
        diff = np.abs(after[k] - expect[k])
        if k.startswith("enc.") and k not in eng.bag_keys:
            named = g.reshape(g.shape[0], -1).max(axis=1) > 0
            quiet = ~named
            assert quiet.sum() > 100 and named.sum() > 100, k
            # (a named row whose contributions cancel to an exact 0 in the oracle keeps ~1e-13 of cancellation noise on the
            # device, and Adam turns g / (|g| + eps) into a visible fraction of lr: 1e-5 absolute covers it; rows the feed does
            # not name at all are held BIT-equal to the eager pass in tests/test_gpu_split.py)
            assert (diff[quiet] <= 1e-4 * move[quiet] + 1e-5).all(), (mode, k, float(diff[quiet].max()))
            assert np.median(diff[quiet]) <= 1e-9, (mode, k, float(np.median(diff[quiet])))
            g, move, diff = g[named], move[named], diff[named]
        signal = (g == 0) | (g > 1e-4 * g.max())
        bad = signal & (diff > 2e-3 * move + 2e-7)
        # (a d x d matrix: ONE query on the other side of a relu / arg-min decision changes a whole row or column of its gradient —
        # d elements at once, 0.39 % of a 256 x 256 matrix: two such rows are allowed whatever the 0.2 % comes to)
        allowed = max(2, 2e-3 * signal.sum())
        if not k.startswith("enc.") and after[k].ndim == 2:
            allowed = max(allowed, 2 * after[k].shape[1])
        # (a relation VECTOR's gradient is one sum over the batches that use the relation: a single query on the other side of a
        # decision shifts every element a little — seen once in ten full runs: 149 of 256 elements off by up to 3 % of the move;
        # such a vector passes if the whole of it stays within 5 % of its largest move)
        shifted = after[k].ndim == 1 and float(diff[signal].max()) <= 5e-2 * float(move.max())
        assert bad.sum() <= allowed or shifted, (mode, k, int(bad.sum()), int(signal.sum()), float(diff[signal].max()), float(move.max()))
    for k in set(after) - set(keys):
        assert np.array_equal(after[k], before[k]), (mode, k)
    eng.materialize()
    assert float(eng.grads.abs().max()) == 0.0
    if mode == "deferred" and d_rides(eng):
        assert eng.gemm_rides() == rides0 + 1, "the pair GEMM was expected to ride in the Adam pass"
    if mode == "train-step" and not eng.bag_keys and eng.dim % 64 == 0:
        assert eng.split_steps() == splits0 + 1, "gqe_train_step was expected to run as a split step"
    if mode == "train-step" and eng.bag_keys and d_rides(eng):
        # the two-call sequence inside gqe_train_step defers the pair GEMM: next to the non-temporal pass over tables beyond the
        # Infinity Cache its units ride SPREAD through the pass's grid when GQE_RIDE_SPREAD=1 (gqe_dev.h, GqeGemmRide.spread)
        assert eng.gemm_rides() == rides0 + 1, "the pair GEMM was expected to ride in the Adam pass"
    if mode == "deferred":
        eng.set_deferred_gemm(False)


def d_rides(eng):
    """can this engine's Adam pass carry the pair-GEMM units?  (d % 64 == 0; in front of its chunks when the tables are inside the
    Infinity Cache, spread through its grid when they are beyond it and GQE_RIDE_SPREAD=1 asks for it: include/gqe.h)"""
    stream = 12 * sum(int(np.prod(shape)) for k, (off, shape) in eng.layout.entries.items() if k.startswith("enc."))
    return eng.dim % 64 == 0 and (stream <= (192 << 20) or os.environ.get("GQE_RIDE_SPREAD", "0") != "0")


def _full_size_vs_oracle(workload, d, dec, inter, min_params, n_relations, zipf=None, min_longest_list=0, mode="two-call"):
    """One full-mix iteration (9 x 512 queries, ONE grouped launch, index feed resident in HBM) of a BASELINE workload at its
    REAL table sizes against the fp64 oracle (the oracle only gathers the rows a batch names, so it stays cheap):
      * scores and losses;
      * gradients on every row the iteration touches (and nothing elsewhere), relation / Pre / Post gradients in full;
      * linearity (weights x 2), and a fused Adam step that leaves no gradient behind and moves every row that had one
        (including, for bag modes, the rows of the word table, which are reached only through bags)."""
    import torch
    import bench
    from graphqembed_amd import synth
    from graphqembed_amd.tensorize import FormulaPlan, pack_margin_batches, table_key
    B = 512
    wl = bench.Workload(workload, d, dec, inter, synth.FULL_MIX, B, n_distinct=1, zipf=zipf)
    assert wl.layout.total >= min_params and sum(len(v) for v in wl.g.relations.values()) == n_relations
    eng = wl.engine()
    items = wl.item_sets[0]
    if min_longest_list:     # a skewed batch: some row collects hundreds of contributions
        from collections import Counter
        c = Counter()
        for (f, t, ng, a, w, m) in items:
            if f.target_mode not in wl.g.bags:
                c.update((f.target_mode, int(x)) for x in np.concatenate([t, ng]))
            for i, am in enumerate(f.anchor_modes):
                if am not in wl.g.bags:
                    c.update((am, int(x)) for x in a[i])
        assert c.most_common(1)[0][1] >= min_longest_list, c.most_common(3)
    host = eng.params.cpu().numpy()
    params = {k: host[off:off + int(np.prod(shape))].reshape(shape) for k, (off, shape) in eng.layout.entries.items()}
    params[O.BAGS_KEY] = {m: csr for m, csr in wl.g.bags.items()}
    grads = {k: np.zeros(v.shape, dtype=np.float32) for k, v in params.items() if k != O.BAGS_KEY}
    grads[O.BAGS_KEY] = params[O.BAGS_KEY]
    packed, want_l, want_p, want_n = [], [], [], []
    for (f, t, ng, a, w, m) in items:
        packed.append((FormulaPlan(f, eng.layout, inter), t, ng, a, w, m))
        l, sp, sn, _ = O.margin_fwd_bwd(params, O.make_plan(f.query_type, f.rels), dec, inter, t, ng, a, margin=m, weight=w, grads=grads)
        want_l.append(l)
        want_p.append(sp)
        want_n.append(sn)
    descs, idx, n = pack_margin_batches(packed)
    didx = torch.from_numpy(idx).cuda()
    losses, pos, neg = eng.margin_fwd_bwd(descs, didx, n, want_scores=True)
    np.testing.assert_allclose(pos.cpu().numpy(), np.concatenate(want_p), atol=SCORE_ATOL, rtol=1e-4)
    np.testing.assert_allclose(neg.cpu().numpy(), np.concatenate(want_n), atol=SCORE_ATOL, rtol=1e-4)
    np.testing.assert_allclose(losses.cpu().numpy()[:-1], want_l, rtol=LOSS_RTOL)
    eng.materialize()
    g1 = eng.grads.clone()
    got = g1.cpu().numpy()
    for k, (off, shape) in eng.layout.entries.items():
        gk, wk = got[off:off + int(np.prod(shape))].reshape(shape), grads[k]
        scale = max(float(np.abs(wk).max()), 1e-12)
        if k.startswith("enc."):
            rows = np.flatnonzero(np.abs(wk).max(axis=1) > 0)
            assert len(rows) > 100, k
            np.testing.assert_allclose(gk[rows], wk[rows], rtol=3e-3, atol=3e-5 * scale, err_msg=k)
            rest = np.ones(shape[0], dtype=bool)
            rest[rows] = False
            # untouched rows hold nothing; a touched row whose contributions cancel to an exact 0 in the oracle may keep
            # cancellation noise on the device (seen: 1e-13 against gradients of 1e-4)
            assert float(np.abs(gk[rest]).max()) <= 1e-7 * scale, k
        else:
            np.testing.assert_allclose(gk, wk, rtol=3e-3, atol=3e-5 * scale, err_msg=k)
    # linearity in the loss weight
    eng.zero_grads(list(eng.layout.entries))
    descs2, _, _ = pack_margin_batches([(pl, t, ng, a, 2.0 * w, m) for (pl, t, ng, a, w, m) in packed])
    eng.margin_fwd_bwd(descs2, didx, n)
    eng.materialize()
    assert float((eng.grads - 2.0 * g1).abs().max()) <= 3e-5 * 2.0 * float(g1.abs().max())
    # a fused Adam step: no gradient left anywhere, every row with a gradient moved (a first step leaves zero-gradient
    # rows where they are: lr * 0), nothing became non-finite
    keys = set().union(*[p[0].touched for p in packed])
    before = eng.params.clone()
    eng.adam_step(keys)
    eng.materialize()
    assert float(eng.grads.abs().max()) == 0.0
    moved = (eng.params != before)
    for k in keys:
        off, shape = eng.layout.entries[k]
        if not k.startswith("enc."):
            continue
        gmax = np.abs(grads[k]).reshape(shape[0], -1).max(axis=1)       # (a gradient of 1e-16 moves nothing: lr * g / (|g| + 1e-8))
        touched = torch.from_numpy(gmax > 1e-4 * gmax.max()).cuda()
        m_k = moved[off:off + int(np.prod(shape))].view(shape[0], -1).any(dim=1)
        bad = torch.nonzero(touched & ~m_k).flatten()[:4].tolist()
        assert not bad, (k, bad, [float(gmax[r]) for r in bad], float(gmax.max()))
    assert bool(torch.isfinite(eng.params).all())
    # ... and the VALUES of a step: one more iteration, stepped as `mode` says, against the oracle's Adam step
    _second_step_vs_oracle(eng, wl, items, descs, didx, n, keys, dec, inter, mode)
    if zipf:
        # The Adam pass above walked the hub rows' long lists and PROMOTED them (include/gqe.h, gqe_hot_rows): from now on their
        # contributions are added into dense accumulators with float atomics instead of being linked.  Two more steps through
        # that path: gradients against the oracle on the parameters of the moment (materialize folds accumulators and lists
        # alike), then a step that must consume every accumulator.
        assert eng.hot_rows() >= 4, eng.hot_rows()
        if wl.g.bags:
            # hot WORD rows got sub-lists (include/gqe.h, gqe_hot_sub_lists): the two steps below link the bags' nodes onto them
            # and the gather launch behind the fused kernel sums them into the accumulators
            heads, on = eng.hot_sub_lists()
            assert heads >= 2 * eng.hot_rows() // 3 and on == (os.environ.get("GQE_HOT_SUB", "1") != "0"), (heads, on, eng.hot_rows())
        for rnd in range(2):
            host = eng.params.cpu().numpy()
            params = {k: host[off:off + int(np.prod(shape))].reshape(shape) for k, (off, shape) in eng.layout.entries.items()}
            params[O.BAGS_KEY] = {m: csr for m, csr in wl.g.bags.items()}
            grads = {k: np.zeros(v.shape, dtype=np.float32) for k, v in params.items() if k != O.BAGS_KEY}
            grads[O.BAGS_KEY] = params[O.BAGS_KEY]
            want_l = []
            for (f, t, ng, a, w, m) in items:
                l, _, _, _ = O.margin_fwd_bwd(params, O.make_plan(f.query_type, f.rels), dec, inter, t, ng, a, margin=m, weight=w, grads=grads)
                want_l.append(l)
            losses, _, _ = eng.margin_fwd_bwd(descs, didx, n)
            np.testing.assert_allclose(losses.cpu().numpy()[:-1], want_l, rtol=LOSS_RTOL)
            if rnd == 0:
                eng.materialize()
                got = eng.grads.cpu().numpy()
                for k, (off, shape) in eng.layout.entries.items():
                    gk, wk = got[off:off + int(np.prod(shape))].reshape(shape), grads[k]
                    scale = max(float(np.abs(wk).max()), 1e-12)
                    np.testing.assert_allclose(gk, wk, rtol=3e-3, atol=3e-5 * scale, err_msg="hot-row step: " + k)
            before = eng.params.clone()
            eng.adam_step(keys)
            eng.materialize()
            assert float(eng.grads.abs().max()) == 0.0                       # nothing left on a list or in an accumulator
            if rnd == 1:   # a step fed by accumulators alone (no materialised gradient): the hub rows moved
                for k in keys:
                    if not k.startswith("enc."):
                        continue
                    off, shape = eng.layout.entries[k]
                    gmax = np.abs(grads[k]).reshape(shape[0], -1).max(axis=1)
                    hub = int(np.argmax(np.abs(grads[k]).reshape(shape[0], -1).sum(axis=1)))
                    if gmax[hub] > 0:
                        assert bool((eng.params[off + hub * shape[1]: off + (hub + 1) * shape[1]] != before[off + hub * shape[1]: off + (hub + 1) * shape[1]]).any()), (k, hub)
        assert bool(torch.isfinite(eng.params).all())
    eng.close()


STEP_MODES = ["two-call", "deferred", "train-step"]   # gqe_margin_fwd_bwd + gqe_adam_step | with gqe_set_deferred_gemm | gqe_train_step


@pytest.mark.parametrize("mode", ["two-call", "train-step"])
def test_reddit_synth_config5_full_size(mode):
    """BASELINE config 5 at its real size: reddit-synth (500 k users / 400 k posts / 2 k communities, the 12 directed
    relations of reddit/data_utils_new.py:193-197, posts = EmbeddingBag mean over 5..30 of 50 k words), d=256.
    (gqe_train_step runs the two-call sequence here: bag tables, tables beyond the Infinity Cache.)"""
    _full_size_vs_oracle("reddit-synth", 256, "bilinear-diag", "min", 141 * 10 ** 6, 12, mode=mode)


@pytest.mark.parametrize("mode", STEP_MODES)
@pytest.mark.parametrize("dec,inter,P", [("bilinear-diag", "min", 12582912), ("bilinear", "mean", 12810496)])
def test_bio_synth_configs_full_size(dec, inter, P, mode):
    """BASELINE configs 2-4 at their real size: bio-synth (97 000 nodes in 5 modes, 14 directed relations), d=128,
    P = 12 582 912 (bilinear-diag + SetIntersection) / 12 810 496 (full Bilinear) parameters — the workload bench.py times,
    stepped every way the library offers: the timed path (gqe_train_step's split step) is pinned to the oracle here."""
    _full_size_vs_oracle("bio-synth", 128, dec, inter, P, 14, mode=mode)


@pytest.mark.parametrize("mode", ["two-call", "train-step"])
def test_bio_synth_zipf_full_size(mode):
    """The full mix on a HEAVY-TAILED bio-synth graph (node degrees ~ 1 / rank: the data shape the reference was written for,
    graph.py:108-122): hub rows collect 100+ gradient contributions in one step.  Same checks as the uniform case."""
    _full_size_vs_oracle("bio-synth", 128, "bilinear-diag", "min", 12582912, 14, zipf=1.0, min_longest_list=100, mode=mode)


def test_reddit_synth_zipf_full_size():
    """reddit-synth with Zipfian word frequencies and node degrees (a real vocabulary: reddit/data_utils_new.py:155,162-169):
    the most frequent word row is linked by thousands of bag contributions in one step."""
    _full_size_vs_oracle("reddit-synth", 256, "bilinear-diag", "min", 141 * 10 ** 6, 12, zipf=1.0, min_longest_list=50)
