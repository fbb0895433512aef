"""The oracle (oracle/netquery_numpy.py) against every golden vector produced by
the reference itself (oracle/make_golden.py).  CPU only."""
import copy
import glob
import json
import os

import numpy as np
import pytest

from golden_utils import (GOLDEN, adam1_files, case_names, eval_calls, eval_files, load_case, load_params, load_reddit_params,
                          load_tables, model_files, reddit_files)
from oracle import netquery_numpy as O

# fp32 reference vs fp64 oracle
SCORE_ATOL = 2e-6
LOSS_RTOL = 2e-6
GRAD_RTOL, GRAD_ATOL = 2e-4, 2e-7


@pytest.mark.parametrize("path,dec,inter,d", model_files(), ids=lambda v: os.path.basename(v) if isinstance(v, str) and v.endswith(".npz") else None)
def test_scores_loss_grads(path, dec, inter, d):
    z = np.load(path)
    params = load_params(z, d)
    names = case_names(z)
    assert names
    for case in names:
        c = load_case(z, case)
        plan = O.make_plan(c["type"], c["rels"])
        pos = O.forward_scores(params, plan, dec, inter, c["target"], c["anchors"])
        neg = O.forward_scores(params, plan, dec, inter, c["neg"], c["anchors"])
        np.testing.assert_allclose(pos, c["pos"], atol=SCORE_ATOL, rtol=1e-5, err_msg=case)
        np.testing.assert_allclose(neg, c["negscore"], atol=SCORE_ATOL, rtol=1e-5, err_msg=case)
        loss, sp, sn, grads = O.margin_fwd_bwd(params, plan, dec, inter, c["target"], c["neg"], c["anchors"],
                                               margin=c["margin"])
        np.testing.assert_allclose(sp, c["pos"], atol=SCORE_ATOL, rtol=1e-5)
        np.testing.assert_allclose(loss, c["loss"], rtol=LOSS_RTOL, atol=1e-7, err_msg=case)
        touched = O.touched_keys(plan, dec, inter)
        assert touched == set(c["grads"].keys()), case
        for k, g in c["grads"].items():
            scale = max(np.abs(g).max(), 1e-12)
            np.testing.assert_allclose(grads[k], g, rtol=GRAD_RTOL, atol=GRAD_ATOL + 1e-5 * scale,
                                       err_msg="%s %s" % (case, k))
        for k in set(grads) - touched:
            assert not grads[k].any(), (case, k)


@pytest.mark.parametrize("path,dec,inter,d", model_files(32), ids=lambda v: os.path.basename(v) if isinstance(v, str) and v.endswith(".npz") else None)
def test_adam_three_steps(path, dec, inter, d):
    z = np.load(path)
    p0 = load_params(z, d)
    for case in case_names(z):
        c = load_case(z, case)
        if "adam_neg" not in c:
            continue
        plan = O.make_plan(c["type"], c["rels"])
        params = {k: v.astype(np.float64) for k, v in p0.items()}
        state = {}
        touched = O.touched_keys(plan, dec, inter)
        for step in range(3):
            loss, _, _, grads = O.margin_fwd_bwd(params, plan, dec, inter, c["target"], c["adam_neg"][step],
                                                 c["anchors"], margin=c["margin"])
            # step 0 is exact; later steps inherit Adam's noise amplification (see below)
            np.testing.assert_allclose(loss, c["adam_loss"][step], rtol=2e-6 if step == 0 else 2e-2, err_msg=case)
            O.adam_step(params, grads, state, touched)
        assert touched == set(c["adam_delta"].keys())
        for k, delta in c["adam_delta"].items():
            got = params[k] - p0[k].astype(np.float64)
            # Adam's first steps are sign-like: dp = lr * g / (|g| + 1e-8).  An element whose exact
            # gradient is 0 but whose fp32 gradient is rounding noise (~1e-9, e.g. the cancelling
            # projection g - xhat (xhat.g)) moves by a visible fraction of lr in the reference, and
            # min/relu then flip discretely.  So: nearly every element must agree tightly, a few
            # noise-driven outliers may differ by a few lr.
            diff = np.abs(got - delta)
            assert diff.max() < 4e-2, (case, k, diff.max())
            assert np.median(diff) < 5e-4, (case, k, np.median(diff))
        for k in set(params) - touched:
            assert np.array_equal(params[k], p0[k].astype(np.float64))


def test_adam_single_step_formula():
    """One torch.optim.Adam step on random (p, g, m, v, step) against the restatement."""
    import torch
    rng = np.random.RandomState(0)
    p = rng.randn(50, 8).astype(np.float32)
    params = {"x": p.astype(np.float64)}
    tp = torch.nn.Parameter(torch.from_numpy(p.copy()))
    opt = torch.optim.Adam([tp], lr=0.01)
    state = {}
    for step in range(4):
        g = (rng.randn(50, 8) * 10 ** rng.uniform(-6, 0, size=(50, 8))).astype(np.float32)
        g[rng.rand(50, 8) < 0.3] = 0
        tp.grad = torch.from_numpy(g.copy())
        opt.step()
        O.adam_step(params, {"x": g.astype(np.float64)}, state, ["x"])
        np.testing.assert_allclose(params["x"], tp.detach().numpy(), rtol=0, atol=3e-6)


@pytest.mark.parametrize("path,dec,inter,d", model_files(), ids=lambda v: os.path.basename(v) if isinstance(v, str) and v.endswith(".npz") else None)
def test_torch_port_matches_golden(path, dec, inter, d):
    """oracle/netquery_torch.py (bench.py's cpu_baseline) reproduces the reference's scores,
    loss and gradients."""
    from oracle.netquery_torch import TorchPort
    z = np.load(path)
    params = load_params(z, d)
    for case in case_names(z):
        c = load_case(z, case)
        port = TorchPort(params, dec, inter)
        plan = O.make_plan(c["type"], c["rels"])
        loss = port.margin_loss(plan, c["target"], c["neg"], c["anchors"], c["margin"])
        loss.backward()
        np.testing.assert_allclose(loss.item(), c["loss"], rtol=1e-6, err_msg=case)
        got = port.grads()
        for k, g in c["grads"].items():
            np.testing.assert_allclose(got[k], g, rtol=1e-4, atol=1e-7 + 1e-5 * np.abs(g).max(), err_msg="%s %s" % (case, k))
        assert set(k for k, g in got.items() if g is not None) == set(c["grads"].keys())


@pytest.mark.parametrize("path,dec,inter,d", reddit_files(), ids=lambda v: os.path.basename(v) if isinstance(v, str) and v.endswith(".npz") else None)
def test_reddit_embedding_bag_cases(path, dec, inter, d):
    """Posts are an nn.EmbeddingBag (mean over word rows) in the reference: scores, loss, gradients (numpy
    oracle and torch port) and the 3-step Adam trajectory."""
    from oracle.netquery_torch import TorchPort
    z = np.load(path)
    params = load_reddit_params(z)
    names = case_names(z)
    assert len(names) == (9 if d == 32 else 11)          # d=128: hard negatives for all four intersection types
    assert d == 32 or {"3-inter.hard", "3-inter_chain.hard"} <= set(names)
    for case in names:
        c = load_case(z, case)
        plan = O.make_plan(c["type"], c["rels"])
        loss, sp, sn, grads = O.margin_fwd_bwd(params, plan, dec, inter, c["target"], c["neg"], c["anchors"], margin=c["margin"])
        np.testing.assert_allclose(sp, c["pos"], atol=SCORE_ATOL, rtol=1e-5, err_msg=case)
        np.testing.assert_allclose(sn, c["negscore"], atol=SCORE_ATOL, rtol=1e-5, err_msg=case)
        np.testing.assert_allclose(loss, c["loss"], rtol=LOSS_RTOL, atol=1e-7, err_msg=case)
        assert O.touched_keys(plan, dec, inter) == set(c["grads"].keys()), case
        port = TorchPort(params, dec, inter)
        tl = port.margin_loss(plan, c["target"], c["neg"], c["anchors"], c["margin"])
        tl.backward()
        tg = port.grads()
        for k, g in c["grads"].items():
            scale = max(np.abs(g).max(), 1e-12)
            np.testing.assert_allclose(grads[k], g, rtol=GRAD_RTOL, atol=GRAD_ATOL + 1e-5 * scale, err_msg="%s %s" % (case, k))
            np.testing.assert_allclose(tg[k], g, rtol=1e-4, atol=1e-7 + 1e-5 * scale, err_msg="torch port %s %s" % (case, k))
        if "adam_neg" not in c:
            continue
        # Adam trajectory
        p = {k: (v.astype(np.float64) if k != O.BAGS_KEY else v) for k, v in params.items()}
        state = {}
        touched = O.touched_keys(plan, dec, inter)
        for step in range(3):
            l, _, _, g = O.margin_fwd_bwd(p, plan, dec, inter, c["target"], c["adam_neg"][step], c["anchors"], margin=c["margin"])
            np.testing.assert_allclose(l, c["adam_loss"][step], rtol=2e-6 if step == 0 else 3e-2, err_msg=case)
            O.adam_step(p, g, state, touched)
        for k, delta in c["adam_delta"].items():
            diff = np.abs(p[k] - params[k].astype(np.float64) - delta)
            assert diff.max() < 4e-2 and np.median(diff) < 5e-4, (case, k)


_npz_id = lambda v: os.path.basename(v) if isinstance(v, str) and v.endswith(".npz") else None


@pytest.mark.parametrize("path,dec,inter,d", eval_files(), ids=_npz_id)
def test_eval_fixture_scores(path, dec, inter, d):
    """Every forward call eval_auc_queries / eval_perc_queries made in the reference (utils.py:35-91), d=32 for three
    decoder families and d=128: the oracle reproduces the scores; the recorded AUC follows from the recorded scores."""
    from graphqembed_amd.utils import _auc
    z = np.load(path)
    params = load_params(z, d)
    by_call = {}
    for n, qtype, rels, target, anchors, scores in eval_calls(z):
        got = O.forward_scores(params, O.make_plan(qtype, rels), dec, inter, target, anchors)
        np.testing.assert_allclose(got, scores, atol=SCORE_ATOL, rtol=1e-5, err_msg="call %d (%s)" % (n, qtype))
        by_call[n] = scores
    assert len(by_call) > 20
    summary = json.loads(str(z["summary"]))
    for tag, info in summary.items():                      # AUC protocol: first half of a call = positives, second half = negatives
        labels, preds = [], []
        for ci in info["auc_calls"]:
            sc = by_call[ci]
            labels += [1] * (len(sc) // 2) + [0] * (len(sc) // 2)
            preds += list(np.nan_to_num(sc))
        assert abs(_auc(np.asarray(labels), np.asarray(preds)) - info["auc"]) < 1e-9, tag


@pytest.mark.parametrize("path,dec,inter,d", adam1_files(), ids=_npz_id)
def test_adam_one_step_from_the_golden_gradient(path, dec, inter, d):
    """adam1_*.npz: the reference's gradient and its parameters after ONE torch.optim.Adam step — the restated Adam
    (oracle adam_step, the formula the device kernel implements) maps one onto the other to fp32 rounding."""
    z = np.load(path)
    model = np.load(os.path.join(GOLDEN, "model_%s_%s_d%d.npz" % (dec, inter, d)))
    p0 = load_params(model, d)
    cases = sorted(set(k.split("/")[0] for k in z.files))
    assert len(cases) == 3
    for case in cases:
        grads = {k[len(case) + 6:]: z[k].astype(np.float64) for k in z.files if k.startswith(case + "/grad/")}
        after = {k[len(case) + 7:]: z[k] for k in z.files if k.startswith(case + "/after/")}
        assert set(grads) == set(after) and grads
        params = {k: p0[k].astype(np.float64) for k in grads}
        O.adam_step(params, grads, {}, list(grads))
        for k in grads:
            np.testing.assert_allclose(params[k], after[k], rtol=0, atol=2e-7, err_msg="%s %s" % (case, k))


@pytest.mark.parametrize("dec,inter,d", [("bilinear-diag", "min", 32), ("bilinear", "mean", 32), ("transe", "min-simple", 32), ("bilinear-diag", "min", 128)])
def test_oracle_replays_the_recorded_run_train(dec, inter, d):
    """The reference's own 5-iteration run_train (train_*.npz) through the oracle: weighted sum of the batch losses (1 / 0.01 /
    0.005, train_helpers.py:51,69-72), one Adam step per iteration on the touched tensors only, per-tensor step counters — the
    iteration losses to 1e-6 in float64 and float32 alike, the counters exactly, the ema_loss lines of the recorded log from the
    oracle's losses (update_loss, train_helpers.py:11-17; the average starts over at the phase switch)."""
    import json
    from gpu_utils import ema_series, fixture_iterations, oracle_replay
    from golden_utils import load_params, parse_train_log
    z = np.load(os.path.join(GOLDEN, "train_%s_%s_d%d.npz" % (dec, inter, d)))
    p0 = load_params(z, d)
    its = fixture_iterations(z)
    ref = np.array([float(z["it%d/loss" % i]) for i in range(len(its))])
    log = parse_train_log(json.loads(str(z["log"])))
    assert log["edge_conv"] == 1 and len(log["evals"]) == 2 and log["macro"] is not None and log["improvement"] is not None
    assert all(len(e["scores"]) == 11 for e in log["evals"])
    for dt in (np.float64, np.float32):
        losses, params, steps = oracle_replay(p0, dec, inter, its, dt)
        np.testing.assert_allclose(losses, ref, rtol=1e-6)
        for k in p0:
            assert steps.get(k, 0) == (int(z["touched/" + k]) if "touched/" + k in z.files else 0), k
        ema = ema_series(losses, resets={log["edge_conv"] + 1})
        for i, x in log["iters"]:
            assert abs(ema[i] - x) < 2e-6, (i, ema[i], x)
        if dt is np.float64:       # medians: the bulk of every tensor lands on the recorded parameters (the tail is Adam on rounding noise)
            for k in params:
                diff = np.abs(params[k] - p0[k] - z["delta/" + k])
                assert np.median(diff) < 1e-6, (k, np.median(diff))


def test_trainlong_fixture_is_self_consistent():
    """trainlong_*.npz: 400 losses per seed, the recorded ema_loss lines ARE the moving average of the recorded losses (restart at
    the phase switch), five evaluations of eleven lines each, the macro line is the mean of the last evaluation's AUCs."""
    import glob
    import json
    from gpu_utils import ema_series
    from golden_utils import parse_train_log
    files = sorted(glob.glob(os.path.join(GOLDEN, "trainlong*.npz")))
    assert len(files) == 7          # three decoder families at d = 32, the headline and the full-Bilinear pair at d = 128, the Reddit-shaped world, --opt sgd
    for path in files:
        z = np.load(path)
        meta = json.loads(str(z["meta"]))
        for seed in meta["seeds"]:
            pre = "s%d/" % seed
            loss = z[pre + "loss"]
            log = parse_train_log(json.loads(str(z[pre + "log"])))
            assert len(loss) == meta["max_iter"] == 400 and log["edge_conv"] == 99
            nb = z[pre + "n_batches"]
            assert (nb[:100] == 1).all() and (nb[100:] == 11).all()
            ema = ema_series(loss, resets={100})
            assert [i for i, _ in log["iters"]] == list(range(0, 400, 20))
            for i, x in log["iters"]:
                assert abs(ema[i] - x) < 1e-6
            assert [e["iteration"] for e in log["evals"]] == [100, 100, 200, 300, 399]
            final = log["evals"][-1]["scores"]
            assert len(final) == 11 and abs(np.mean([v[0] for v in final.values()]) - log["macro"]) < 1e-6
            first = np.mean([v[0] for v in log["evals"][0]["scores"].values()])
            assert abs((log["macro"] - first) / first - log["improvement"]) < 2e-5
            assert log["macro"] > first + (0.05 if meta.get("optimizer") == "sgd" else 0.1)      # (training fits what the AUC sets hold: a trajectory with a signal)
