"""Shared helpers for the -m gpu parity tests (everything goes through the C ABI)."""
import numpy as np

from oracle import netquery_numpy as O


def engine_from_params(params, d, dec, inter, **kw):
    import torch
    from graphqembed_amd.engine import ArenaLayout, Engine
    layout = ArenaLayout()
    for k, v in params.items():
        if k != O.BAGS_KEY:
            layout.add(k, v.shape)
    if O.BAGS_KEY in params:
        kw["bags"] = {O.table_key(m): csr for m, csr in params[O.BAGS_KEY].items()}
    eng = Engine(d, dec, inter, layout, **kw)
    load_params(eng, params)
    return eng


def load_params(eng, params):
    import torch
    for k, v in params.items():
        if k == O.BAGS_KEY:
            continue
        eng.layout.view(eng.params, k).copy_(torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)))


def read_arena(eng, flat):
    import torch
    if flat is eng.grads:
        eng.materialize()          # embedding-row gradients live in per-row lists until asked for
    torch.cuda.synchronize()
    host = flat.cpu().numpy()
    out = {}
    for k, (off, shape) in eng.layout.entries.items():
        out[k] = host[off:off + int(np.prod(shape))].reshape(shape).copy()
    return out


def plan_for(eng, qtype, rels):
    from graphqembed_amd.graph import Formula
    from graphqembed_amd.tensorize import FormulaPlan
    return FormulaPlan(Formula(qtype, rels), eng.layout, eng.inter_decoder)


def random_params(rng, d, dec, inter, sizes, kinds, bag_modes=(), n_words=64):
    """Parameters with the reference's shapes/initial distributions for a small schema.  Modes listed in
    ``bag_modes`` get an EmbeddingBag feature: a word table [n_words, d] and a CSR of 3-12 word ids per node
    (row index of node i of such a mode = bag i + 1, bag 0 being a dummy so that toy_batch's 1-based rows work)."""
    params = {}
    bags = {}
    for m, n in sizes.items():
        if m in bag_modes:
            params[O.table_key(m)] = rng.normal(0, 1.0 / d, (n_words, d)).astype(np.float32)
            lens = rng.randint(3, 13, size=n + 2)
            ptr = np.zeros(n + 3, dtype=np.int32)
            ptr[1:] = np.cumsum(lens)
            bags[m] = (ptr, rng.randint(0, n_words, size=int(ptr[-1])).astype(np.int32))
            continue
        params[O.table_key(m)] = rng.normal(0, 1.0 / d, (n + 2, d)).astype(np.float32)
    rels = []
    for (a, name, b) in kinds:
        for r in ((a, name, b), (b, name, a)):
            if r not in rels:
                rels.append(r)
    for r in rels:
        if dec == "bilinear":
            lim = np.sqrt(6.0 / (2 * d))
            params[O.rel_key(r)] = rng.uniform(-lim, lim, (d, d)).astype(np.float32)
        else:
            params[O.rel_key(r)] = rng.uniform(-6 / np.sqrt(d), 6 / np.sqrt(d), d).astype(np.float32)
    if not inter.endswith("simple"):
        lim = np.sqrt(6.0 / (2 * d))
        for m in sizes:
            params[O.pre_key(m)] = rng.uniform(-lim, lim, (d, d)).astype(np.float32)
            params[O.post_key(m)] = rng.uniform(-lim, lim, (d, d)).astype(np.float32)
    if bags:
        params[O.BAGS_KEY] = bags
    return params


# one formula per query type on a 3-mode toy schema a-b, a-c, b-c, a-a
TOY_SIZES = {"a": 90, "b": 70, "c": 50}
TOY_KINDS = (("a", "ab", "b"), ("a", "ac", "c"), ("b", "bc", "c"), ("a", "aa", "a"))
TOY_FORMULAS = {
    "1-chain": (("a", "ab", "b"),),
    "2-chain": (("a", "ab", "b"), ("b", "bc", "c")),
    "3-chain": (("c", "ac", "a"), ("a", "aa", "a"), ("a", "ab", "b")),
    "2-inter": (("a", "ab", "b"), ("a", "ac", "c")),
    "3-inter": (("a", "ab", "b"), ("a", "aa", "a"), ("a", "ac", "c")),
    "3-inter_chain": (("a", "ac", "c"), (("a", "ab", "b"), ("b", "bc", "c"))),
    "3-chain_inter": (("c", "ac", "a"), (("a", "ab", "b"), ("a", "aa", "a"))),
}


# ... and sizes whose tables (n + 2 rows: 91 / 71 / 53) leave a remainder when sharded over 2, 3 or 4 ranks
TOY_SIZES_ODD = {"a": 89, "b": 69, "c": 51}


def toy_batch(rng, qtype, B, hub=False, sizes=None):
    """Random row indices (1..n) for a TOY_FORMULAS batch: (target, neg, anchors[k,B])."""
    sizes = sizes or TOY_SIZES
    plan = O.make_plan(qtype, TOY_FORMULAS[qtype])
    nt = sizes[plan["target_mode"]]
    target = rng.randint(1, nt + 1, B).astype(np.int32)
    neg = rng.randint(1, nt + 1, B).astype(np.int32)
    anchors = np.stack([rng.randint(1, sizes[m] + 1, B) for m in plan["anchor_modes"]]).astype(np.int32)
    if hub:                       # many queries hit the same rows -> atomics collide
        target[: B // 2] = target[0]
        anchors[:, : B // 2] = anchors[:, :1]
        neg[B // 3:] = neg[0]
    return target, neg, anchors


# ---- run_train trajectories against what the reference recorded (tests/golden/train_*.npz, trainlong_*.npz) ----------------
def oracle_replay(p0, dec, inter, iterations, dtype=np.float32, lr=0.01, signal=None, sgd=False):
    """The oracle's own trajectory over recorded iterations: ``iterations`` = [[(query type, rels, target, neg, anchors, weight,
    margin), ...] per iteration]; every iteration is zero_grad -> weighted margin losses -> backward -> Adam on the touched
    tensors (train_helpers.py:50-79).  Returns (iteration losses, final params, per-tensor step counts).  In float32 this is the
    yardstick of the device tests: how far fp32 arithmetic alone lands from the reference's recorded numbers.  ``signal`` (a
    dict, filled in place): per tensor the mask of SIGNAL elements — those whose gradient in every step that touched the tensor
    was exactly 0 or above 1e-4 of the tensor's largest; everywhere else Adam's sign-like first steps (dp = lr g / (|g| + 1e-8))
    turn summation-order noise into lr-sized moves in the reference itself."""
    params = {k: (v if k == O.BAGS_KEY else np.array(v, dtype=dtype)) for k, v in p0.items()}      # (the bag registry travels as it is)
    state, losses = {}, []
    for batches in iterations:
        grads = O.zero_grads_like(params, dtype)
        touched, total = set(), 0.0
        for (qtype, rels, t, ng, a, w, m) in batches:
            plan = O.make_plan(qtype, rels)
            l, _, _, _ = O.margin_fwd_bwd(params, plan, dec, inter, t, ng, a, margin=m, weight=w, grads=grads, dtype=dtype)
            total += w * float(l)
            touched |= O.touched_keys(plan, dec, inter)
        losses.append(total)
        if signal is not None:
            for k in touched:
                g = np.abs(grads[k])
                signal[k] = signal.get(k, np.ones(g.shape, dtype=bool)) & ((g == 0) | (g > 1e-4 * g.max()))
        if sgd:
            O.sgd_step(params, grads, touched, lr=lr)                 # torch.optim.SGD(momentum=0), bio/train.py:59-60
        else:
            O.adam_step(params, grads, state, touched, lr=lr)
    return np.asarray(losses), params, {k: st["step"] for k, st in state.items()}


def fixture_iterations(z):
    """The recorded batches of a train_*.npz fixture in oracle_replay's form."""
    import json
    from golden_utils import to_rels
    out, i = [], 0
    while "it%d/n" % i in z.files:
        batches = []
        for j in range(int(z["it%d/n" % i])):
            meta = json.loads(str(z["it%d/b%d/meta" % (i, j)]))
            w = 1.0 if meta["type"] == "1-chain" else (0.005 if "inter" in meta["type"] else 0.01)
            batches.append((meta["type"], to_rels(meta["rels"]), z["it%d/b%d/target" % (i, j)], z["it%d/b%d/neg" % (i, j)],
                            z["it%d/b%d/anchors" % (i, j)], w, float(meta["margin"])))
        out.append(batches)
        i += 1
    return out


def ema_series(losses, resets=(), alpha=0.01):
    """update_loss (train_helpers.py:11-17) over a loss series; ``resets`` = iterations at which the average starts over (the
    phase switch, train_helpers.py:59)."""
    out, ema = [], None
    for i, l in enumerate(losses):
        if i in resets:
            ema = None
        ema = l if ema is None else (1 - alpha) * ema + alpha * l
        out.append(ema)
    return np.asarray(out)


def eval_quanta(test_queries):
    """Per query type: (AUC quantum, percentile quantum) of the reference's statistics on these evaluation sets — one
    (positive, negative) pair changing order moves the AUC over n queries with one negative each by 1 / n^2 (half of it on a
    tie), and one rank flip against one of a query's k negatives moves the mean percentile by 100 / (k n_full)."""
    out = {}
    for qt, by_f in test_queries["one_neg"].items():
        n = sum(len(v) for v in by_f.values())
        full = [q for v in test_queries["full_neg"][qt].values() for q in v]
        for hard, tag in ((False, qt), (True, "Hard-" + qt)):
            if hard and "inter" not in qt:
                continue
            ks = [len(q.hard_neg_samples if hard else q.neg_samples) for q in full]
            out[tag] = (1.0 / (n * n), 100.0 / (min(ks) * len(full)))
    return out


def compare_train_logs(mine, ref, loss_allow, quanta, resets, auc_flips, perc_flips, what=""):
    """Line for line, run_train's log against the reference's: the same lines in the same order; every ``ema_loss`` within the
    moving average of ``loss_allow`` (per-iteration loss allowances) + the 1e-6 the log prints to; every ``val AUC`` within
    ``auc_flips`` pair flips and every ``val perc`` within ``perc_flips`` rank flips of the type's quantum (eval_quanta);
    macro average and improvement within what those flips can move them by.  Returns the largest deviations seen, in units of
    their allowances."""
    from golden_utils import parse_train_log
    a, b = parse_train_log(mine), parse_train_log(ref)
    strip = lambda lines: [l.split(": ")[0] if (" val AUC: " in l or l.startswith(("Test macro", "Improvement"))) else l.split(";")[0] for l in lines]
    assert strip(mine) == strip(ref), (what, strip(mine), strip(ref))
    assert a["edge_conv"] == b["edge_conv"], what
    worst = {"ema": 0.0, "auc": 0.0, "perc": 0.0}
    allow_ema = ema_series(loss_allow, resets)
    for (i, x), (j, y) in zip(a["iters"], b["iters"]):
        assert i == j
        tol = allow_ema[i] + 1.5e-6
        assert abs(x - y) <= tol, (what, "ema_loss", i, x, y, tol)
        worst["ema"] = max(worst["ema"], abs(x - y) / tol)
    assert len(a["evals"]) == len(b["evals"]), what
    macro_tol = 0.0
    for ea, eb in zip(a["evals"], b["evals"]):
        assert ea["iteration"] == eb["iteration"] and list(ea["scores"]) == list(eb["scores"]), what
        tols = []
        for tag, (auc, perc) in ea["scores"].items():
            rauc, rperc = eb["scores"][tag]
            qa, qp = quanta[tag]
            assert abs(auc - rauc) <= auc_flips * qa + 1.5e-6, (what, tag, "AUC", ea["iteration"], auc, rauc, qa)
            assert abs(perc - rperc) <= perc_flips * qp + 1.5e-6, (what, tag, "perc", ea["iteration"], perc, rperc, qp)
            worst["auc"] = max(worst["auc"], abs(auc - rauc) / qa)
            worst["perc"] = max(worst["perc"], abs(perc - rperc) / qp)
            tols.append(auc_flips * qa)
        macro_tol = float(np.mean(tols))
    if b["macro"] is not None:
        assert abs(a["macro"] - b["macro"]) <= macro_tol + 1.5e-6, (what, "macro", a["macro"], b["macro"])
    if b["improvement"] is not None:
        assert abs(a["improvement"] - b["improvement"]) <= 4 * macro_tol / max(b["macro"], 1e-6) + 1.5e-6, (what, a["improvement"], b["improvement"])
    return worst
