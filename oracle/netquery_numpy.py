"""ORACLE (test infrastructure, not product code) — numpy restatement of the
reference's conjunctive-query hot path, forward AND hand-derived backward.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module; the product path (``graphqembed_amd``) never does.

Parity status: the reference ships no golden vectors / known-answer tests for
this path, and all of its arithmetic lives in PyTorch.  This
restatement is therefore pinned against outputs of the reference itself, run in
the build container by ``oracle/make_golden.py`` (reference imported through a
scripted lib2to3 transform) and committed as ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` checks every fixture.

What is restated (reference file:line):
  encode            netquery/encoders.py:40-43 + bio/data_utils.py:20-21
  project / chains  netquery/decoders.py:142-150 (bilinear), 200-208 (transe),
                    228-236 (bilinear-diag)
  intersections     netquery/decoders.py:288-300 (SetIntersection),
                    311-319 (SimpleSetIntersection)
  forward dispatch  netquery/model.py:70-109
  margin loss       netquery/model.py:122-126
  cosine            torch.nn.CosineSimilarity(dim=0, eps=1e-8) as torch 2.10
                    computes it: x.y / (max(|x|,eps) * max(|y|,eps))
  Adam / SGD        torch.optim.Adam / SGD defaults used at bio/train.py:59-62,
                    per-tensor step counters, untouched tensors skipped
                    (SURVEY.md Appendix B "granularity rule").

Layout: vectors are rows ``[B, d]`` here (the reference uses ``[d, B]`` columns;
the math is the same).  ``dtype`` selects float32 (mimic) or float64 (truth).
"""
from __future__ import annotations

import numpy as np

COS_EPS = 1e-8
CHAIN_TYPES = ("1-chain", "2-chain", "3-chain")
DECODERS = ("bilinear-diag", "transe", "bilinear")
INTER_DECODERS = ("min", "mean", "min-simple", "mean-simple")


# ----------------------------------------------------------------------------
# formula plan
# ----------------------------------------------------------------------------
def rev(rel):
    """netquery/graph.py:4-5"""
    return (rel[2], rel[1], rel[0])


def table_key(mode):
    return "enc.feat-%s.weight" % mode          # netquery/encoders.py:26


def rel_key(rel):
    return "path_dec." + "_".join(rel)          # netquery/decoders.py:226


def pre_key(mode):
    return "inter_dec.%s_premat" % mode         # netquery/decoders.py:283


def post_key(mode):
    return "inter_dec.%s_postmat" % mode        # netquery/decoders.py:286


def _as_rel(r):
    return tuple(r)


def make_plan(query_type, rels):
    """Flatten a formula into what the maths needs (netquery/model.py:70-109,
    netquery/graph.py:17-24).

    chain types:  ``chain`` = relations applied on the TARGET side, in order.
    inter types:  ``branches[i]`` = relations applied to anchor i (already
                  reversed, in application order); ``inter_mode``;
                  ``final`` = relations applied to the intersection output
                  (3-chain_inter only).
    """
    def norm(x):
        return tuple(norm(y) for y in x) if isinstance(x[0], (tuple, list)) else tuple(x)
    rels = norm(rels)
    plan = {"type": query_type, "target_mode": rels[0][0]}
    if query_type in CHAIN_TYPES:
        plan["chain"] = [_as_rel(r) for r in rels]
        plan["anchor_modes"] = [rels[-1][2]]
    elif query_type in ("2-inter", "3-inter"):
        plan["branches"] = [[rev(r)] for r in rels]
        plan["anchor_modes"] = [r[2] for r in rels]
        plan["inter_mode"] = rels[0][0]
        plan["final"] = []
    elif query_type == "3-inter_chain":
        plan["branches"] = [[rev(rels[0])], [rev(r) for r in rels[1][::-1]]]
        plan["anchor_modes"] = [rels[0][2], rels[1][-1][2]]
        plan["inter_mode"] = rels[0][0]
        plan["final"] = []
    elif query_type == "3-chain_inter":
        plan["branches"] = [[rev(rels[1][0])], [rev(rels[1][1])]]
        plan["anchor_modes"] = [rels[1][0][2], rels[1][1][2]]
        plan["inter_mode"] = rels[0][2]
        plan["final"] = [rev(rels[0])]
    else:
        raise ValueError("unknown query type %r" % (query_type,))
    return plan


# ----------------------------------------------------------------------------
# primitive ops, forward + backward
# ----------------------------------------------------------------------------
BAGS_KEY = "__bags__"   # params[BAGS_KEY] = {mode: (ptr[n+1], ids[nnz])}: modes whose feature is an EmbeddingBag


def _bag(params, mode):
    bags = params.get(BAGS_KEY)
    return None if bags is None else bags.get(mode)


def _encode(params, mode, rows, dt):
    """rows of the mode's table, L2-normalised, NO eps (encoders.py:41-43).  For a bag mode (Reddit
    posts: nn.EmbeddingBag, mode 'mean', reddit/data_utils_new.py:155,162-169) ``rows`` are bag indices
    and the raw vector is the mean of the bag's word rows."""
    bag = _bag(params, mode)
    W = params[table_key(mode)]
    if bag is None:
        raw = W[rows].astype(dt)
    else:
        ptr, ids = bag
        raw = np.stack([W[ids[ptr[r]:ptr[r + 1]]].astype(dt).mean(axis=0) for r in np.asarray(rows)])
    nrm = np.sqrt((raw * raw).sum(axis=1, keepdims=True))
    return raw / nrm, nrm


def _encode_bwd(grads, mode, rows, xhat, nrm, g):
    """d(x/|x|): (g - xhat (xhat.g)) / |x|, scatter-ADD into the dense table grad
    (duplicate rows accumulate, as a dense nn.Embedding backward does); a bag spreads its
    gradient / len over its word rows (EmbeddingBag mean backward)."""
    gx = (g - xhat * (xhat * g).sum(axis=1, keepdims=True)) / nrm
    bag = _bag(grads, mode)
    if bag is None:
        np.add.at(grads[table_key(mode)], rows, gx)
    else:
        ptr, ids = bag
        for b, r in enumerate(np.asarray(rows)):
            w = ids[ptr[r]:ptr[r + 1]]
            np.add.at(grads[table_key(mode)], w, np.broadcast_to(gx[b] / len(w), (len(w), gx.shape[1])))


def _project(dec, params, rel, v):
    """path_dec.project (decoders.py:149-150, 207-208, 235-236): anchor side."""
    w = params[rel_key(rel)].astype(v.dtype)
    if dec == "bilinear-diag":
        return v * w
    if dec == "transe":
        return v + w
    return v @ w.T                    # M . v  for every row


def _project_bwd(dec, params, grads, rel, v, g):
    w = params[rel_key(rel)].astype(v.dtype)
    if dec == "bilinear-diag":
        grads[rel_key(rel)] += (g * v).sum(axis=0)
        return g * w
    if dec == "transe":
        grads[rel_key(rel)] += g.sum(axis=0)
        return g
    grads[rel_key(rel)] += g.T @ v    # sum_b g_b v_b^T
    return g @ w                      # M^T g


def _cos(x, y):
    nx = np.maximum(np.sqrt((x * x).sum(axis=1)), COS_EPS)
    ny = np.maximum(np.sqrt((y * y).sum(axis=1)), COS_EPS)
    return (x * y).sum(axis=1) / (nx * ny), nx, ny


def _cos_bwd(x, y, s, nx, ny, gs):
    """grad of cos wrt x and y (valid where the norms exceed eps — always, on
    this path, unless a query vector collapses to 0)."""
    gs = gs[:, None]
    s = s[:, None]
    nx = nx[:, None]
    ny = ny[:, None]
    gx = gs * (y / (nx * ny) - s * x / (nx * nx))
    gy = gs * (x / (nx * ny) - s * y / (ny * ny))
    return gx, gy


def _intersect(inter, params, mode, es):
    """inter_dec(e1, e2, mode[, e3]) (decoders.py:288-300 / 311-319).
    Returns (q, cache)."""
    agg_min = inter.startswith("min")
    if inter.endswith("simple"):
        stack = np.stack(es)
        if agg_min:
            arg = stack.argmin(axis=0)         # first minimum, as torch.min does
            q = np.take_along_axis(stack, arg[None], axis=0)[0]
        else:
            arg, q = None, stack.mean(axis=0)
        return q, {"arg": arg}
    dt = es[0].dtype
    P = params[pre_key(mode)].astype(dt)
    Q = params[post_key(mode)].astype(dt)
    zs = [e @ P.T for e in es]
    hs = np.stack([np.maximum(z, 0) for z in zs])
    if agg_min:
        arg = hs.argmin(axis=0)
        h = np.take_along_axis(hs, arg[None], axis=0)[0]
    else:
        arg, h = None, hs.mean(axis=0)
    return h @ Q.T, {"arg": arg, "zs": zs, "h": h}


def _intersect_bwd(inter, params, grads, mode, es, cache, gq):
    n = len(es)
    agg_min = inter.startswith("min")
    if inter.endswith("simple"):
        if agg_min:
            return [gq * (cache["arg"] == i) for i in range(n)]
        return [gq / n for _ in range(n)]
    dt = es[0].dtype
    P = params[pre_key(mode)].astype(dt)
    Q = params[post_key(mode)].astype(dt)
    grads[post_key(mode)] += gq.T @ cache["h"]
    gh = gq @ Q
    out = []
    for i in range(n):
        ghi = gh * (cache["arg"] == i) if agg_min else gh / n
        gz = ghi * (cache["zs"][i] > 0)          # relu'(0) = 0
        grads[pre_key(mode)] += gz.T @ es[i]
        out.append(gz @ P)
    return out


# ----------------------------------------------------------------------------
# forward (scores only)
# ----------------------------------------------------------------------------
def forward_scores(params, plan, dec, inter, target_rows, anchor_rows, dtype=np.float64):
    """QueryEncoderDecoder.forward (model.py:70-109).  ``target_rows[B]`` and
    ``anchor_rows[k][B]`` are TABLE ROWS (node_maps[mode][node] + 1)."""
    t, _ = _encode(params, plan["target_mode"], target_rows, dtype)
    if plan["type"] in CHAIN_TYPES:
        a, _ = _encode(params, plan["anchor_modes"][0], anchor_rows[0], dtype)
        return _chain_score(dec, params, plan["chain"], t, a)[0]
    q, _ = _query_vector(params, plan, dec, inter, anchor_rows, dtype)
    return _cos(t, q)[0]


def _chain_score(dec, params, chain, t, a):
    """path_dec.forward(target, anchor, rels) (decoders.py:142-147, 200-205, 228-233)."""
    dt = t.dtype
    if dec == "bilinear-diag":
        act = t
        for r in chain:
            act = act * params[rel_key(r)].astype(dt)
        return (act * a).sum(axis=1), {"u": act}
    if dec == "transe":
        u = t
        for r in chain:
            u = u + params[rel_key(r)].astype(dt)
        s, na, nu = _cos(a, u)
        return s, {"u": u, "nu": nu, "na": na}
    acts = [t]
    for r in chain:
        acts.append(acts[-1] @ params[rel_key(r)].astype(dt))
    s, nu, na = _cos(acts[-1], a)
    return s, {"acts": acts, "u": acts[-1], "nu": nu, "na": na}


def _query_vector(params, plan, dec, inter, anchor_rows, dtype):
    cache = {"enc": [], "steps": []}
    es = []
    for i, branch in enumerate(plan["branches"]):
        x, nrm = _encode(params, plan["anchor_modes"][i], anchor_rows[i], dtype)
        cache["enc"].append((x, nrm))
        steps = []
        v = x
        for r in branch:
            steps.append((r, v))
            v = _project(dec, params, r, v)
        cache["steps"].append(steps)
        es.append(v)
    q, icache = _intersect(inter, params, plan["inter_mode"], es)
    cache["es"], cache["inter"] = es, icache
    fsteps = []
    for r in plan["final"]:
        fsteps.append((r, q))
        q = _project(dec, params, r, q)
    cache["final_steps"] = fsteps
    return q, cache


# ----------------------------------------------------------------------------
# fused margin loss forward + backward
# ----------------------------------------------------------------------------
def zero_grads_like(params, dtype=np.float64):
    out = {k: np.zeros(v.shape, dtype=dtype) for k, v in params.items() if k != BAGS_KEY}
    if BAGS_KEY in params:
        out[BAGS_KEY] = params[BAGS_KEY]      # the scatter needs the bag structure too
    return out


def margin_fwd_bwd(params, plan, dec, inter, target_rows, neg_rows, anchor_rows,
                   margin=1.0, weight=1.0, grads=None, dtype=np.float64):
    """loss = mean_b max(0, margin - (s+_b - s-_b))  (model.py:122-126) and the
    gradient of ``weight * loss`` w.r.t. every parameter, ACCUMULATED into
    ``grads`` (dict keyed like ``params``; created if None).

    The query side (anchors, projections, intersection) is computed once and
    shared by the positive and the negative score — the reference recomputes
    it (model.py:122-123); gradients are identical.
    Returns (loss, pos_scores, neg_scores, grads)."""
    if grads is None:
        grads = zero_grads_like(params, dtype)
    target_rows = np.asarray(target_rows)
    neg_rows = np.asarray(neg_rows)
    B = len(target_rows)
    tm = plan["target_mode"]
    tp, ntp = _encode(params, tm, target_rows, dtype)
    tn, ntn = _encode(params, tm, neg_rows, dtype)

    if plan["type"] in CHAIN_TYPES:
        am = plan["anchor_modes"][0]
        a, na = _encode(params, am, anchor_rows[0], dtype)
        sp, cp = _chain_score(dec, params, plan["chain"], tp, a)
        sn, cn = _chain_score(dec, params, plan["chain"], tn, a)
    else:
        q, qc = _query_vector(params, plan, dec, inter, anchor_rows, dtype)
        sp, nxp, nyp = _cos(tp, q)
        sn, nxn, nyn = _cos(tn, q)

    hinge = margin - (sp - sn)
    loss = np.maximum(hinge, 0).mean()
    active = (hinge > 0).astype(dtype)          # clamp(min=0) has zero grad at 0
    gsp = -weight * active / B
    gsn = weight * active / B

    if plan["type"] in CHAIN_TYPES:
        ga = np.zeros_like(a)
        for (t, nt, rows, s, c, gs) in ((tp, ntp, target_rows, sp, cp, gsp),
                                         (tn, ntn, neg_rows, sn, cn, gsn)):
            gt, ga_part = _chain_bwd(dec, params, grads, plan["chain"], t, a, s, c, gs)
            ga += ga_part
            _encode_bwd(grads, tm, rows, t, nt, gt)
        _encode_bwd(grads, am, np.asarray(anchor_rows[0]), a, na, ga)
        return loss, sp, sn, grads

    gtp, gq1 = _cos_bwd(tp, q, sp, nxp, nyp, gsp)
    gtn, gq2 = _cos_bwd(tn, q, sn, nxn, nyn, gsn)
    _encode_bwd(grads, tm, target_rows, tp, ntp, gtp)
    _encode_bwd(grads, tm, neg_rows, tn, ntn, gtn)
    gq = gq1 + gq2
    for (r, vin) in qc["final_steps"][::-1]:
        gq = _project_bwd(dec, params, grads, r, vin, gq)
    ges = _intersect_bwd(inter, params, grads, plan["inter_mode"], qc["es"], qc["inter"], gq)
    for i, steps in enumerate(qc["steps"]):
        g = ges[i]
        for (r, vin) in steps[::-1]:
            g = _project_bwd(dec, params, grads, r, vin, g)
        x, nrm = qc["enc"][i]
        _encode_bwd(grads, plan["anchor_modes"][i], np.asarray(anchor_rows[i]), x, nrm, g)
    return loss, sp, sn, grads


def _chain_bwd(dec, params, grads, chain, t, a, s, c, gs):
    """backward of path_dec.forward; returns (grad wrt normalised target,
    grad wrt normalised anchor)."""
    dt = t.dtype
    if dec == "bilinear-diag":
        # s = sum_j t_j (prod_i w_i,j) a_j
        ws = [params[rel_key(r)].astype(dt) for r in chain]
        prod = np.ones_like(ws[0])
        for w in ws:
            prod = prod * w
        g = gs[:, None]
        for i, r in enumerate(chain):
            others = np.ones_like(prod)
            for j, w in enumerate(ws):
                if j != i:
                    others = others * w
            grads[rel_key(r)] += (g * t * a * others).sum(axis=0)
        return g * prod * a, g * prod * t
    if dec == "transe":
        ga, gu = _cos_bwd(a, c["u"], s, c["na"], c["nu"], gs)
        for r in chain:
            grads[rel_key(r)] += gu.sum(axis=0)
        return gu, ga
    gu, ga = _cos_bwd(c["u"], a, s, c["nu"], c["na"], gs)
    g = gu
    for i in range(len(chain) - 1, -1, -1):
        M = params[rel_key(chain[i])].astype(dt)
        grads[rel_key(chain[i])] += c["acts"][i].T @ g     # act_{i+1} = act_i M
        g = g @ M.T
    return g, ga


# ----------------------------------------------------------------------------
# optimisers (torch defaults)
# ----------------------------------------------------------------------------
def adam_step(params, grads, state, touched, lr=0.01, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam single-tensor update, applied ONLY to tensors in
    ``touched`` (grad is not None), each with its own step counter.
    ``state[key] = {"step": int, "m": array, "v": array}``; in place."""
    for k in touched:
        st = state.setdefault(k, {"step": 0, "m": np.zeros_like(params[k]), "v": np.zeros_like(params[k])})
        st["step"] += 1
        g = grads[k].astype(params[k].dtype)
        st["m"] += (1.0 - b1) * (g - st["m"])                  # lerp_
        st["v"] *= b2
        st["v"] += (1.0 - b2) * g * g
        bc1 = 1.0 - b1 ** st["step"]
        bc2 = 1.0 - b2 ** st["step"]
        denom = np.sqrt(st["v"]) / (bc2 ** 0.5) + eps
        params[k] -= (lr / bc1) * (st["m"] / denom)


def sgd_step(params, grads, touched, lr=0.01):
    """torch.optim.SGD(momentum=0) (bio/train.py:60)."""
    for k in touched:
        params[k] -= lr * grads[k].astype(params[k].dtype)


def touched_keys(plan, dec, inter):
    """Parameter tensors a batch of this formula gives a (dense) gradient to."""
    keys = {table_key(plan["target_mode"])}
    for m in plan["anchor_modes"]:
        keys.add(table_key(m))
    if plan["type"] in CHAIN_TYPES:
        keys.update(rel_key(r) for r in plan["chain"])
    else:
        for br in plan["branches"]:
            keys.update(rel_key(r) for r in br)
        keys.update(rel_key(r) for r in plan["final"])
        if not inter.endswith("simple"):
            keys.add(pre_key(plan["inter_mode"]))
            keys.add(post_key(plan["inter_mode"]))
    return keys
