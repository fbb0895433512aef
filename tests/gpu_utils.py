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
