"""Vectorised synthetic graphs and query pools for benchmarks and large parity runs.

The Bio data set of the reference is not redistributable offline (netquery README.md:15
points at a download), so throughput is measured on "bio-synth" (SURVEY.md §8d): same
5-mode schema, 97 000 nodes, 9 undirected relation kinds (14 directed), 60 000 uniform
random edges per kind.  Graph and queries are generated directly as numpy CSR / int32
row arrays (no Python objects): a query pool is a set of random walks with the shape of
each query type; regular negatives are uniform nodes of the target mode (what
model.py:118 does for 1-chain; for the other types a uniform node is a valid negative
with probability > 0.99 on this sparse graph), hard negatives are neighbours of the first
anchor (satisfy one branch).  Table row of local node i = i + 1 (row 0 = dummy).
"""
from __future__ import annotations

import numpy as np

from .data_utils import BIO_SYNTH_EDGES_PER_KIND, BIO_SYNTH_KINDS, BIO_SYNTH_SIZES
from .graph import Formula, _reverse_relation


def zipf_draw(rng, n, size, exponent, perm=None):
    """``size`` draws from {0 .. n-1} with P(rank k) ~ 1 / (k + 1)^exponent (inverse CDF); ``perm`` maps rank -> item
    (the hubs are then scattered over the id range, as in a real graph, instead of being the first ids)."""
    w = 1.0 / np.power(np.arange(1, n + 1, dtype=np.float64), exponent)
    cdf = np.cumsum(w)
    k = np.searchsorted(cdf, rng.random_sample(size) * cdf[-1], side="right")
    k = np.minimum(k, n - 1)
    return k if perm is None else perm[k]


class CsrGraph(object):
    """Directed relations as CSR over local node indices of the source mode."""

    def __init__(self, mode_sizes, kinds=BIO_SYNTH_KINDS, edges_per_kind=BIO_SYNTH_EDGES_PER_KIND, seed=0, zipf=None):
        """``zipf`` = exponent s: edge endpoints are drawn with P(node of degree rank k) ~ 1 / (k + 1)^s instead of
        uniformly — the heavy-tailed degree distribution of the reference's data (protein hubs in Bio, popular
        communities in Reddit: graph.py:108-122 keeps the real adjacency).  One rank order per mode, shared by all of the
        mode's relations: a hub is a hub in every relation it takes part in."""
        rng = np.random.RandomState(seed)
        self.mode_sizes = dict(mode_sizes)
        self.modes = sorted(mode_sizes)
        self.zipf = zipf
        hub_order = {m: np.random.RandomState(seed + 101 + i).permutation(mode_sizes[m]) for i, m in enumerate(self.modes)} if zipf else None
        # rows of each mode's embedding table: len(node_maps[mode]) + 1 with the extra -1 key (bio/data_utils.py:14-17)
        self.table_rows = {m: n + 2 for m, n in self.mode_sizes.items()}
        self.bags = {}           # mode -> (ptr int32[n+1], ids int32[nnz]) for nn.EmbeddingBag feature modes
        self.relations = {}
        self.csr = {}
        for (ma, name, mb) in kinds:
            self.relations.setdefault(ma, [])
            self.relations.setdefault(mb, [])
            if (mb, name) not in self.relations[ma]:
                self.relations[ma].append((mb, name))
            if (ma, name) not in self.relations[mb]:
                self.relations[mb].append((ma, name))
            if zipf:
                u = zipf_draw(rng, mode_sizes[ma], edges_per_kind, zipf, hub_order[ma])
                v = zipf_draw(rng, mode_sizes[mb], edges_per_kind, zipf, hub_order[mb])
            else:
                u = rng.randint(0, mode_sizes[ma], size=edges_per_kind)
                v = rng.randint(0, mode_sizes[mb], size=edges_per_kind)
            if ma == mb:
                keep = u != v
                u, v = u[keep], v[keep]
                u, v = np.concatenate([u, v]), np.concatenate([v, u])
                self.csr[(ma, name, mb)] = self._build(u, v, mode_sizes[ma])
            else:
                self.csr[(ma, name, mb)] = self._build(u, v, mode_sizes[ma])
                self.csr[(mb, name, ma)] = self._build(v, u, mode_sizes[mb])
        self.rels = sorted(self.csr.keys())

    @staticmethod
    def _build(src, dst, n):
        span = int(dst.max()) + 1 if len(dst) else 1        # distinct (src, dst) pairs in lexicographic order
        keys = np.unique(src.astype(np.int64) * span + dst.astype(np.int64))
        s, t = keys // span, keys % span
        indptr = np.zeros(n + 1, dtype=np.int64)
        indptr[1:] = np.cumsum(np.bincount(s, minlength=n))
        return indptr, t

    def out_relations(self, mode):
        return [(mode, name, to) for (to, name) in self.relations[mode]]

    def random_edges(self, rel, n, rng):
        indptr, indices = self.csr[rel]
        pos = rng.randint(0, len(indices), size=n)
        return np.searchsorted(indptr, pos, side="right") - 1, indices[pos]

    def random_neighbors(self, rel, nodes, rng):
        """One random neighbour per node (-1 where the node has none)."""
        indptr, indices = self.csr[rel]
        lo = indptr[nodes]
        deg = indptr[nodes + 1] - lo
        off = (rng.random_sample(len(nodes)) * np.maximum(deg, 1)).astype(np.int64)
        out = indices[np.minimum(lo + off, len(indices) - 1)]
        return np.where(deg > 0, out, -1)


class QueryPool(object):
    """n queries of one Formula as int32 TABLE ROWS: target[n], anchors[k,n], neg[n], hard[n]."""

    def __init__(self, formula, target, anchors, neg, hard):
        self.formula = formula
        self.n = len(target)
        self.target = (target + 1).astype(np.int32)
        self.anchors = (np.stack(anchors) + 1).astype(np.int32)
        self.neg = (neg + 1).astype(np.int32)
        self.hard = None if hard is None else (hard + 1).astype(np.int32)


def _walk(g, start_nodes, rels, rng):
    """Follow rels from start_nodes; returns (end nodes, ok mask)."""
    cur, ok = start_nodes, np.ones(len(start_nodes), dtype=bool)
    for r in rels:
        nxt = g.random_neighbors(r, np.where(ok, cur, 0), rng)
        ok &= nxt >= 0
        cur = nxt
    return cur, ok


def sample_pool(g, qtype, rels, n, rng):
    """Random query pool of one formula; ``rels`` nested as in netquery/graph.py:42-54."""
    formula = Formula(qtype, rels)
    tm = formula.target_mode
    out_t, out_a, got = [], [], 0
    first = rels[0]
    while got < n:
        m = max(2 * (n - got), 256)
        t, a0 = g.random_edges(first, m, rng)          # target -first-> a0 (or the chain's 1st variable)
        ok = np.ones(m, dtype=bool)
        if qtype in ("1-chain", "2-chain", "3-chain"):
            end, ok2 = _walk(g, a0, rels[1:], rng)
            anchors, ok = [end], ok & ok2
        elif qtype in ("2-inter", "3-inter"):
            anchors = [a0]
            for r in rels[1:]:
                a, ok2 = _walk(g, t, [r], rng)
                anchors.append(a)
                ok &= ok2
        elif qtype == "3-inter_chain":
            a1, ok2 = _walk(g, t, list(rels[1]), rng)
            anchors, ok = [a0, a1], ok & ok2
        else:                                          # 3-chain_inter: t -r0-> v ; v -r1-> a0' , v -r2-> a1'
            v = a0
            b0, ok0 = _walk(g, v, [rels[1][0]], rng)
            b1, ok1 = _walk(g, v, [rels[1][1]], rng)
            anchors, ok = [b0, b1], ok0 & ok1
        out_t.append(t[ok])
        out_a.append([a[ok] for a in anchors])
        got += int(ok.sum())
    target = np.concatenate(out_t)[:n]
    anchors = [np.concatenate([chunk[i] for chunk in out_a])[:n] for i in range(len(out_a[0]))]
    neg = rng.randint(0, g.mode_sizes[tm], size=n)
    hard = None
    if "inter" in qtype:
        if qtype == "3-chain_inter":
            # a node reached from a neighbour of anchor 0 (satisfies one branch of the intersection)
            v, ok = _walk(g, anchors[0], [_reverse_relation(rels[1][0])], rng)
            hard, ok2 = _walk(g, np.where(ok, v, 0), [_reverse_relation(rels[0])], rng)
            ok &= ok2
        else:
            hard, ok = _walk(g, anchors[0], [_reverse_relation(rels[0])], rng)
        hard = np.where(ok, hard, neg)
    return QueryPool(formula, target, anchors, neg, hard)


def enumerate_formulas(g, qtype, limit, rng):
    """Up to ``limit`` distinct relation combinations valid for the schema."""
    combos = []
    for _ in range(limit * 20):
        if len(combos) >= limit:
            break
        r1 = g.rels[rng.randint(len(g.rels))]
        pick = lambda mode: g.out_relations(mode)[rng.randint(len(g.out_relations(mode)))]
        if qtype == "1-chain":
            c = (r1,)
        elif qtype == "2-chain":
            c = (r1, pick(r1[2]))
        elif qtype == "3-chain":
            r2 = pick(r1[2])
            c = (r1, r2, pick(r2[2]))
        elif qtype == "2-inter":
            c = (r1, pick(r1[0]))
        elif qtype == "3-inter":
            c = (r1, pick(r1[0]), pick(r1[0]))
        elif qtype == "3-inter_chain":
            r2 = pick(r1[0])
            c = (r1, (r2, pick(r2[2])))
        else:
            c = (r1, (pick(r1[2]), pick(r1[2])))
        if c not in combos:
            combos.append(c)
    return combos


FULL_MIX = (("1-chain", 1.0, False), ("2-chain", 0.01, False), ("3-chain", 0.01, False),
            ("2-inter", 0.005, False), ("2-inter", 0.005, True),
            ("3-inter", 0.005, False), ("3-inter", 0.005, True),
            ("3-inter_chain", 0.005, False), ("3-inter_chain", 0.005, True))
"""BASELINE 'Bio full conjunctive mix' iteration (SURVEY.md §8d C3): the reference's
post-burn-in schedule (train_helpers.py:51,64-72) over {1/2/3-chain, 2/3-inter, 3-inter_chain}:
weights 1 / 0.01 / 0.005, intersections once with regular and once with hard negatives."""


def bio_synth(seed=0, sizes=None, edges_per_kind=None, zipf=None):
    return CsrGraph(sizes or BIO_SYNTH_SIZES, BIO_SYNTH_KINDS,
                    edges_per_kind or BIO_SYNTH_EDGES_PER_KIND, seed=seed, zipf=zipf)


REDDIT_SYNTH_KINDS = (("user", "up", "post"), ("user", "down", "post"), ("user", "make", "post"), ("user", "comment", "post"),
                      ("user", "subscribe", "community"), ("post", "belong", "community"))
"""The Reddit schema of the reference: 6 undirected kinds = the 12 directed relations of
reddit/data_utils_new.py:193-197 (user->post up/down/make/comment, user->community subscribe, post->community
belong, and their reverses)."""
REDDIT_SYNTH_SIZES = {"user": 500000, "post": 400000, "community": 2000}
REDDIT_SYNTH_EDGES_PER_KIND = 1000000
REDDIT_SYNTH_WORDS = 50000
REDDIT_SYNTH_BAG_LEN = (5, 30)


def reddit_synth(seed=0, sizes=None, edges_per_kind=None, n_words=None, bag_len=None, zipf=None):
    """BASELINE config 5 stand-in (the Reddit data of the reference is private): 3 modes, 12
    directed relations, user 500 k / post 400 k / community 2 k nodes, 1 M uniform random edges per kind.  Features as
    reddit/data_utils_new.py:153-169: user and community are ``nn.Embedding(N + 1, d)`` with row = node + 1; a post
    is an ``nn.EmbeddingBag(num_words, d)`` (mode mean) over its word set — here 5..30 uniform words of a 50 k
    vocabulary.  Bag i + 1 belongs to post i (bag 0 is a dummy, so that "row = node + 1" holds for every mode).
    ``zipf`` = exponent: node degrees AND word frequencies follow 1 / rank^zipf (a real vocabulary is Zipfian:
    reddit/data_utils_new.py:155,162-169 builds the EmbeddingBag over the posts' actual words)."""
    sizes = dict(sizes or REDDIT_SYNTH_SIZES)
    g = CsrGraph(sizes, REDDIT_SYNTH_KINDS, edges_per_kind or REDDIT_SYNTH_EDGES_PER_KIND, seed=seed, zipf=zipf)
    n_words = int(n_words or REDDIT_SYNTH_WORDS)
    lo, hi = bag_len or REDDIT_SYNTH_BAG_LEN
    rng = np.random.RandomState(seed + 7)
    lens = rng.randint(lo, hi + 1, size=sizes["post"] + 1)
    ptr = np.zeros(sizes["post"] + 2, dtype=np.int64)
    ptr[1:] = np.cumsum(lens)
    if zipf:
        ids = zipf_draw(rng, n_words, int(ptr[-1]), zipf, np.random.RandomState(seed + 11).permutation(n_words))
    else:
        ids = rng.randint(0, n_words, size=int(ptr[-1]))
    g.bags = {"post": (ptr.astype(np.int32), ids.astype(np.int32))}
    g.table_rows = {"user": sizes["user"] + 1, "community": sizes["community"] + 1, "post": n_words}
    return g


def make_pools(g, types, formulas_per_type, pool_size, seed=0):
    rng = np.random.RandomState(seed + 1)
    pools = {}
    for qt in types:
        pools[qt] = [sample_pool(g, qt, rels, pool_size, rng)
                     for rels in enumerate_formulas(g, qt, formulas_per_type, rng)]
    return pools


def mix_iteration(pools, mix, step, batch_size, rank=0, world=1):
    """The (formula, target, neg, anchors, weight, margin) items of one training iteration.
    Every rank draws the SAME formula per batch and its own slice (data parallel)."""
    items = []
    for j, (qt, w, hard) in enumerate(mix):
        plist = pools[qt]
        p = plist[(step * 7 + j) % len(plist)]
        start = ((step * world + rank) * batch_size) % max(p.n - batch_size + 1, 1)
        sl = slice(start, start + batch_size)
        neg = p.hard[sl] if hard else p.neg[sl]
        items.append((p.formula, p.target[sl], neg, p.anchors[:, sl], w / world, 1.0))
    return items
