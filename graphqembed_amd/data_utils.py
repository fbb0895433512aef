"""Data loading / synthetic data for the conjunctive-query path.

Reference behaviour followed:
  * pickle layout ``graph_data.pkl = (rels, adj_lists, node_maps)``
    netquery/bio/data_utils.py:11-23 (``load_graph``)
  * query files = list of ``(query_graph, neg_samples, hard_neg_samples)``
    grouped ``{type: {Formula: [Query]}}``  netquery/data_utils.py:10-35
  * the "bio-synth" stand-in for the (absent, un-downloadable) Bio data set is
    SURVEY.md §8d's synthetic configuration: 5 modes / 97 000 nodes /
    9 undirected relation kinds (14 directed) / 60 000 edges per kind.
"""
from __future__ import annotations

import pickle
from collections import defaultdict

import numpy as np

from .graph import Graph, Query

# (mode_a, relation name, mode_b) undirected kinds of the bio-synth graph.
BIO_SYNTH_KINDS = (
    ("drug", "ddi", "drug"),
    ("drug", "targets", "protein"),
    ("drug", "causes", "sideeffect"),
    ("drug", "treats", "disease"),
    ("protein", "ppi", "protein"),
    ("protein", "has_function", "function"),
    ("protein", "assoc", "disease"),
    ("function", "is_a", "function"),
    ("disease", "is_a", "disease"),
)
BIO_SYNTH_SIZES = {"drug": 10000, "disease": 10000, "protein": 40000,
                   "sideeffect": 10000, "function": 27000}
BIO_SYNTH_EDGES_PER_KIND = 60000

# A miniature with the same schema, used by the golden fixtures and CPU tests.
BIO_TINY_SIZES = {"drug": 60, "disease": 50, "protein": 120, "sideeffect": 40, "function": 70}
BIO_TINY_EDGES_PER_KIND = 500


def make_synthetic_graph(mode_sizes, kinds=BIO_SYNTH_KINDS, edges_per_kind=BIO_SYNTH_EDGES_PER_KIND, seed=0):
    """Uniform-random heterogeneous graph in the reference's in-memory format.

    Returns ``(relations, adj_lists, node_ids)``:
      relations  {mode: [(to_mode, rel_name), ...]}
      adj_lists  {(mode, rel_name, to_mode): defaultdict(set)}   (both directions)
      node_ids   {mode: [global int ids]}  (ids are unique across modes)
    """
    rng = np.random.RandomState(seed)
    node_ids, base = {}, 0
    for mode in sorted(mode_sizes):
        node_ids[mode] = list(range(base, base + mode_sizes[mode]))
        base += mode_sizes[mode]
    relations = defaultdict(list)
    adj_lists = {}
    for (ma, name, mb) in kinds:
        fwd, rev = (ma, name, mb), (mb, name, ma)
        if (mb, name) not in relations[ma]:
            relations[ma].append((mb, name))
        if (ma, name) not in relations[mb]:
            relations[mb].append((ma, name))
        adj_lists.setdefault(fwd, defaultdict(set))
        adj_lists.setdefault(rev, defaultdict(set))
        us = rng.randint(0, mode_sizes[ma], size=edges_per_kind)
        vs = rng.randint(0, mode_sizes[mb], size=edges_per_kind)
        a0, b0 = node_ids[ma][0], node_ids[mb][0]
        for u, v in zip(us.tolist(), vs.tolist()):
            u, v = a0 + u, b0 + v
            if u == v:
                continue
            adj_lists[fwd][u].add(v)
            adj_lists[rev][v].add(u)
    return dict(relations), adj_lists, node_ids


REDDIT_RELATIONS = {   # netquery/reddit/data_utils_new.py:193-197
    "user": [("post", "up"), ("post", "down"), ("post", "make"), ("post", "comment"), ("community", "subscribe")],
    "post": [("user", "up"), ("user", "down"), ("user", "make"), ("user", "comment"), ("community", "belong")],
    "community": [("post", "belong"), ("user", "subscribe")],
}


def make_reddit_tiny(n_user=80, n_post=120, n_comm=12, n_words=60, seed=0):
    """The tiny Reddit-shaped world of the golden fixtures (oracle/make_golden.py::RedditWorld and the tests that rebuild it): node
    ids PER MODE, 500 random edges per user-post kind, 250 per kind that involves communities, 3..12 word ids per post.
    Returns (relations, adj_lists, post_words) with ``post_words[p]`` an int64 array.  The draw order is part of the fixtures."""
    rng = np.random.RandomState(seed)
    sizes = {"user": n_user, "post": n_post, "community": n_comm}
    adj_lists = {}
    for m1, lst in REDDIT_RELATIONS.items():
        for (m2, name) in lst:
            adj_lists.setdefault((m1, name, m2), defaultdict(set))
    for (m1, name, m2) in list(adj_lists.keys()):
        if m1 > m2:
            continue                                   # fill each undirected kind once
        n_edges = 500 if "community" not in (m1, m2) else 250
        for u, v in zip(rng.randint(0, sizes[m1], n_edges).tolist(), rng.randint(0, sizes[m2], n_edges).tolist()):
            adj_lists[(m1, name, m2)][u].add(v)
            adj_lists[(m2, name, m1)][v].add(u)
    post_words = {p: rng.randint(0, n_words, size=rng.randint(3, 13)).astype(np.int64) for p in range(n_post)}
    return REDDIT_RELATIONS, adj_lists, post_words


def make_node_maps(node_ids):
    """``{mode: {node_id: row}}`` with the reference's extra ``-1 -> -1`` entry
    (netquery/bio/data_utils.py:13-15); table row of a node = map value + 1."""
    maps = {m: {n: i for i, n in enumerate(ids)} for m, ids in node_ids.items()}
    for m in maps:
        maps[m][-1] = -1
    return maps


def load_graph_data(data_dir):
    """Read ``graph_data.pkl`` (a Python-2 pickle in the original data set)."""
    with open(data_dir + "/graph_data.pkl", "rb") as f:
        rels, adj_lists, node_maps = pickle.load(f, encoding="latin1")
    return rels, adj_lists, node_maps


def load_queries(data_file, keep_graph=False):
    with open(data_file, "rb") as f:
        raw = pickle.load(f, encoding="latin1")
    return [Query.deserialize(info, keep_graph=keep_graph) for info in raw]


def group_by_formula(queries):
    out = defaultdict(lambda: defaultdict(list))
    for q in queries:
        out[q.formula.query_type][q.formula].append(q)
    return out


def load_queries_by_formula(data_file):
    return group_by_formula(load_queries(data_file))


def load_queries_by_type(data_file, keep_graph=True):
    out = defaultdict(list)
    for q in load_queries(data_file, keep_graph=keep_graph):
        out[q.formula.query_type].append(q)
    return out


def split_test_queries(raw_infos):
    """``{"full_neg"|"one_neg": {type: {Formula: [Query]}}}``; a query with more
    than one stored negative goes to ``full_neg`` (netquery/data_utils.py:27-35)."""
    out = {"full_neg": defaultdict(lambda: defaultdict(list)),
           "one_neg": defaultdict(lambda: defaultdict(list))}
    for raw in raw_infos:
        key = "full_neg" if len(raw[1]) > 1 else "one_neg"
        q = Query.deserialize(raw)
        out[key][q.formula.query_type][q.formula].append(q)
    return out


def load_test_queries_by_formula(data_file):
    with open(data_file, "rb") as f:
        raw = pickle.load(f, encoding="latin1")
    return split_test_queries(raw)
