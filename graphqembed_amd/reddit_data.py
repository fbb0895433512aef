"""The Reddit entry point: ``load_graph`` of netquery/reddit/data_utils_new.py:143-182 on this package.

What the reference's loader does, and what stays:
  * three pickles in ``info_dir``: ``adj_lists.pkl`` ({(mode, relation, mode): {node id: set(node ids)}}), ``rels.pkl``
    ({mode: [(to mode, relation)]}, reddit/data_utils_new.py:193-197) and ``post_words.pkl`` ({post id: collection of word ids});
  * node ids are PER MODE (user 3 and post 3 are different nodes);
  * users and communities get an ``nn.Embedding(count + 1, d)`` each, indexed by id + 1 (row 0 is never read);  ``count`` is the
    number of distinct ids that appear as a KEY of an adjacency list whose source mode is that mode (lines 148-149);
  * posts get ``nn.EmbeddingBag(num_words, d)`` (mode "mean", torch's default) over each post's word ids; ``num_words`` is the
    number of distinct word ids (line 150);
  * every table is initialised N(0, 1/d) (lines 158-159);
  * returns ``(graph, feature_modules)``.

What changes: the reference hides the id -> row convention and the word lists inside the ``_feature_func`` closure it hands to
``Graph``; here the fused kernel does the gather (encoders.DirectEncoder: table row = id + 1, a post's row = its bag index), so
the same information travels as data on the returned graph — ``graph.node_maps`` ({mode: {id: id}} for the table modes) and
``graph.bags`` ({"post": {post id: word ids}}) — where ``utils.get_encoder(0, graph, out_dims, feature_modules, cuda)`` picks it
up: the reference's call sequence (reddit/new_train.py:29-59) runs unchanged (examples/train_reddit.py).

An id the reference's tables cannot hold raises here as it does there (there: an index error inside nn.Embedding at the first
batch that names it; here: a KeyError from the encoder's row lookup, or a ValueError at load time for word ids)."""
from __future__ import annotations

import os
import pickle

import numpy as np
import torch
import torch.nn as nn

from .graph import Graph

TABLE_MODES = ("user", "community")
BAG_MODE = "post"


def _load_pickle(path):
    with open(path, "rb") as f:
        return pickle.load(f, encoding="latin1")      # (the data set's files are Python-2 pickles)


def read_info(info_dir):
    """(adj_lists, relations, post_words) as the three pickles hold them."""
    adj_lists = _load_pickle(os.path.join(info_dir, "adj_lists.pkl"))
    relations = _load_pickle(os.path.join(info_dir, "rels.pkl"))
    post_words = _load_pickle(os.path.join(info_dir, "post_words.pkl"))
    return adj_lists, relations, post_words


def count_sources(adj_lists, mode):
    """reddit/data_utils_new.py:148-149: distinct ids that are a key of an adjacency list leaving ``mode``."""
    return len(set(node for rel, adj in adj_lists.items() if rel[0] == mode for node in adj))


def build(adj_lists, relations, post_words, embed_dim=16):
    """``load_graph`` behind the file reads: (graph, feature_modules) from the three objects."""
    counts = {mode: count_sources(adj_lists, mode) for mode in TABLE_MODES}
    words = set(int(w) for ws in post_words.values() for w in ws)
    num_words = len(words)
    if words and (min(words) < 0 or max(words) >= num_words):
        raise ValueError("post_words.pkl: word ids must be 0..%d (the EmbeddingBag has one row per distinct word, "
                         "reddit/data_utils_new.py:150,154); found ids in [%d, %d]" % (num_words - 1, min(words), max(words)))
    bags = {post: [int(w) for w in ws] for post, ws in post_words.items()}
    empty = [p for p, ws in bags.items() if not ws]
    if empty:
        raise ValueError("post_words.pkl: %d posts without words (e.g. %r): their feature would be the mean of nothing" % (len(empty), empty[0]))
    feature_modules = {
        BAG_MODE: nn.EmbeddingBag(num_words, embed_dim),
        "user": nn.Embedding(counts["user"] + 1, embed_dim),
        "community": nn.Embedding(counts["community"] + 1, embed_dim),
    }
    for module in feature_modules.values():
        module.weight.data.normal_(0, 1. / embed_dim)
    feature_dims = {mode: module.weight.size()[1] for mode, module in feature_modules.items()}
    graph = Graph(None, feature_dims, relations, adj_lists)
    # what the reference's _feature_func closure knows, as data for the encoder (module docstring)
    graph.node_maps = {mode: dict([(i, i) for i in range(counts[mode])] + [(-1, -1)]) for mode in TABLE_MODES}
    graph.bags = {BAG_MODE: bags}
    return graph, feature_modules


def load_graph(info_dir, embed_dim=16, cuda=False):
    """reddit/data_utils_new.py:143-182.  ``cuda`` is accepted and ignored: the hot path runs on the GPU either way."""
    return build(*read_info(info_dir), embed_dim=embed_dim)


# ---- converted (flat) files: tools/convert_data.py --reddit ---------------------------------------------------------------------
def flat_node_maps(adj_lists, post_words):
    """{mode: {id: index}} for flatdata.FlatGraph.from_reference: identity for the table modes over every id the files name (an id
    beyond the reference's table is still caught by the encoder when a batch names it), the bag order for posts."""
    ids = {m: set() for m in TABLE_MODES + (BAG_MODE,)}
    for (a, _, b), adj in adj_lists.items():
        for u, neigh in adj.items():
            ids[a].add(u)
            ids[b].update(neigh)
    maps = {m: {i: i for i in range(max(ids[m]) + 1 if ids[m] else 0)} for m in TABLE_MODES}
    order = list(post_words.keys()) + sorted(ids[BAG_MODE] - set(post_words.keys()))
    maps[BAG_MODE] = {p: i for i, p in enumerate(order)}
    for m in maps:
        maps[m][-1] = -1
    return maps


def save_post_words(path, post_words, node_map):
    """post_words.npz: CSR of word ids in the order of the flat graph's post index."""
    order = sorted((i, p) for p, i in node_map.items() if p in post_words)
    ptr = np.zeros(len(order) + 1, dtype=np.int64)
    ptr[1:] = np.cumsum([len(post_words[p]) for _, p in order])
    ids = np.concatenate([np.asarray(list(post_words[p]), dtype=np.int32) for _, p in order]) if order else np.zeros(0, np.int32)
    np.savez(path, ptr=ptr, ids=ids, posts=np.asarray([p for _, p in order], dtype=np.int64))


def load_flat_graph(flat_dir, embed_dim=16):
    """The converted directory (graph.npz + post_words.npz) -> (graph, feature_modules, flat_graph): the same objects ``load_graph``
    returns, plus the FlatGraph the converted query files are read against (flatdata.load_queries_by_formula(path, flat_graph))."""
    from . import flatdata
    flat = flatdata.FlatGraph.load(os.path.join(flat_dir, "graph.npz"))
    relations, adj_lists, _ = flat.to_reference()
    z = np.load(os.path.join(flat_dir, "post_words.npz"))
    ptr, ids, posts = z["ptr"], z["ids"], z["posts"]
    post_words = {int(p): ids[ptr[i]:ptr[i + 1]].tolist() for i, p in enumerate(posts)}
    graph, feature_modules = build(adj_lists, relations, post_words, embed_dim)
    # the flat files index posts by the flat graph's order: the encoder's flat-row translation needs that map
    graph.node_maps[BAG_MODE] = dict([(int(n), i) for i, n in enumerate(flat.node_ids[flat.modes.index(BAG_MODE)]) if n >= 0] + [(-1, -1)])
    return graph, feature_modules, flat


# ---- a small data set in the reference's layout (no Reddit data ships with the repository: the original is private) ------------
RELATIONS = {   # reddit/data_utils_new.py:193-197
    "user": [("post", "up"), ("post", "down"), ("post", "make"), ("post", "comment"), ("community", "subscribe")],
    "post": [("user", "up"), ("user", "down"), ("user", "make"), ("user", "comment"), ("community", "belong")],
    "community": [("post", "belong"), ("user", "subscribe")],
}
KINDS = (("user", "up", "post"), ("user", "down", "post"), ("user", "make", "post"), ("user", "comment", "post"),
         ("user", "subscribe", "community"), ("post", "belong", "community"))      # the six undirected kinds behind RELATIONS
QUERY_TYPES = {2: ["2-chain", "2-inter"], 3: ["3-chain", "3-inter", "3-inter_chain", "3-chain_inter"]}


def write_synthetic_dataset(out_dir, n_user=300, n_post=400, n_comm=30, n_words=200, edges_per_kind=2500, bag_len=(3, 12),
                            train_queries=(800, 1200), test_per_type=20, held_out=0.1, full_negs=20, seed=0):
    """Write a random Reddit-shaped data set with the files reddit/new_train.py:29-46 reads: adj_lists.pkl (the TRAINING graph: a
    tenth of the edges is held out), rels.pkl, post_words.pkl, train_edges.pkl, {val,test}_edges-split.pkl, train_queries_{2,3}.pkl,
    {val,test}_queries_{2,3}-clean.pkl — protocol-2 pickles of the objects the reference pickles (serialize() tuples, defaultdict(set)
    adjacency).  Held-out files mix queries with one stored negative and with up to ``full_negs`` (load_test_queries_by_formula
    splits them).  Returns a summary string."""
    import copy
    import random
    from collections import defaultdict
    from .graph import Query
    os.makedirs(out_dir, exist_ok=True)
    rng = np.random.RandomState(seed)
    random.seed(seed)
    sizes = {"user": n_user, "post": n_post, "community": n_comm}
    adj = {(m1, name, m2): defaultdict(set) for m1, lst in RELATIONS.items() for (m2, name) in lst}

    def link(m1, name, m2, u, v):
        adj[(m1, name, m2)][u].add(v)
        adj[(m2, name, m1)][v].add(u)
    for (m1, name, m2) in KINDS:
        n_edges = edges_per_kind if m2 != "community" else max(edges_per_kind // 3, sizes[m1])
        for u, v in zip(rng.randint(0, sizes[m1], n_edges).tolist(), rng.randint(0, sizes[m2], n_edges).tolist()):
            link(m1, name, m2, u, v)
    # every id owns at least one outgoing edge, so that the reference's counts (distinct adjacency keys) are the id ranges
    for u in range(n_user):
        link("user", "make", "post", u, int(rng.randint(0, n_post)))
    for p in range(n_post):
        link("post", "belong", "community", p, int(rng.randint(0, n_comm)))
    for c in range(n_comm):
        link("community", "subscribe", "user", c, int(rng.randint(0, n_user)))
    post_words = {p: set(rng.randint(0, n_words, size=int(rng.randint(bag_len[0], bag_len[1] + 1))).tolist()) for p in range(n_post)}
    used = sorted(set(w for ws in post_words.values() for w in ws))
    remap = {w: i for i, w in enumerate(used)}                 # (word ids dense: 0 .. number of distinct words - 1, as clean_words leaves them)
    post_words = {p: set(remap[w] for w in ws) for p, ws in post_words.items()}
    dims = {m: 1 for m in RELATIONS}
    full = Graph(None, dims, RELATIONS, adj)
    edges = full.get_all_edges(seed=seed)
    seen, held = set(), []
    for (u, rel, v) in edges:                                  # one direction of each held-out edge
        key = (min((rel[0], u), (rel[2], v)), max((rel[0], u), (rel[2], v)), rel[1])
        if key in seen:
            continue
        seen.add(key)
        if len(held) < held_out * len(edges) / 2 and len(adj[rel][u]) > 2 and len(adj[(rel[2], rel[1], rel[0])][v]) > 2:
            held.append((u, rel, v))
    train_adj = copy.deepcopy(adj)
    train = Graph(None, dims, RELATIONS, train_adj)
    train.remove_edges(held)

    def dump(name, obj):
        with open(os.path.join(out_dir, name), "wb") as f:
            pickle.dump(obj, f, protocol=2)
    dump("adj_lists.pkl", train.adj_lists)
    dump("rels.pkl", RELATIONS)
    dump("post_words.pkl", post_words)
    train_edges = train.get_all_edges(seed=seed + 1)
    dump("train_edges.pkl", [Query(("1-chain", e), None, None, keep_graph=True).serialize() for e in train_edges])
    half = len(held) // 2
    for name, part in (("val", held[:half]), ("test", held[half:])):
        infos = []
        for k, e in enumerate(part):
            n_neg = 1 if k % 2 == 0 else full_negs
            infos.append(Query(("1-chain", e), full.get_negative_edge_samples(e, n_neg), None, n_neg + 1, keep_graph=True).serialize())
        dump("%s_edges-split.pkl" % name, infos)
    counts = {"train_edges": len(train_edges), "held_out_edges": len(held)}
    for arity, n in zip((2, 3), train_queries):
        qs = train.sample_queries(arity, n, 1)
        dump("train_queries_%d.pkl" % arity, [q.serialize() for q in qs])
        counts["train_queries_%d" % arity] = len(qs)
        for name in ("val", "test"):
            infos = []
            for neg_max in (1, full_negs):
                infos += [q.serialize() for q in full.sample_test_queries(train, QUERY_TYPES[arity], test_per_type, neg_max)]
            dump("%s_queries_%d-clean.pkl" % (name, arity), infos)
    return "%s: %d users, %d posts (%d words), %d communities; %s" % (out_dir, n_user, n_post, len(used), n_comm, counts)
