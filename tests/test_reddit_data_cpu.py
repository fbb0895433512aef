"""CPU: the Reddit entry point (graphqembed_amd/reddit_data.py = reddit/data_utils_new.py:143-182) on a synthetic data set written in
the reference's file layout, and its converter (tools/convert_data.py --reddit)."""
import os
import pickle
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def data_dir(tmp_path_factory):
    from graphqembed_amd import reddit_data
    d = str(tmp_path_factory.mktemp("reddit"))
    reddit_data.write_synthetic_dataset(d, n_user=90, n_post=120, n_comm=12, n_words=80, edges_per_kind=700, train_queries=(150, 200),
                                        test_per_type=6, seed=3)
    return d


def test_files_are_the_reference_layout(data_dir):
    """What reddit/new_train.py:29-46 opens, as protocol-2 pickles of the reference's objects."""
    names = ["adj_lists", "rels", "post_words", "train_edges", "val_edges-split", "test_edges-split"]
    names += ["train_queries_%d" % i for i in (2, 3)] + ["%s_queries_%d-clean" % (s, i) for s in ("val", "test") for i in (2, 3)]
    for n in names:
        assert os.path.exists(os.path.join(data_dir, n + ".pkl")), n
    with open(os.path.join(data_dir, "adj_lists.pkl"), "rb") as f:
        adj = pickle.load(f, encoding="latin1")
    assert len(adj) == 12 and all(len(r) == 3 for r in adj)                     # 12 directed relations, data_utils_new.py:193-197
    for (a, name, b), lists in adj.items():                                      # both directions populated (graph.py:194,243,251)
        back = adj[(b, name, a)]
        assert all(u in back[v] for u, neigh in lists.items() for v in neigh)
    with open(os.path.join(data_dir, "train_queries_3.pkl"), "rb") as f:
        raw = pickle.load(f, encoding="latin1")
    assert isinstance(raw, list) and len(raw[0]) == 3 and raw[0][0][0].startswith("3-")


def test_load_graph_builds_what_the_reference_builds(data_dir):
    """load_graph: table sizes from the reference's own counting rules (lines 148-150), N(0, 1/d) initialisation, (graph,
    feature_modules) returned; the id -> row convention (users / communities id + 1, posts -> bags of word rows) reaches the
    encoder through get_encoder exactly as the reference's call sequence has it."""
    import torch
    from graphqembed_amd import reddit_data, utils
    from graphqembed_amd.encoders import DirectEncoder
    torch.manual_seed(0)
    graph, feats = reddit_data.load_graph(data_dir, 16)
    adj, rels, post_words = reddit_data.read_info(data_dir)
    num_users = len(set(i for rel, a in adj.items() for i in a if rel[0] == "user"))
    num_comm = len(set(i for rel, a in adj.items() for i in a if rel[0] == "community"))
    num_words = len(set(w for ws in post_words.values() for w in ws))
    assert isinstance(feats["post"], torch.nn.EmbeddingBag) and feats["post"].mode == "mean"
    assert tuple(feats["post"].weight.shape) == (num_words, 16)
    assert tuple(feats["user"].weight.shape) == (num_users + 1, 16) and tuple(feats["community"].weight.shape) == (num_comm + 1, 16)
    assert (num_users, num_comm) == (90, 12)
    assert abs(float(feats["user"].weight.detach().std()) - 1.0 / 16) < 0.01
    assert graph.relations == rels and set(graph.adj_lists) == set(adj) and graph.feature_dims == {m: 16 for m in rels}
    enc = utils.get_encoder(0, graph, {m: 16 for m in rels}, feats, False)
    assert isinstance(enc, DirectEncoder)
    assert np.array_equal(enc.rows([0, 5, 89], "user"), [1, 6, 90]) and np.array_equal(enc.rows([11], "community"), [12])
    with pytest.raises(KeyError):
        enc.rows([90], "user")                     # beyond nn.Embedding(num_users + 1): the reference fails at the lookup too
    ptr, ids = enc.bag_csr["post"]
    posts = list(post_words.keys())
    rows = enc.rows(posts[:7], "post")
    for p, r in zip(posts[:7], rows):
        assert sorted(ids[ptr[r]:ptr[r + 1]].tolist()) == sorted(post_words[p])
    assert sorted(k for k in dict(enc.named_parameters())) == ["feat-community.weight", "feat-post.weight", "feat-user.weight"]


def test_load_graph_refuses_what_the_reference_cannot_index(tmp_path, data_dir):
    from graphqembed_amd import reddit_data
    adj, rels, post_words = reddit_data.read_info(data_dir)
    gappy = {p: set(w + 1000 for w in ws) for p, ws in post_words.items()}       # ids beyond the EmbeddingBag's rows
    with pytest.raises(ValueError, match="word ids"):
        reddit_data.build(adj, rels, gappy, 8)
    some = dict(post_words)
    some[next(iter(some))] = set()
    with pytest.raises(ValueError, match="without words"):
        reddit_data.build(adj, rels, some, 8)


def test_query_files_load_and_converter_round_trips(data_dir, tmp_path):
    """The query pickles through data_utils (the reference's loaders) and through tools/convert_data.py --reddit +
    load_flat_graph / flatdata: the same graph, the same bags, and flat lists whose rows are the pickle queries' rows."""
    from graphqembed_amd import data_utils, flatdata, reddit_data, utils
    train = data_utils.load_queries_by_formula(os.path.join(data_dir, "train_edges.pkl"))
    assert list(train) == ["1-chain"] and sum(len(v) for v in train["1-chain"].values()) > 1000
    for i, types in ((2, {"2-chain", "2-inter"}), (3, {"3-chain", "3-inter", "3-inter_chain", "3-chain_inter"})):
        train.update(data_utils.load_queries_by_formula(os.path.join(data_dir, "train_queries_%d.pkl" % i)))
        held = data_utils.load_test_queries_by_formula(os.path.join(data_dir, "val_queries_%d-clean.pkl" % i))
        assert set(held["one_neg"]) == types and set(held["full_neg"]) <= types and held["full_neg"]
        for by in held["full_neg"].values():
            assert all(len(q.neg_samples) > 1 for qs in by.values() for q in qs)
    edges = data_utils.load_test_queries_by_formula(os.path.join(data_dir, "val_edges-split.pkl"))
    assert set(edges["one_neg"]) == {"1-chain"} and set(edges["full_neg"]) == {"1-chain"}

    out = str(tmp_path / "flat")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "convert_data.py"), "--reddit", data_dir, out], stdout=subprocess.DEVNULL)
    graph, feats = reddit_data.load_graph(data_dir, 8)
    fgraph, ffeats, flat = reddit_data.load_flat_graph(out, 8)
    assert {k: tuple(v.weight.shape) for k, v in feats.items()} == {k: tuple(v.weight.shape) for k, v in ffeats.items()}
    for rel, adj in graph.adj_lists.items():
        for u, neigh in adj.items():
            assert fgraph.adj_lists[rel].get(u, set()) == neigh
    assert {p: sorted(w) for p, w in graph.bags["post"].items()} == {p: sorted(w) for p, w in fgraph.bags["post"].items()}
    enc = utils.get_encoder(0, graph, {m: 8 for m in graph.relations}, feats, False)
    fenc = utils.get_encoder(0, fgraph, {m: 8 for m in fgraph.relations}, ffeats, False)
    ftrain = flatdata.load_queries_by_formula(os.path.join(out, "train_queries_3.npz"), flat)
    n = 0
    for qt, by in ftrain.items():
        for f, lst in by.items():
            qs = train[qt][f]
            assert len(lst) == len(qs)
            pool = lst.flat_pool
            if f.target_mode != "post":
                assert np.array_equal(fenc.flat_rows(pool.target, f.target_mode), enc.rows([q.target_node for q in qs], f.target_mode))
            if f.target_mode == "post":          # a post's table row is its bag: compare the bags themselves
                fr, r = fenc.flat_rows(pool.target, "post"), enc.rows([q.target_node for q in qs], "post")
                (fp, fi), (p, i) = fenc.bag_csr["post"], enc.bag_csr["post"]
                for a, b in list(zip(fr, r))[:20]:
                    assert sorted(fi[fp[a]:fp[a + 1]]) == sorted(i[p[b]:p[b + 1]])
            for k, m in enumerate(f.anchor_modes):
                if m != "post":
                    assert np.array_equal(fenc.flat_rows(pool.anchors[k], m), enc.rows([q.anchor_nodes[k] for q in qs], m))
            n += len(qs)
    assert n == 200
