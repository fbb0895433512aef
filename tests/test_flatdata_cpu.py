"""Flat data formats + converter (graphqembed_amd/flatdata.py, tools/convert_data.py): round trips against the
reference-format objects, on the tiny synthetic graph and on the golden query fixture (serialised exactly as the
reference's query pickles are: (query_graph, neg_samples, hard_neg_samples), netquery/graph.py:93-100)."""
import os
import pickle
import subprocess
import sys

import numpy as np
import pytest

from graphqembed_amd import data_utils, flatdata, graph as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def _world():
    rel, adj, ids = data_utils.make_synthetic_graph(data_utils.BIO_TINY_SIZES, edges_per_kind=data_utils.BIO_TINY_EDGES_PER_KIND, seed=0)
    return rel, adj, data_utils.make_node_maps(ids)


def test_graph_round_trip(tmp_path):
    rel, adj, maps = _world()
    g = flatdata.FlatGraph.from_reference(rel, adj, maps)
    g.save(tmp_path / "graph.npz")
    g2 = flatdata.FlatGraph.load(tmp_path / "graph.npz")
    rel2, adj2, maps2 = g2.to_reference()
    assert {m: sorted(v) for m, v in rel2.items()} == {m: sorted(v) for m, v in rel.items()}
    assert maps2 == maps
    for r, a in adj.items():
        assert {u: s for u, s in a.items() if s} == dict(adj2[r])
    # the 1-chain negative universe = Graph.full_lists (as rows)
    ref = G.Graph(None, {m: 8 for m in rel}, rel, adj)
    rows = g2.all_rows()
    for m in ref.full_lists:
        assert sorted(rows[m]) == sorted(maps[m][n] + 1 for n in ref.full_lists[m])


def test_query_file_conversion_matches_the_object_path(tmp_path):
    from graphqembed_amd.encoders import DirectEncoder
    from graphqembed_amd.tensorize import FormulaQueries
    rel, adj, maps = _world()
    g = flatdata.FlatGraph.from_reference(rel, adj, maps)
    with open(os.path.join(GOLDEN, "queries_tiny.pkl"), "rb") as f:
        data = pickle.load(f)
    raw = [info for infos in data["train"].values() for info in infos]
    pools = flatdata.convert_query_file(raw, g)
    flatdata.save_pools(tmp_path / "train.npz", pools)
    pools = flatdata.load_pools(tmp_path / "train.npz")
    assert sum(p.n for v in pools.values() for p in v) == len(raw)
    enc = DirectEncoder(None, {}, node_maps=maps)
    by_formula = data_utils.group_by_formula([G.Query.deserialize(i) for i in raw])
    for qt in pools:
        for p in pools[qt]:
            fq = FormulaQueries(p.formula, by_formula[qt][p.formula], enc)
            assert np.array_equal(fq.target, p.target) and np.array_equal(fq.anchors, p.anchors)
            if fq.neg_ptr is None:          # edges carry no stored negatives (drawn from the whole mode, model.py:118)
                assert p.neg_ptr[-1] == 0
            else:
                assert np.array_equal(fq.neg_ptr, p.neg_ptr) and np.array_equal(fq.neg_rows, p.neg_rows)
            if "inter" in qt:
                assert np.array_equal(fq.hard_ptr, p.hard_ptr) and np.array_equal(fq.hard_rows, p.hard_rows)
    # and back to Query objects
    back = flatdata.pools_to_queries(pools, g)
    tup = lambda x: None if x is None else tuple(x)
    key = lambda q: (q.formula, q.target_node, q.anchor_nodes, tup(q.neg_samples), tup(q.hard_neg_samples))
    want = {key(q) for qs in by_formula.values() for ql in qs.values() for q in ql}
    got = {key(q) for q in back}
    assert got == want


def test_converter_cli_on_python2_style_pickles(tmp_path):
    rel, adj, maps = _world()
    src, dst = tmp_path / "data", tmp_path / "flat"
    src.mkdir()
    with open(src / "graph_data.pkl", "wb") as f:            # protocol 2 = what Python 2 wrote
        pickle.dump((rel, {k: dict(v) for k, v in adj.items()}, maps), f, protocol=2)
    with open(os.path.join(GOLDEN, "queries_tiny.pkl"), "rb") as f:
        data = pickle.load(f)
    with open(src / "train_queries_2.pkl", "wb") as f:
        pickle.dump(data["train"]["2-inter"] + data["train"]["2-chain"], f, protocol=2)
    with open(src / "notes.pkl", "wb") as f:
        pickle.dump({"not": "queries"}, f, protocol=2)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "convert_data.py"), str(src), str(dst)],
                         capture_output=True, text=True, check=True).stdout
    assert "graph.npz" in out and "train_queries_2.npz" in out and "notes: not a query list" in out
    pools = flatdata.load_pools(dst / "train_queries_2.npz")
    assert set(pools) == {"2-inter", "2-chain"}
    g = flatdata.FlatGraph.load(dst / "graph.npz")
    assert g.modes == sorted(rel.keys())


def test_native_sampler_from_flat_graph():
    from graphqembed_amd.sampler import NativeSampler
    rel, adj, maps = _world()
    ref = G.Graph(None, {m: 8 for m in rel}, rel, adj)
    s = NativeSampler.from_flat(flatdata.FlatGraph.from_reference(rel, adj, maps))
    for q in s.sample(80, q_type="3-inter_chain", neg_sample_max=10 ** 6, seed=1).to_queries():
        assert ref._is_subgraph(q.query_graph)
        negs, hard = ref.get_negative_samples(q.query_graph)
        assert set(q.neg_samples) == set(negs) and set(q.hard_neg_samples) == set(hard)


def test_flat_query_lists_stand_in_for_query_lists(tmp_path):
    """flatdata.load_queries_by_formula / load_test_queries_by_formula on converted files: the dictionaries run_train and eval_*
    take, with PoolQueryList values — tensorize.PoolRows built from them (no Query object) equals PoolRows built from the
    reference-format Query lists; the test split follows the reference's rule (more than one stored negative -> full_neg,
    netquery/data_utils.py:27-35); indexing a list yields Query objects with the real node ids."""
    from graphqembed_amd.encoders import DirectEncoder
    from graphqembed_amd.tensorize import PoolRows
    rel, adj, maps = _world()
    g = flatdata.FlatGraph.from_reference(rel, adj, maps)
    with open(os.path.join(GOLDEN, "queries_tiny.pkl"), "rb") as f:
        data = pickle.load(f)

    class M(object):
        enc = DirectEncoder(None, {}, node_maps=maps)
    # training file
    raw = [info for infos in data["train"].values() for info in infos]
    flatdata.save_pools(tmp_path / "train.npz", flatdata.convert_query_file(raw, g))
    flat = flatdata.load_queries_by_formula(tmp_path / "train.npz", g)
    objs = data_utils.group_by_formula([G.Query.deserialize(i) for i in raw])
    assert set(flat) == set(objs)
    n = 0
    for qt in objs:
        assert set(flat[qt]) == set(objs[qt])
        for f in objs[qt]:
            a, b = PoolRows(M, f, flat[qt][f]), PoolRows(M, f, objs[qt][f])
            assert len(flat[qt][f]) == len(objs[qt][f]) == a.n == b.n
            assert np.array_equal(a.target, b.target) and np.array_equal(a.anchors, b.anchors)
            for hard in (False, True):
                x, y = a.lists(M, hard), b.lists(M, hard)
                assert (x is None) == (y is None), (qt, hard)
                if x is not None:
                    assert np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1])
            q0, r0 = flat[qt][f][0], objs[qt][f][0]
            assert (q0.target_node, q0.anchor_nodes, q0.formula) == (r0.target_node, r0.anchor_nodes, r0.formula)
            assert [q.target_node for q in flat[qt][f][1:3]] == [q.target_node for q in objs[qt][f][1:3]]
            n += 1
    assert n >= 10
    # held-out file: the reference's split
    raw_test = [info for infos in data["test"].values() for info in infos]
    flatdata.save_pools(tmp_path / "test.npz", flatdata.convert_query_file(raw_test, g))
    flat_t = flatdata.load_test_queries_by_formula(tmp_path / "test.npz", g)
    objs_t = data_utils.split_test_queries(raw_test)
    for key in ("one_neg", "full_neg"):
        assert set(flat_t[key]) == set(objs_t[key])
        for qt in objs_t[key]:
            assert set(flat_t[key][qt]) == set(objs_t[key][qt])
            for f in objs_t[key][qt]:
                a, b = PoolRows(M, f, flat_t[key][qt][f]), PoolRows(M, f, objs_t[key][qt][f])
                # (grouping by formula keeps the file's order inside a formula on both sides)
                assert np.array_equal(a.target, b.target) and np.array_equal(a.anchors, b.anchors)
                # (Query.deserialize hands a list of n negatives to random.sample(.., n), as the reference does, graph.py:99-100: every
                # load draws a new order; the converted file keeps the one its conversion drew — the same SETS per query)
                x, y = a.lists(M, False), b.lists(M, False)
                assert np.array_equal(x[0], y[0])
                for i in range(a.n):
                    assert sorted(x[1][x[0][i]:x[0][i + 1]]) == sorted(y[1][y[0][i]:y[0][i + 1]])
    # a list without its graph still trains (arrays only) but cannot hand out Query objects
    bare = flatdata.load_queries_by_formula(tmp_path / "train.npz")
    some = next(iter(next(iter(bare.values())).values()))
    with pytest.raises(Exception, match="row arrays only"):
        some[0]


def test_flat_query_list_slices_stay_flat(tmp_path):
    """A slice of a PoolQueryList is a PoolQueryList over the sub-arrays (targets, anchors, re-based CSR negatives): held-out splits
    of sampled or converted lists never build Query objects; NativeSampler(...).sample(...).query_lists() hands the sampler's
    output over in that form."""
    rel, adj, maps = _world()
    g = flatdata.FlatGraph.from_reference(rel, adj, maps)
    with open(os.path.join(GOLDEN, "queries_tiny.pkl"), "rb") as f:
        data = pickle.load(f)
    raw = [info for infos in data["test"].values() for info in infos]
    flatdata.save_pools(tmp_path / "t.npz", flatdata.convert_query_file(raw, g))
    lists = flatdata.load_queries_by_formula(tmp_path / "t.npz", g)
    checked = 0
    for by in lists.values():
        for f, l in by.items():
            if len(l) < 4:
                continue
            for sl in (slice(1, 3), slice(None, -1), slice(-2, None), slice(0, 0)):
                sub = l[sl]
                assert isinstance(sub, flatdata.PoolQueryList) and len(sub) == len(range(len(l))[sl])
                whole = l.queries()[sl]
                got = sub.queries() if len(sub) else []
                assert [(q.target_node, q.anchor_nodes, sorted(q.neg_samples or [])) for q in got] == \
                       [(q.target_node, q.anchor_nodes, sorted(q.neg_samples or [])) for q in whole]
            with pytest.raises(Exception, match="step 1"):
                l[::2]
            checked += 1
    assert checked >= 3
