"""Flat data formats + converter (graphqembed_amd/flatdata.py, tools/convert_data.py): round trips against the
reference-format objects, on the tiny synthetic graph and on the golden query fixture (serialised exactly as the
reference's query pickles are: (query_graph, neg_samples, hard_neg_samples), netquery/graph.py:93-100)."""
import os
import pickle
import subprocess
import sys

import numpy as np

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
