#!/usr/bin/env python3
"""One-time converter: the reference's pickles -> flat npz archives (graphqembed_amd/flatdata.py).

    python tools/convert_data.py <data_dir> <out_dir>

Reads <data_dir>/graph_data.pkl and every other *.pkl that holds a list of serialised queries
(train_edges.pkl, train_queries_2.pkl, val_queries_3.pkl, ... — netquery/bio/train.py:30-58), writes
<out_dir>/graph.npz and one <name>.npz per query file.  Python-2 pickles are read with latin1 decoding."""
import glob
import os
import pickle
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphqembed_amd import flatdata  # noqa: E402


def main(data_dir, out_dir):
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(data_dir, "graph_data.pkl"), "rb") as f:
        rels, adj_lists, node_maps = pickle.load(f, encoding="latin1")
    g = flatdata.FlatGraph.from_reference(rels, adj_lists, node_maps)
    g.save(os.path.join(out_dir, "graph.npz"))
    print("graph.npz: %d modes, %d relations, %d edges" % (len(g.modes), len(g.relations), sum(len(i) for i in g.idx)))
    for path in sorted(glob.glob(os.path.join(data_dir, "*.pkl"))):
        name = os.path.splitext(os.path.basename(path))[0]
        if name == "graph_data":
            continue
        with open(path, "rb") as f:
            raw = pickle.load(f, encoding="latin1")
        if not (isinstance(raw, list) and raw and isinstance(raw[0], tuple) and len(raw[0]) == 3):
            print("%s: not a query list, skipped" % name)
            continue
        pools = flatdata.convert_query_file(raw, g)
        flatdata.save_pools(os.path.join(out_dir, name + ".npz"), pools)
        print("%s.npz: %d queries, %d formulas" % (name, sum(p.n for v in pools.values() for p in v), sum(len(v) for v in pools.values())))


if __name__ == "__main__":
    if len(sys.argv) != 3:
        raise SystemExit(__doc__)
    main(sys.argv[1], sys.argv[2])
