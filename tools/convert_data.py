#!/usr/bin/env python3
"""One-time converter: the reference's pickles -> flat npz archives (graphqembed_amd/flatdata.py).

    python tools/convert_data.py <data_dir> <out_dir>
    python tools/convert_data.py --reddit <data_dir> <out_dir>

Reads <data_dir>/graph_data.pkl and every other *.pkl that holds a list of serialised queries
(train_edges.pkl, train_queries_2.pkl, val_queries_3.pkl, ... — netquery/bio/train.py:30-58), writes
<out_dir>/graph.npz and one <name>.npz per query file.  Python-2 pickles are read with latin1 decoding.

--reddit: the Reddit data set's layout (netquery/reddit/data_utils_new.py:143-147, reddit/new_train.py:33-46): the graph is
adj_lists.pkl + rels.pkl + post_words.pkl (node ids per mode); writes graph.npz, post_words.npz (CSR of every post's word ids)
and the query files (train_edges, val_edges-split, test_queries_2-clean, ...).  graphqembed_amd.reddit_data.load_flat_graph reads
the result."""
import glob
import os
import pickle
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphqembed_amd import flatdata  # noqa: E402


GRAPH_FILES = ("graph_data", "adj_lists", "rels", "post_words")


def main(data_dir, out_dir, reddit=False):
    os.makedirs(out_dir, exist_ok=True)
    if reddit:
        from graphqembed_amd import reddit_data
        adj_lists, rels, post_words = reddit_data.read_info(data_dir)
        node_maps = reddit_data.flat_node_maps(adj_lists, post_words)
        reddit_data.save_post_words(os.path.join(out_dir, "post_words.npz"), post_words, node_maps[reddit_data.BAG_MODE])
        print("post_words.npz: %d posts, %d word occurrences" % (len(post_words), sum(len(w) for w in post_words.values())))
    else:
        with open(os.path.join(data_dir, "graph_data.pkl"), "rb") as f:
            rels, adj_lists, node_maps = pickle.load(f, encoding="latin1")
    g = flatdata.FlatGraph.from_reference(rels, adj_lists, node_maps)
    g.save(os.path.join(out_dir, "graph.npz"))
    print("graph.npz: %d modes, %d relations, %d edges" % (len(g.modes), len(g.relations), sum(len(i) for i in g.idx)))
    for path in sorted(glob.glob(os.path.join(data_dir, "*.pkl"))):
        name = os.path.splitext(os.path.basename(path))[0]
        if name in GRAPH_FILES:
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
    args = [a for a in sys.argv[1:] if a != "--reddit"]
    if len(args) != 2:
        raise SystemExit(__doc__)
    main(args[0], args[1], reddit="--reddit" in sys.argv[1:])
