"""Flat on-disk formats + the one-time converter from the reference's pickles.

The reference ships Python-2 pickles of nested Python objects:
    graph_data.pkl            (rels, adj_lists, node_maps)                      netquery/bio/data_utils.py:12
    {train,val,test}_*.pkl    [(query_graph, neg_samples, hard_neg_samples)]    netquery/graph.py:93-100,
                                                                                 netquery/data_utils.py:10-35
which it re-walks object by object every batch.  Converted ONCE (run the converter where the data lives:
``python tools/convert_data.py <data_dir> <out_dir>``), the same information is

    graph.npz     modes, relations, per-relation CSR over LOCAL node indices (index = node_maps value, table row =
                  index + 1), the node id of every index
    <name>.npz    per formula: target rows[n], anchor rows[k, n], CSR negative rows, CSR hard-negative rows

— exactly the int32 arrays the fused kernel's index feed is sliced from (tensorize.pack_margin_batches) and the
native sampler's CSR (include/gqe_sampler.h).  Everything is a plain ``numpy.savez`` archive; structure that is not
an array (mode names, relation triples, formulas) travels as one JSON string.
"""
from __future__ import annotations

import json

import numpy as np

from .graph import Formula, Query
from .sampler import FormulaPool


def _jsonable(rels):
    return [_jsonable(r) for r in rels] if isinstance(rels, (tuple, list)) and rels and isinstance(rels[0], (tuple, list)) else list(rels)


def _rels_from_json(x):
    if x and isinstance(x[0], list):
        return tuple(_rels_from_json(r) for r in x)
    return tuple(x)


# ------------------------------------------------------------------------------------------------
# graph
# ------------------------------------------------------------------------------------------------
class FlatGraph(object):
    """modes[M], sizes[M], relations [(src mode, name, dst mode)], CSR per relation (local indices), node ids."""

    def __init__(self, modes, sizes, relations, ptr, idx, node_ids):
        self.modes, self.sizes, self.relations = list(modes), [int(s) for s in sizes], [tuple(r) for r in relations]
        self.ptr, self.idx, self.node_ids = ptr, idx, node_ids

    @classmethod
    def from_reference(cls, rels, adj_lists, node_maps):
        """From the contents of graph_data.pkl.  ``node_maps[mode][node] = index`` (the extra -1 entry is ignored)."""
        modes = sorted(rels.keys())
        maps = {m: {n: i for n, i in node_maps[m].items() if n >= 0 and i >= 0} for m in modes}
        sizes = [max(maps[m].values()) + 1 if maps[m] else 0 for m in modes]
        relations = []
        for m in modes:
            for to, name in rels[m]:
                if (m, name, to) not in relations:
                    relations.append((m, name, to))
        for r in list(adj_lists.keys()):
            if tuple(r) not in relations:
                relations.append(tuple(r))
        ptrs, idxs = [], []
        for (a, name, b) in relations:
            adj = adj_lists.get((a, name, b), {})
            deg = np.zeros(sizes[modes.index(a)] + 1, dtype=np.int64)
            for u, neigh in adj.items():
                deg[maps[a][u] + 1] = len(neigh)
            ptr = np.cumsum(deg)
            idx = np.zeros(int(ptr[-1]), dtype=np.int32)
            for u, neigh in adj.items():
                if neigh:
                    p = ptr[maps[a][u]]
                    idx[p:p + len(neigh)] = sorted(maps[b][v] for v in neigh)
            ptrs.append(ptr)
            idxs.append(idx)
        node_ids = []
        for m, size in zip(modes, sizes):
            arr = np.full(size, -1, dtype=np.int64)
            for n, i in maps[m].items():
                arr[i] = n
            node_ids.append(arr)
        return cls(modes, sizes, relations, ptrs, idxs, node_ids)

    def save(self, path):
        arrays = {"meta": np.array(json.dumps({"modes": self.modes, "sizes": self.sizes,
                                               "relations": [list(r) for r in self.relations]}))}
        for k, (p, i) in enumerate(zip(self.ptr, self.idx)):
            arrays["ptr%d" % k], arrays["idx%d" % k] = p, i
        for k, n in enumerate(self.node_ids):
            arrays["nodes%d" % k] = n
        np.savez(path, **arrays)

    @classmethod
    def load(cls, path):
        z = np.load(path, allow_pickle=False)
        meta = json.loads(str(z["meta"]))
        nr, nm = len(meta["relations"]), len(meta["modes"])
        return cls(meta["modes"], meta["sizes"], meta["relations"], [z["ptr%d" % k] for k in range(nr)],
                   [z["idx%d" % k] for k in range(nr)], [z["nodes%d" % k] for k in range(nm)])

    def to_reference(self):
        """Back to ``(rels, adj_lists, node_maps)`` — the reference's in-memory form (for Graph / tests)."""
        from collections import defaultdict
        rels = defaultdict(list)
        adj_lists = {}
        mid = {m: k for k, m in enumerate(self.modes)}
        for k, (a, name, b) in enumerate(self.relations):
            if (b, name) not in rels[a]:
                rels[a].append((b, name))
            adj = defaultdict(set)
            ids_a, ids_b = self.node_ids[mid[a]], self.node_ids[mid[b]]
            ptr, idx = self.ptr[k], self.idx[k]
            for u in np.nonzero(np.diff(ptr))[0]:
                adj[int(ids_a[u])] = set(int(x) for x in ids_b[idx[ptr[u]:ptr[u + 1]]])
            adj_lists[(a, name, b)] = adj
        node_maps = {m: {int(n): i for i, n in enumerate(self.node_ids[mid[m]]) if n >= 0} for m in self.modes}
        for m in node_maps:
            node_maps[m][-1] = -1
        return dict(rels), adj_lists, node_maps

    def all_rows(self):
        """{mode: int32 table rows of every node that has an edge} — the 1-chain negative universe (model.py:118)."""
        out = {}
        for k, m in enumerate(self.modes):
            seen = np.zeros(self.sizes[k], dtype=bool)
            for r, (a, _, b) in enumerate(self.relations):
                if a == m:
                    seen[np.nonzero(np.diff(self.ptr[r]))[0]] = True
            out[m] = (np.nonzero(seen)[0] + 1).astype(np.int32)
        return out


# ------------------------------------------------------------------------------------------------
# queries
# ------------------------------------------------------------------------------------------------
def pools_from_queries(queries, row_of):
    """[Query] -> {query_type: [FormulaPool]}; ``row_of(nodes, mode) -> int32 rows`` (DirectEncoder.rows)."""
    groups = {}
    for q in queries:
        groups.setdefault(q.formula, []).append(q)
    out = {}
    for f, qs in groups.items():
        target = row_of([q.target_node for q in qs], f.target_mode)
        anchors = np.stack([row_of([q.anchor_nodes[i] for q in qs], m) for i, m in enumerate(f.anchor_modes)])

        def csr(lists):
            ptr = np.zeros(len(lists) + 1, dtype=np.int64)
            ptr[1:] = np.cumsum([0 if l is None else len(l) for l in lists])
            flat = [x for l in lists if l is not None for x in l]
            return ptr, (row_of(flat, f.target_mode) if flat else np.zeros(0, dtype=np.int32))
        out.setdefault(f.query_type, []).append(FormulaPool(f, target, anchors, *csr([q.neg_samples for q in qs]),
                                                           *csr([q.hard_neg_samples for q in qs])))
    return out


def save_pools(path, pools):
    arrays, meta = {}, []
    k = 0
    for qt in sorted(pools):
        for p in pools[qt]:
            meta.append({"query_type": qt, "rels": _jsonable(p.formula.rels)})
            arrays["target%d" % k], arrays["anchors%d" % k] = p.target, p.anchors
            arrays["neg_ptr%d" % k], arrays["neg%d" % k] = p.neg_ptr, p.neg_rows
            arrays["hard_ptr%d" % k], arrays["hard%d" % k] = p.hard_ptr, p.hard_rows
            k += 1
    arrays["meta"] = np.array(json.dumps(meta))
    np.savez(path, **arrays)


def load_pools(path):
    z = np.load(path, allow_pickle=False)
    out = {}
    for k, m in enumerate(json.loads(str(z["meta"]))):
        f = Formula(m["query_type"], _rels_from_json(m["rels"]))
        out.setdefault(f.query_type, []).append(FormulaPool(f, z["target%d" % k], z["anchors%d" % k], z["neg_ptr%d" % k],
                                                           z["neg%d" % k], z["hard_ptr%d" % k], z["hard%d" % k]))
    return out


def pools_to_queries(pools, graph):
    """{type: [FormulaPool]} -> [Query] with real node ids (query graphs are not stored: chains lose their
    intermediate variables, which nothing downstream of sampling reads — model.py:70-109 uses anchors + target)."""
    mid = {m: k for k, m in enumerate(graph.modes)}
    out = []
    for qt in pools:
        for p in pools[qt]:
            f = p.formula
            tid = graph.node_ids[mid[f.target_mode]]
            for i in range(p.n):
                q = Query.__new__(Query)
                q.formula = f
                q.target_node = int(tid[p.target[i] - 1])
                q.anchor_nodes = tuple(int(graph.node_ids[mid[m]][p.anchors[k, i] - 1]) for k, m in enumerate(f.anchor_modes))
                q.query_graph = None
                negs = p.neg_rows[p.neg_ptr[i]:p.neg_ptr[i + 1]]
                q.neg_samples = [int(x) for x in tid[negs - 1]] if len(negs) else None
                hard = p.hard_rows[p.hard_ptr[i]:p.hard_ptr[i + 1]]
                q.hard_neg_samples = [int(x) for x in tid[hard - 1]] if "inter" in qt else None
                out.append(q)
    return out


def convert_query_file(raw_infos, flat_graph):
    """The list a reference query pickle holds -> pools (rows through the graph's node index)."""
    mid = {m: k for k, m in enumerate(flat_graph.modes)}
    luts = {}
    for m, k in mid.items():
        ids = flat_graph.node_ids[k]
        luts[m] = {int(n): i + 1 for i, n in enumerate(ids) if n >= 0}

    def row_of(nodes, mode):
        lut = luts[mode]
        return np.fromiter((lut[n] for n in nodes), dtype=np.int32, count=len(nodes))
    return pools_from_queries([Query.deserialize(info) for info in raw_infos], row_of)


# ------------------------------------------------------------------------------------------------
# flat query lists behind the reference's dictionaries: {type: {Formula: list}} for run_train / eval_* without one Python object
# per query
# ------------------------------------------------------------------------------------------------
class PoolQueryList(object):
    """One formula's queries as the list ``run_train`` / ``eval_auc_queries`` / ``eval_perc_queries`` expect — ``len``, indexing,
    slicing, iteration — backed by a ``FormulaPool`` of row arrays.  ``QueryEncoderDecoder.pool_rows`` takes the arrays as they
    are (no per-query lookups: the training windows, the native loop and the cached evaluation never build a ``Query``); anything
    that does index the list (the per-Query fallback paths) gets ``Query`` objects with real node ids, built on first use
    (``pools_to_queries``: needs the ``FlatGraph`` for the ids)."""

    def __init__(self, pool, flat_graph=None):
        self.flat_pool, self.flat_graph = pool, flat_graph
        self._queries = None

    def __len__(self):
        return self.flat_pool.n

    def queries(self):
        if self._queries is None:
            if self.flat_graph is None:
                raise Exception("this query list is backed by row arrays only: pass the FlatGraph to its loader to get Query objects")
            self._queries = pools_to_queries({self.flat_pool.formula.query_type: [self.flat_pool]}, self.flat_graph)
        return self._queries

    def __getitem__(self, i):
        if isinstance(i, slice):     # a slice of the list is again a list of row arrays (held-out splits, windows)
            keep = np.zeros(self.flat_pool.n, dtype=bool)
            keep[i] = True
            if i.step not in (None, 1):
                raise Exception("PoolQueryList slices keep the list's order: step 1 only")
            return PoolQueryList(_take(self.flat_pool, keep), self.flat_graph)
        return self.queries()[i]

    def __iter__(self):
        return iter(self.queries())


def _take(pool, mask):
    """The sub-pool of the queries selected by a boolean mask (CSR lists re-based)."""
    keep = np.flatnonzero(mask)

    def csr(ptr, rows):
        lens = (ptr[1:] - ptr[:-1])[keep]
        out = np.zeros(len(keep) + 1, dtype=np.int64)
        out[1:] = np.cumsum(lens)
        src = np.repeat(ptr[:-1][keep] - out[:-1], lens) + np.arange(int(out[-1]))
        return out, rows[src]
    return FormulaPool(pool.formula, pool.target[keep], pool.anchors[:, keep], *csr(pool.neg_ptr, pool.neg_rows), *csr(pool.hard_ptr, pool.hard_rows))


def load_queries_by_formula(path, flat_graph=None):
    """``data_utils.load_queries_by_formula`` on a converted file: {query type: {Formula: PoolQueryList}} (netquery/data_utils.py:
    37-44)."""
    out = {}
    for qt, pools in load_pools(path).items():
        out[qt] = {p.formula: PoolQueryList(p, flat_graph) for p in pools}
    return out


def load_test_queries_by_formula(path, flat_graph=None):
    """``data_utils.load_test_queries_by_formula`` on a converted file: a query with more than one stored negative goes to
    "full_neg", the others to "one_neg" (netquery/data_utils.py:27-35)."""
    out = {"full_neg": {}, "one_neg": {}}
    for qt, pools in load_pools(path).items():
        for p in pools:
            many = (p.neg_ptr[1:] - p.neg_ptr[:-1]) > 1
            for key, mask in (("full_neg", many), ("one_neg", ~many)):
                if mask.any():
                    out[key].setdefault(qt, {})[p.formula] = PoolQueryList(_take(p, mask) if not mask.all() else p, flat_graph)
    return out
