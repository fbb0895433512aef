"""Formula -> launch descriptor, Query lists -> int32 structure-of-arrays.

The reference re-walks Python objects for every batch (list comprehensions over
``Query`` objects plus a dict lookup per node, netquery/model.py:74-103,
bio/data_utils.py:20-21 — ~25 % of its forward time).  Here a formula's
queries are turned ONCE into contiguous int32 row arrays; a batch is then a slice.
"""
from __future__ import annotations

import random

import numpy as np

from .engine import QTYPES
from .graph import _reverse_relation, CHAIN_TYPES


def table_key(mode):
    return "enc.feat-%s.weight" % mode


def rel_key(rel):
    return "path_dec." + "_".join(rel)


def pre_key(mode):
    return "inter_dec.%s_premat" % mode


def post_key(mode):
    return "inter_dec.%s_postmat" % mode


class FormulaPlan(object):
    """Static part of a gqe_batch for one Formula + the parameter tensors it touches.

    chain types : hops[0] = rels in target->anchor order (applied on the target side,
                  netquery/decoders.py:142-147 / 200-205 / 228-233).
    inter types : hops[i] = reversed relations in the order model.py:80-91,102-105
                  applies them to anchor i; ``final`` = rev(rels[0]) for 3-chain_inter
                  (model.py:107); intersection mode = target mode, except 3-chain_inter
                  where it is the mode of the intersection variable (model.py:106).
    """

    def __init__(self, formula, layout, inter_kind):
        qt, rels = formula.query_type, formula.rels
        self.formula = formula
        self.qtype = QTYPES[qt]
        self.target_mode = formula.target_mode
        self.anchor_modes = tuple(formula.anchor_modes)
        hop_rels, final_rel, inter_mode = [], None, None
        if qt in CHAIN_TYPES:
            hop_rels = [list(rels)]
        elif qt in ("2-inter", "3-inter"):
            hop_rels = [[_reverse_relation(r)] for r in rels]
            inter_mode = formula.target_mode
        elif qt == "3-inter_chain":
            hop_rels = [[_reverse_relation(rels[0])], [_reverse_relation(r) for r in rels[1][::-1]]]
            inter_mode = formula.target_mode
        else:  # 3-chain_inter
            hop_rels = [[_reverse_relation(rels[1][0])], [_reverse_relation(rels[1][1])]]
            inter_mode = rels[0][-1]
            final_rel = _reverse_relation(rels[0])
        keys = {table_key(self.target_mode)}
        keys.update(table_key(m) for m in self.anchor_modes)
        for br in hop_rels:
            keys.update(rel_key(r) for r in br)
        self.static = {
            "qtype": self.qtype,
            "n_anchors": len(self.anchor_modes),
            "target_table": layout.offset(table_key(self.target_mode)),
            "anchor_table": [layout.offset(table_key(m)) for m in self.anchor_modes],
            "hops": [[layout.offset(rel_key(r)) for r in br] for br in hop_rels],
        }
        if final_rel is not None:
            self.static["final"] = layout.offset(rel_key(final_rel))
            keys.add(rel_key(final_rel))
        if inter_mode is not None and not inter_kind.endswith("simple"):
            self.static["pre"] = layout.offset(pre_key(inter_mode))
            self.static["post"] = layout.offset(post_key(inter_mode))
            keys.update((pre_key(inter_mode), post_key(inter_mode)))
        self.touched = frozenset(keys)

    def batch(self, n, idx_offset, out_offset, weight=1.0, margin=1.0):
        d = dict(self.static)
        d.update(n=int(n), idx_offset=int(idx_offset), out_offset=int(out_offset),
                 weight=float(weight), margin=float(margin))
        return d


def pack_margin_batches(items):
    """items: [(plan, target[n], neg[n], anchors[k,n], weight, margin)] ->
    (descs, idx int32[...], n_scores).  Index layout per batch: target | neg | anchors."""
    descs, chunks, off, out = [], [], 0, 0
    for plan, target, neg, anchors, weight, margin in items:
        n = len(target)
        descs.append(plan.batch(n, off, out, weight, margin))
        chunks.extend((np.asarray(target, np.int32), np.asarray(neg, np.int32),
                       np.asarray(anchors, np.int32).reshape(-1)))
        off += (2 + plan.static["n_anchors"]) * n
        out += n
    return descs, np.concatenate(chunks), out


def pack_forward_batches(items):
    """items: [(plan, target[n], anchors[k,n])] -> (descs, idx, n_scores); layout target | anchors."""
    descs, chunks, off, out = [], [], 0, 0
    for plan, target, anchors in items:
        n = len(target)
        descs.append(plan.batch(n, off, out))
        chunks.extend((np.asarray(target, np.int32), np.asarray(anchors, np.int32).reshape(-1)))
        off += (1 + plan.static["n_anchors"]) * n
        out += n
    return descs, np.concatenate(chunks), out


def pack_candidate_batches(items):
    """items: [(plan, anchors[k,n], cand_ptr[n+1], cand_rows[nnz])] -> (descs, idx, n_scores): evaluation
    against candidate lists; per batch the index layout is anchors | cand_ptr | cand_rows and the scores come
    back flat in candidate order."""
    descs, chunks, off, out = [], [], 0, 0
    for plan, anchors, ptr, rows in items:
        n = len(ptr) - 1
        d = plan.batch(n, off, out)
        d["n_candidates"] = int(len(rows))
        descs.append(d)
        chunks.extend((np.asarray(anchors, np.int32).reshape(-1), np.asarray(ptr, np.int32), np.asarray(rows, np.int32)))
        off += plan.static["n_anchors"] * n + n + 1 + len(rows)
        out += len(rows)
    return descs, np.concatenate(chunks), out


class FormulaQueries(object):
    """All queries of one Formula as int32 row arrays (+ CSR negatives)."""

    def __init__(self, formula, queries, enc):
        self.formula = formula
        self.n = len(queries)
        self.target = enc.rows([q.target_node for q in queries], formula.target_mode)
        self.anchors = np.stack([enc.rows([q.anchor_nodes[i] for q in queries], m)
                                 for i, m in enumerate(formula.anchor_modes)])
        self.neg_ptr, self.neg_rows = self._csr([q.neg_samples for q in queries], enc, formula.target_mode)
        self.hard_ptr, self.hard_rows = self._csr([q.hard_neg_samples for q in queries], enc, formula.target_mode)

    @staticmethod
    def _csr(lists, enc, mode):
        if any(l is None for l in lists):
            return None, None
        ptr = np.zeros(len(lists) + 1, dtype=np.int64)
        ptr[1:] = np.cumsum([len(l) for l in lists])
        flat = [x for l in lists for x in l]
        return ptr, enc.rows(flat, mode)

    def sample_negatives(self, start, end, hard, rng, all_rows=None):
        """One negative row per query of the slice, uniformly from its stored list
        (hard / regular), or — 1-chain training — uniformly from ``all_rows``
        (every node of the target mode, netquery/model.py:116-120)."""
        n = end - start
        if all_rows is not None:
            return all_rows[rng.randint(0, len(all_rows), size=n)]
        ptr, rows = (self.hard_ptr, self.hard_rows) if hard else (self.neg_ptr, self.neg_rows)
        if ptr is None:
            raise Exception("queries of formula %s carry no %snegative samples" % (self.formula, "hard " if hard else ""))
        lo = ptr[start:end]
        cnt = ptr[start + 1:end + 1] - lo
        return rows[lo + (rng.random_sample(n) * cnt).astype(np.int64)]


def reference_negative_nodes(graph, formula, queries, hard_negatives):
    """The reference's negative choice, call for call (netquery/model.py:113-120), so
    that the same ``random`` seed reproduces the same negatives."""
    if "inter" not in formula.query_type and hard_negatives:
        raise Exception("Hard negative examples can only be used with intersection queries")
    if hard_negatives:
        return [random.choice(q.hard_neg_samples) for q in queries]
    if formula.query_type == "1-chain":
        full = graph.full_lists[formula.target_mode]
        return [random.choice(full) for _ in queries]
    return [random.choice(q.neg_samples) for q in queries]


class PoolRows(object):
    """The table rows of ONE formula's query list, looked up once (the tensorize step): target rows [n],
    anchor rows [k, n], and — built when first asked for — the negative / hard-negative lists as CSR arrays of rows.  The
    list is kept alive so that its identity stays a valid cache key; it is assumed not to change."""

    def __init__(self, model, formula, pool):
        enc = model.enc
        self.pool, self.formula, self.n = pool, formula, len(pool)
        self._csr = {}
        flat = getattr(pool, "flat_pool", None)
        if flat is not None:        # flatdata.PoolQueryList: the arrays are the list (rows of the converted files = index + 1)
            # (an EmbeddingBag mode's table rows are bag indices: DirectEncoder.flat_rows translates; identity elsewhere)
            conv = getattr(enc, "flat_rows", lambda r, m: r)
            self.target = conv(flat.target, formula.target_mode)
            self.anchors = np.stack([conv(flat.anchors[i], m) for i, m in enumerate(formula.anchor_modes)])
            for hard, ptr, rows in ((False, flat.neg_ptr, flat.neg_rows), (True, flat.hard_ptr, flat.hard_rows)):
                ptr = np.asarray(ptr, dtype=np.int64)
                ok = len(ptr) > 1 and (ptr[1:] > ptr[:-1]).all()
                self._csr[hard] = (ptr, conv(np.asarray(rows, dtype=np.int32), formula.target_mode)) if ok else None
            return
        self.target = enc.rows([q.target_node for q in pool], formula.target_mode)
        self.anchors = np.stack([enc.rows([q.anchor_nodes[i] for q in pool], m) for i, m in enumerate(formula.anchor_modes)])
        self._probe = self._take_probe()

    def _take_probe(self):
        """Identity of the first / middle / last query and of their negative lists (+ the lists' lengths): what an in-place
        shuffle, a re-sampling of the negatives or an edit of the list almost surely changes."""
        pool = self.pool
        if getattr(pool, "flat_pool", None) is not None or len(pool) == 0:
            return None
        out = []
        for i in (0, len(pool) // 2, len(pool) - 1):
            q = pool[i]
            for l in (getattr(q, "neg_samples", None), getattr(q, "hard_neg_samples", None)):
                out.append((id(q), id(l), -1 if l is None else len(l)))
        return tuple(out)

    def still_valid(self):
        return self.n == len(self.pool) and getattr(self, "_probe", None) == self._take_probe()

    def lists(self, model, hard):
        """(ptr[n + 1], rows) of every query's negative (hard-negative) list, or None if some query has none."""
        if hard not in self._csr:
            lists = [(q.hard_neg_samples if hard else q.neg_samples) for q in self.pool]
            if any(l is None or len(l) == 0 for l in lists):
                self._csr[hard] = None
            else:
                ptr = np.zeros(len(lists) + 1, dtype=np.int64)
                ptr[1:] = np.cumsum([len(l) for l in lists])
                rows = model.enc.rows([n for l in lists for n in l], self.formula.target_mode)
                self._csr[hard] = (ptr, rows)
        return self._csr[hard]
