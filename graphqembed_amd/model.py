"""``QueryEncoderDecoder`` — the drop-in boundary (netquery/model.py:57-127).

Same constructor and the same two public methods as the reference:
    QueryEncoderDecoder(graph, enc, path_dec, inter_dec)
    .forward(formula, queries, source_nodes)              -> scores[B]
    .margin_loss(formula, queries, hard_negatives=False, margin=1) -> 0-dim loss
(``loss.backward()`` / ``optimizer.step()`` keep working), plus the fused fast path the
trainer uses:
    .margin_step(items)   all (formula, slice) batches of an iteration in ONE grouped launch;
    .train_step(items, optimizer)   the same + the FusedAdam step as one library call (gqe_train_step).

What differs is where the arithmetic happens: parameters are re-homed into one flat fp32
arena in HBM and every score / gradient is produced by libgqe.so's HIP kernels
(gqe_forward / gqe_margin_fwd_bwd, include/gqe.h).  There is no torch fallback.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from .engine import ArenaLayout, Engine
from .tensorize import (FormulaPlan, pack_candidate_batches, pack_forward_batches, pack_margin_batches,
                        reference_negative_nodes)


class _MarginLossFn(torch.autograd.Function):
    """loss value now; gradients when (and scaled by what) autograd asks for them."""

    @staticmethod
    def forward(ctx, anchor, model, plan, target, neg, anchors, margin):
        ctx.model, ctx.plan, ctx.margin = model, plan, margin
        ctx.rows = (target, neg, anchors)
        n = len(target)
        descs, idx, n_scores = pack_forward_batches([(plan, np.concatenate([target, neg]),
                                                      np.concatenate([anchors, anchors], axis=1))])
        scores = model.engine.forward(descs, idx, n_scores)
        hinge = torch.clamp(margin - (scores[:n] - scores[n:]), min=0)
        return hinge.mean()

    @staticmethod
    def backward(ctx, grad_out):
        model, plan = ctx.model, ctx.plan
        target, neg, anchors = ctx.rows
        weight = float(grad_out.item())
        model._prepare_grads(plan.touched)
        descs, idx, n_scores = pack_margin_batches([(plan, target, neg, anchors, weight, ctx.margin)])
        model.engine.margin_fwd_bwd(descs, idx)
        model.engine.materialize()      # param.grad views must see the embedding-row gradients
        model._mark_touched(plan.touched)
        return (None,) * 7


class QueryEncoderDecoder(nn.Module):
    """Encoder-decoder that scores conjunctive queries (edges, metapaths, intersections)."""

    def __init__(self, graph, enc, path_dec, inter_dec, device=None, max_queries=8192, max_batches=16, rank=0, world=1,
                 lazy_adam=False):
        super(QueryEncoderDecoder, self).__init__()
        self.enc = enc
        self.path_dec = path_dec
        self.inter_dec = inter_dec
        self.graph = graph
        layout = ArenaLayout()
        dims = set()
        for name, p in self.named_parameters():
            layout.add(name, p.shape)
            dims.add(int(p.shape[-1]))
        if len(dims) != 1:
            raise Exception("the fused path needs one embedding dimension for every mode, got %s" % sorted(dims))
        self.dim = dims.pop()
        self.layout = layout
        bags = {"enc.feat-%s.weight" % m: csr for m, csr in getattr(enc, "bag_csr", {}).items()}
        self.engine = Engine(self.dim, path_dec.kind, inter_dec.kind, layout, device=device,
                             max_queries=max_queries, max_batches=max_batches, bags=bags,
                             rank=rank, world=world, lazy_adam=lazy_adam)
        # re-home every parameter into the arena (state_dict keys and values unchanged)
        for name, p in self.named_parameters():
            view = layout.view(self.engine.params, name)
            view.copy_(p.data)
            p.data = view
        # the reference's per-module entry points (enc.forward, path_dec.forward / project, inter_dec.forward on [d, B] tensors)
        # are served by the engine too (include/gqe.h: the extension points)
        enc._engine, path_dec._engine, inter_dec._engine = ("enc.", self.engine), ("path_dec.", self.engine), ("inter_dec.", self.engine)
        self._plans = {}
        self._pool_rows = {}       # id(query list) -> tensorize.PoolRows
        self._touched = set()      # tensors with a gradient since the last optimiser step
        self._dirty = set()        # tensors whose grad-arena segment may be non-zero
        self._autograd_anchor = torch.zeros((), device=self.engine.device, requires_grad=True)

    # -- helpers ----------------------------------------------------------------
    def sync(self):
        """Lazy Adam (Engine(lazy_adam=True)): settle the deferred steps so that the ``nn.Parameter`` views show the
        current values.  ``state_dict()`` does it implicitly."""
        self.engine.sync()

    def state_dict(self, *args, **kwargs):
        self.engine.sync()
        return super(QueryEncoderDecoder, self).state_dict(*args, **kwargs)

    def load_state_dict(self, state_dict, *args, **kwargs):
        """The parameters are views of the arena: rows that still owe deferred Adam steps (lazy mode) are settled
        first, otherwise those steps would later be replayed on top of the loaded values."""
        self.engine.sync()
        self.engine.params_changed()      # values arrive behind the library's back: its operand copies of the matrices are rebuilt
        return super(QueryEncoderDecoder, self).load_state_dict(state_dict, *args, **kwargs)

    def plan(self, formula):
        p = self._plans.get(formula)
        if p is None:
            p = self._plans[formula] = FormulaPlan(formula, self.layout, self.inter_dec.kind)
        return p

    def _rows(self, formula, queries, source_nodes):
        target = self.enc.rows(source_nodes, formula.target_mode)
        anchors = np.stack([self.enc.rows([q.anchor_nodes[i] for q in queries], m)
                            for i, m in enumerate(formula.anchor_modes)])
        return target, anchors

    def _param(self, key):
        mod, _, name = key.partition(".")
        obj = getattr(self, mod)
        if name.endswith(".weight") and mod == "enc":
            return getattr(obj, name[:-7]).weight
        return getattr(obj, name)

    def _prepare_grads(self, keys):
        """torch.optim compatibility: expose arena gradients as ``param.grad`` views; a
        tensor whose grad was reset to None (optimizer.zero_grad) starts from zero."""
        stale = []
        for k in keys:
            p = self._param(k)
            if p.grad is None:
                if k in self._dirty:
                    stale.append(k)
                p.grad = self.layout.view(self.engine.grads, k)
        if stale:
            self.engine.zero_grads(stale)

    def _mark_touched(self, keys):
        self._touched.update(keys)
        self._dirty.update(keys)

    # -- reference API ------------------------------------------------------------
    def forward(self, formula, queries, source_nodes):
        """scores[B] of ``source_nodes`` as the target of each query (model.py:70-109)."""
        if len(queries) != len(source_nodes):
            raise Exception("queries and source_nodes differ in length")
        target, anchors = self._rows(formula, queries, source_nodes)
        descs, idx, n = pack_forward_batches([(self.plan(formula), target, anchors)])
        self.engine.params_changed()      # (the reference's entry points: the nn.Parameter views may have been written to)
        return self.engine.forward(descs, idx, n)

    def margin_loss(self, formula, queries, hard_negatives=False, margin=1):
        """mean_b max(0, margin - (s+_b - s-_b)) with one negative per query chosen as the
        reference chooses it (model.py:112-127)."""
        neg_nodes = reference_negative_nodes(self.graph, formula, queries, hard_negatives)
        target, anchors = self._rows(formula, queries, [q.target_node for q in queries])
        neg = self.enc.rows(neg_nodes, formula.target_mode)
        self.engine.params_changed()
        return _MarginLossFn.apply(self._autograd_anchor, self, self.plan(formula), target, neg, anchors, float(margin))

    # -- fused fast path -----------------------------------------------------------
    def margin_step(self, items, want_scores=False, idx_device=None):
        """All batches of one training iteration in one grouped launch.

        items: [(formula, target_rows[n], neg_rows[n], anchor_rows[k,n], loss_weight, margin)]
        Accumulates d(sum_i w_i loss_i) into the gradient arena and returns
        (losses[len(items)+1] device tensor, pos, neg).  The fast path trusts that parameter VALUES are only written by
        the fused optimiser (and load_state_dict): after any other write call ``engine.params_changed()``."""
        packed = [(self.plan(f), t, ng, a, w, m) for (f, t, ng, a, w, m) in items]
        descs, idx, n_scores = pack_margin_batches(packed)
        out = self.engine.margin_fwd_bwd(descs, idx if idx_device is None else idx_device,
                                         n_scores=n_scores, want_scores=want_scores)
        for p in packed:
            self._mark_touched(p[0].touched)
        return out

    def train_step(self, items, optimizer, idx_device=None):
        """``margin_step(items)`` + ``optimizer.step()`` (a ``FusedAdam``) as ONE library call, gqe_train_step (include/gqe.h):
        the iteration of train_helpers.py:76-79.  Knowing the optimiser step when the forward / backward is enqueued lets the
        library run Adam over the rows the batches do not name inside the fused launch ("split step").  Returns
        losses[len(items) + 1] (device tensor).  The d x d matrices' step is enqueued by the next library call (every entry
        point settles it; ``state_dict()`` / ``sync()`` do so before values are read through the ``nn.Parameter`` views)."""
        if not isinstance(optimizer, FusedAdam) or optimizer.model is not self:
            raise Exception("train_step needs this model's FusedAdam")
        packed = [(self.plan(f), t, ng, a, w, m) for (f, t, ng, a, w, m) in items]
        descs, idx, _ = pack_margin_batches(packed)
        keys = set(self._touched)          # (gradients an earlier margin_loss / margin_step left behind are stepped too)
        for p in packed:
            keys.update(p[0].touched)
        losses = self.engine.train_step(descs, idx if idx_device is None else idx_device, keys, optimizer.lr, optimizer.betas, optimizer.eps)
        self._mark_touched(keys)
        optimizer._done()
        return losses

    def forward_candidates(self, formula, queries, candidate_nodes):
        """Scores of every node of ``candidate_nodes[i]`` as the target of ``queries[i]`` — the fused
        replacement of the reference's evaluation trick of repeating a query once per candidate
        (utils.py:58-60, 86-88): the anchors are encoded / projected / intersected once per query.
        Returns (scores flat in list order, ptr[n+1])."""
        lens = [len(c) for c in candidate_nodes]
        ptr = np.zeros(len(queries) + 1, dtype=np.int32)
        ptr[1:] = np.cumsum(lens)
        flat = [x for c in candidate_nodes for x in c]
        rows = self.enc.rows(flat, formula.target_mode)
        anchors = np.stack([self.enc.rows([q.anchor_nodes[i] for q in queries], m)
                            for i, m in enumerate(formula.anchor_modes)])
        descs, idx, n = pack_candidate_batches([(self.plan(formula), anchors, ptr, rows)])
        self.engine.params_changed()
        return self.engine.forward(descs, idx, n), ptr

    def candidate_percentiles(self, formula, queries, candidate_nodes):
        """For each query, the percentile rank (utils.py:26-33) of its FIRST candidate's score among its other
        candidates, computed on the device: fused candidate-list evaluation + gqe_rank_candidates; only one float per
        query comes back.  Returns a device tensor[len(queries)]."""
        scores, ptr = self.forward_candidates(formula, queries, candidate_nodes)
        return self.engine.rank_candidates(scores, ptr)

    POOL_ROWS_KEPT = 512      # query lists whose rows stay cached (least recently used beyond that are dropped with their lists)

    def pool_rows(self, formula, pool):
        """The table rows of ONE formula's query list, looked up once (``tensorize.PoolRows``: target rows, anchor rows, the
        negative / hard-negative lists as CSR arrays of rows): training windows (train_helpers) and every later validation
        (utils.eval_*) then work on arrays instead of Query objects.  The reference reads its lists live; the cache stands in for
        that as long as the list is not changed in place: its length and a probe of it (first / middle / last query and their
        negative lists, by identity) are checked on every use — a shuffled, re-sampled or edited list is looked up again — and
        ``invalidate_pool_rows`` drops entries by hand for changes the probe cannot see.  At most POOL_ROWS_KEPT lists are kept
        (and kept alive), least recently used first out."""
        from .tensorize import PoolRows
        cache = self._pool_rows
        rows = cache.pop(id(pool), None)
        if rows is None or rows.pool is not pool or not rows.still_valid():
            rows = PoolRows(self, formula, pool)
        cache[id(pool)] = rows                       # (most recently used last)
        while len(cache) > self.POOL_ROWS_KEPT:
            cache.pop(next(iter(cache)))
        return rows

    def invalidate_pool_rows(self, pool=None):
        """Forget the cached rows of ``pool`` (of every list if None): after an in-place change of a query list that
        keeps its length and its first / middle / last queries."""
        if pool is None:
            self._pool_rows.clear()
        else:
            self._pool_rows.pop(id(pool), None)

    def candidate_percentiles_rows(self, items):
        """``candidate_percentiles`` on row arrays, several formulas per launch: items = [(formula, anchors[k, n], ptr[n + 1],
        rows)] with the candidate rows of query i = rows[ptr[i]:ptr[i + 1]] (its target first).  Returns one device tensor of
        percentiles in item order."""
        descs, idx, n = pack_candidate_batches([(self.plan(f), a, p, r) for f, a, p, r in items])
        self.engine.params_changed()
        scores = self.engine.forward(descs, idx, n)
        ptr, off = [np.zeros(1, dtype=np.int64)], 0
        for _, _, p, r in items:
            ptr.append(np.asarray(p[1:], dtype=np.int64) + off)
            off += len(r)
        return self.engine.rank_candidates(scores, np.concatenate(ptr).astype(np.int32))

    def score_batches(self, items):
        """items: [(formula, target_rows, anchor_rows)] -> one scores tensor (concatenated)."""
        packed = [(self.plan(f), t, a) for (f, t, a) in items]
        descs, idx, n = pack_forward_batches(packed)
        self.engine.params_changed()
        return self.engine.forward(descs, idx, n)


class _FusedOptimizer(object):
    """optimizer.step() + optimizer.zero_grad() as one HIP pass over the tensors that
    received a gradient this iteration (torch semantics: others are skipped and keep their
    own step count — SURVEY.md Appendix B)."""

    def __init__(self, model, lr):
        self.model = model
        self.lr = lr

    def zero_grad(self, set_to_none=True):
        m = self.model
        if m._dirty:
            m.engine.zero_grads(sorted(m._dirty))
            m._dirty.clear()
        m._touched.clear()
        for p in m.parameters():
            p.grad = None

    def _done(self):
        m = self.model
        m._dirty -= m._touched
        m._touched.clear()
        for p in m.parameters():
            p.grad = None


class FusedAdam(_FusedOptimizer):
    """torch.optim.Adam(lr, betas=(0.9,0.999), eps=1e-8) (netquery/bio/train.py:62)."""

    def __init__(self, model, lr=0.01, betas=(0.9, 0.999), eps=1e-8):
        _FusedOptimizer.__init__(self, model, lr)
        self.betas, self.eps = betas, eps

    def step(self):
        m = self.model
        if m._touched:
            m.engine.adam_step(m._touched, self.lr, self.betas, self.eps)
        self._done()

    def state_dict(self):
        """Moments and per-tensor step counts.  ``engine.sync()`` first: a split train_step leaves the d x d matrices' update to
        the next library call — a checkpoint taken right behind train_step / a native run must hold the moments those step
        counts belong to."""
        e = self.model.engine
        e.sync()
        return {"steps": dict(e.steps), "exp_avg": e.exp_avg.clone(), "exp_avg_sq": e.exp_avg_sq.clone(),
                "lr": self.lr, "betas": self.betas, "eps": self.eps}

    def load_state_dict(self, sd):
        """... and back: the step counts go to the library too (gqe_set_adam_step_count) — run_train's native runs and
        Engine.run_train_step count steps there, and would otherwise restart the bias correction at 1 on warmed-up moments."""
        e = self.model.engine
        e.sync()
        e.steps.update(sd["steps"])
        e.exp_avg.copy_(sd["exp_avg"])
        e.exp_avg_sq.copy_(sd["exp_avg_sq"])
        e.push_step_counts()
        self.lr, self.betas, self.eps = sd["lr"], tuple(sd["betas"]), sd["eps"]


class FusedSGD(_FusedOptimizer):
    """torch.optim.SGD(lr, momentum=0) (netquery/bio/train.py:60)."""

    def step(self):
        m = self.model
        if m._touched:
            m.engine.sgd_step(m._touched, self.lr)
        self._done()
