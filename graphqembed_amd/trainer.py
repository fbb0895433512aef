"""Tensorised trainer: the reference's training schedule (train_helpers.py:40-107) driven from int32
structure-of-arrays instead of Python ``Query`` objects.

``run_train`` keeps the reference's exact object-walking flow (and RNG call order) for drop-in use and
parity; this class is the production feed: a formula's queries live once as contiguous row arrays
(``tensorize.FormulaQueries`` built from Query lists, or ``synth.QueryPool``), a batch is a slice, negatives
are drawn with a vectorised RNG, the nine (formula, slice) batches of an iteration are packed into ONE int32
buffer that libgqe uploads on its side stream while the previous iteration computes, and the loss is only
read back when it is logged.  Sampling stays on the host cores (north star); nothing here is on the GPU.
"""
from __future__ import annotations

import time

import numpy as np

from . import parallel
from .tensorize import pack_margin_batches


class PoolView(object):
    """Uniform view of one formula's queries: target[n], anchors[k,n], negatives (fixed array or CSR)."""

    def __init__(self, pool):
        self.formula = pool.formula
        self.n = pool.n
        self.target = pool.target
        self.anchors = pool.anchors
        self.neg_fixed = getattr(pool, "neg", None)
        self.hard_fixed = getattr(pool, "hard", None)
        self.csr = pool if hasattr(pool, "sample_negatives") else None

    def negatives(self, start, end, hard, rng, all_rows):
        if all_rows is not None:                      # 1-chain: any node of the target mode (model.py:118)
            return all_rows[rng.randint(0, len(all_rows), size=end - start)]
        if self.csr is not None:
            return self.csr.sample_negatives(start, end, hard, rng)
        arr = self.hard_fixed if hard else self.neg_fixed
        if arr is None:
            raise Exception("queries of formula %s carry no %snegative samples" % (self.formula, "hard " if hard else ""))
        return arr[start:end]


class TensorizedTrainer(object):
    def __init__(self, model_or_engine, optimizer, pools_by_type, all_rows_by_mode, batch_size=512, inter_weight=0.005,
                 path_weight=0.01, seed=0, plan_of=None, dist=None, rank=0, world=1, engine=None):
        """pools_by_type: {query_type: [pool, ...]} (one pool per formula); all_rows_by_mode: {mode: int32 rows}
        to draw 1-chain negatives from.  ``model_or_engine``: a QueryEncoderDecoder (``margin_step``) — or any
        object with the same ``margin_step(items)`` method.

        Data parallel: pass ``dist`` (torch.distributed), ``rank``, ``world`` and build the model /
        Engine with the same rank and world.  Every rank draws the same formula per batch, trains on slice
        ``it*world + rank`` of it with loss weight n_rank / n_all_ranks, and the gradients are exchanged between the
        margin launch and the optimiser step (parallel.exchange_sparse, or the dense all-reduce for bag modes).
        With an Engine built with ``shard=(rank, world)`` (row-sharded tables, owner-computes Adam) the trainer drives
        the row-sharded protocol instead: ``plan_of(formula)`` must then return the FormulaPlan on the engine's (local)
        layout; the library's own Adam steps the rank's shards inside gqe_shard_step, so ``optimizer`` only supplies the
        hyper-parameters: it must be an Adam (``FusedAdam``, or a ``torch.optim.Adam`` whose ``param_groups[0]`` carries
        lr / betas / eps) — anything else is refused instead of silently trained with defaults."""
        self.model = model_or_engine
        self.opt = optimizer
        self.types = list(pools_by_type.keys())
        self.pools = {t: [PoolView(p) for p in pools_by_type[t]] for t in self.types}
        self.probs = {t: np.array([p.n for p in self.pools[t]], dtype=np.float64) for t in self.types}
        for t in self.types:
            self.probs[t] /= self.probs[t].sum()
        self.all_rows = all_rows_by_mode
        self.B = batch_size
        self.inter_weight, self.path_weight = inter_weight, path_weight
        self.rng = np.random.RandomState(seed)                 # formula draws: the same stream on every rank
        self.neg_rng = self.rng if world == 1 else np.random.RandomState([seed, 1 + rank])
        self.dist, self.rank, self.world = dist, int(rank), int(world)
        self.engine = engine if engine is not None else getattr(model_or_engine, "engine", None)
        if self.world > 1 and (self.dist is None or self.engine is None):
            raise Exception("data-parallel training needs dist= and an Engine (model.engine or engine=)")
        self._slab = 0
        self.plan_of = plan_of
        self.sharded = self.engine is not None and getattr(self.engine, "sharded", False)
        # one GPU, the library's own optimiser: step() is margin_step + opt.step() back to back and hands the losses out behind
        # both, so the matrix-gradient units may ride in the Adam pass's launch (include/gqe.h, gqe_set_deferred_gemm)
        from .model import _FusedOptimizer
        if self.engine is not None and self.world == 1 and not self.sharded and isinstance(optimizer, _FusedOptimizer):
            self.engine.set_deferred_gemm(True)
        from .model import FusedAdam
        self._one_call = (self.world == 1 and not self.sharded and isinstance(optimizer, FusedAdam) and hasattr(model_or_engine, "train_step")
                          and getattr(optimizer, "model", None) is model_or_engine and not getattr(self.engine, "lazy_adam", False))
        if self.sharded and plan_of is None:
            raise Exception("row-sharded training needs plan_of(formula) -> FormulaPlan on the engine's layout")
        if self.sharded:
            self._adam_hyper()                                 # refuse a non-Adam optimiser up front
        self.ema_loss = None
        self.iterations = 0
        self.queries_seen = 0
        self._session, self._session_open, self._posted = None, False, None   # row-sharded: transport keep-alive, the plan posted ahead

    def _batch(self, qtype, it, weight, hard=False):
        """formula drawn in proportion to its number of queries, slice by the reference's wrap-around rule."""
        plist = self.pools[qtype]
        p = plist[int(self.rng.choice(len(plist), p=self.probs[qtype]))] if len(plist) > 1 else plist[0]
        n, B = p.n, self.B
        start, end = parallel.rank_slice(n, B, it, self.rank, self.world)
        if self.world > 1:   # mean over the global batch = sum_r (n_r / n_all) * mean_r
            n_all = sum(e - s for s, e in (parallel.rank_slice(n, B, it, r, self.world) for r in range(self.world)))
            weight = weight * (end - start) / float(n_all)
        all_rows = self.all_rows[p.formula.target_mode] if qtype == "1-chain" else None
        neg = p.negatives(start, end, hard, self.neg_rng, all_rows)
        return (p.formula, p.target[start:end], neg, p.anchors[:, start:end], weight, 1.0)

    def items(self, it, edge_conv=True):
        out = [self._batch("1-chain", it, 1.0)]
        if edge_conv:
            for t in self.types:
                if t == "1-chain":
                    continue
                if "inter" in t:
                    out.append(self._batch(t, it, self.inter_weight))
                    out.append(self._batch(t, it, self.inter_weight, hard=True))
                else:
                    out.append(self._batch(t, it, self.path_weight))
        return out

    def step(self, it, edge_conv=True):
        """One iteration: sample on the host, one grouped fused launch, one fused optimiser pass."""
        items = self.items(it, edge_conv)
        if self.sharded:
            return self._sharded_step(items)
        if self.world > 1 and self.engine.sparse_exchange:
            slab = sum((2 + x[3].shape[0]) * self.B for x in items)     # the most entries any rank can produce
            if slab != self._slab:
                self.engine.exchange_reserve(slab)
                self._slab = slab
        if self._one_call:     # one GPU, the model's own FusedAdam: the iteration as ONE library call (gqe_train_step)
            losses = self.model.train_step(items, self.opt)
            self.iterations += 1
            self.queries_seen += sum(len(x[1]) for x in items)
            return losses
        losses, _, _ = self.model.margin_step(items)
        if self.world > 1:
            if self.engine.sparse_exchange:
                parallel.exchange_sparse(self.engine, self.dist)
            else:
                parallel.exchange_gradients(self.engine.grads, self.dist, engine=self.engine)
        self.opt.step()
        self.iterations += 1
        self.queries_seen += sum(len(x[1]) for x in items)
        return losses

    def _adam_hyper(self):
        """(lr, betas, eps) of the optimiser the caller handed over, read at every step (a scheduler may change lr).
        gqe_shard_step runs torch.optim.Adam semantics only: SGD, or an object that names no hyper-parameters, raises."""
        opt = self.opt
        groups = getattr(opt, "param_groups", None)
        src = groups[0] if groups else {k: getattr(opt, k) for k in ("lr", "betas", "eps") if hasattr(opt, k)}
        missing = [k for k in ("lr", "betas", "eps") if k not in src]
        if missing or "momentum" in src or type(opt).__name__.lower().endswith("sgd"):
            raise Exception("row-sharded training steps its shards with the library's Adam (gqe_shard_step): the optimiser must be "
                            "an Adam with explicit lr / betas / eps (FusedAdam or torch.optim.Adam); got %s%s"
                            % (type(opt).__name__, (" without " + ", ".join(missing)) if missing else ""))
        return float(src["lr"]), (float(src["betas"][0]), float(src["betas"][1])), float(src["eps"])

    def _shard_ps(self, items):
        packed = [(self.plan_of(f), t, ng, a, w, m) for (f, t, ng, a, w, m) in items]
        descs, idx, _ = pack_margin_batches(packed)
        ps = self.engine.prepare_shard(descs, idx, set().union(*[p[0].touched for p in packed]))
        ps["n_queries"] = sum(len(x[1]) for x in items)
        return ps

    def _sharded_step(self, items, next_items=None):
        """Row-sharded tables (include/gqe.h, gqe_shard_post / gqe_shard_step): the iteration is planned on the host (its
        index feed sorted by owner, the plan published to the other ranks through shared memory), then ONE library call
        fetches the rows, runs the fused launch on them, routes the contributions to their owners and steps the own shards
        (torch.optim.Adam semantics with the optimiser's lr / betas / eps; the step counters are the library's).
        ``next_items``: the following iteration, posted BEFORE this one runs so that no rank waits for a peer's plan."""
        eng = self.engine
        if not self._session_open:
            self._session = parallel.shard_session(eng, self.dist, self.rank, self.world)
            self._session_open = True
        if self._posted is None:
            ps = self._shard_ps(items)
            eng.shard_post(ps)
        else:
            ps = self._posted
        self._posted = None
        if next_items is not None:
            self._posted = self._shard_ps(next_items)
            eng.shard_post(self._posted)
        lr, betas, eps = self._adam_hyper()
        try:
            losses = eng.shard_step(ps, lr, betas, eps)
        except Exception:
            self._posted = None          # the session is poisoned (gqe_shard_step): whatever was posted ahead is gone with it
            self._session_open = False
            raise
        err = getattr(self._session, "error", None)
        if err is not None:
            raise err
        self.iterations += 1
        self.queries_seen += ps["n_queries"]
        return losses

    def run(self, max_iter, burn_in=0, log_every=100, logger=None):
        t0 = time.time()
        losses = None
        nxt = self.items(0, 0 >= burn_in) if self.sharded and max_iter > 0 else None
        for it in range(max_iter):
            if self.sharded:          # one iteration of look-ahead: iteration it + 1 is planned and posted before it runs
                cur, nxt = nxt, (self.items(it + 1, it + 1 >= burn_in) if it + 1 < max_iter else None)
                losses = self._sharded_step(cur, nxt)
            else:
                losses = self.step(it, edge_conv=it >= burn_in)
            if log_every and it % log_every == 0:
                val = float(losses[-1].item())        # the only host sync of the loop
                self.ema_loss = val if self.ema_loss is None else 0.99 * self.ema_loss + 0.01 * val
                if logger is not None:
                    logger.info("Iter: {:d}; loss: {:f}; {:.0f} queries/s".format(it, val, self.queries_seen / max(time.time() - t0, 1e-9)))
        return losses
