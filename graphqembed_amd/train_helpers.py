"""Training / evaluation harness with the behaviour of netquery/train_helpers.py:5-107.

What has to match the reference is the SCHEDULE, not its text:
  * phase 1 trains one 1-chain (edge) batch per iteration until the validation AUC stops improving or ``max_burn_in``
    iterations have run; phase 2 adds, per iteration and per other query type, one batch for chains (weight
    ``path_weight``) or two for intersections (regular and hard negatives, weight ``inter_weight`` each);
  * a batch = a formula drawn with ``np.random.multinomial`` in proportion to its number of queries, then the slice
    ``[it*B % n, (it+1)*B % n)`` of that formula's query list with wrap-around (so B varies in [1, batch_size]);
  * one optimiser step per iteration on the weighted sum of the batch losses; EMA loss; periodic validation; the
    reference's log lines;
  * the order in which the global ``random`` / ``np.random`` streams are consumed (formula draw and negatives of the
    1-chain batch BEFORE the edge-convergence test and its evaluation, then the other types in dictionary order), so
    that a run seeded like the reference's trains on the same batches (tests/test_gpu_api.py::
    test_run_train_reproduces_the_reference_run).

How it is organised here: ``iteration_plan`` yields the batch specs of an iteration, ``draw_batch`` turns a spec into
(formula, query slice), and an executor consumes them — ``FusedExecutor`` collects the iteration's batches as index
arrays and runs ONE grouped fused forward/backward launch (``QueryEncoderDecoder.margin_step``) followed by the fused
optimiser pass; ``EagerExecutor`` keeps the reference's eager flow (``margin_loss`` -> ``backward``) for ``torch.optim``
optimisers.  Evaluation and logging (``evaluate``) are separate from the schedule.

With a fused optimiser the iterations BETWEEN two events of the schedule (phase switch, validation, convergence stop, end) do not go
through Python at all: ``native_run_length`` says how many there are, ``_NativeLoop`` runs them as ONE library call
(``gqe_feeder_run`` with reference streams, include/gqe.h: the formula draws replayed on ``np.random``'s generator, the negatives on
``random``'s, packing, ``gqe_train_step`` / ``gqe_sgd_step``) and hands back the loss history the moving average and the log lines
are made of — the same batches, steps and generator states as the per-batch path (tests/test_gpu_api.py).  ``GQE_RUN_TRAIN_NATIVE=0``
keeps every iteration on the per-batch path.
"""
from __future__ import annotations

import os
from collections import namedtuple

import numpy as np
import torch

from .model import FusedAdam, FusedSGD, _FusedOptimizer
from .tensorize import reference_negative_nodes
from .utils import eval_auc_queries, eval_perc_queries

BatchSpec = namedtuple("BatchSpec", "query_type weight hard")


# ---- schedule -------------------------------------------------------------------------------------------------
def iteration_plan(query_types, all_types, path_weight, inter_weight):
    """Batch specs of one iteration AFTER its leading 1-chain batch: nothing while only edges are trained, else every
    other query type in the order the training dict lists them — intersections twice (regular, then hard negatives)."""
    if not all_types:
        return
    for qt in query_types:
        if qt == "1-chain":
            continue
        if "inter" in qt:
            yield BatchSpec(qt, inter_weight, False)
            yield BatchSpec(qt, inter_weight, True)
        else:
            yield BatchSpec(qt, path_weight, False)


def draw_window(queries_by_formula, iteration, batch_size):
    """(formula, its query list, lo, hi) of one batch: the formula is drawn with probability proportional to its query count
    (one ``np.random.multinomial`` call, as train_helpers.py:96-99 consumes it), [lo, hi) is the iteration's window of that
    formula's list, restarting at the list's end."""
    formulas = list(queries_by_formula)
    sizes = [float(len(queries_by_formula[f])) for f in formulas]
    pick = np.random.multinomial(1, np.array(sizes) / float(sum(sizes)))    # the same probability vector, bit for bit
    formula = formulas[int(pick.argmax())]
    pool = queries_by_formula[formula]
    lo = (iteration * batch_size) % len(pool)
    hi = ((iteration + 1) * batch_size) % len(pool)
    if hi <= lo or hi > len(pool):
        hi = len(pool)
    return formula, pool, lo, hi


def draw_batch(queries_by_formula, iteration, batch_size):
    """(formula, queries) of one batch (draw_window with the window taken out of the list)."""
    formula, pool, lo, hi = draw_window(queries_by_formula, iteration, batch_size)
    return formula, pool[lo:hi]


class Plateau(object):
    """The reference's convergence test on a series of validation scores: the mean of the last ``window`` values no
    longer exceeds the mean of the ``window`` before them by ``tol``."""

    def __init__(self, window=2, tol=1e-6):
        self.window, self.tol, self.scores = window, tol, []

    def add(self, score):
        self.scores.append(score)

    def reset(self):
        self.scores = []

    def reached(self):
        w = self.window
        if len(self.scores) < 2 * w:
            return False
        recent, before = self.scores[-w:], self.scores[-2 * w:-w]
        return float(np.mean(recent)) - float(np.mean(before)) < self.tol


class LossAverage(object):
    """Exponential moving average of the iteration loss (alpha 0.01) + how many losses the current phase has seen."""

    def __init__(self, alpha=0.01):
        self.alpha, self.value, self.count = alpha, None, 0

    def add(self, loss):
        self.count += 1
        self.value = loss if self.value is None else (1 - self.alpha) * self.value + self.alpha * loss
        return self.value

    def reset(self):
        self.value, self.count = None, 0


# ---- executors --------------------------------------------------------------------------------------------------
class FusedExecutor(object):
    """The iteration's batches become index arrays; ``finish`` runs them in one grouped launch."""

    def __init__(self, model, optimizer=None):
        """``optimizer``: a FusedAdam — ``finish`` then runs the iteration's forward / backward AND its optimiser step as one
        library call (model.train_step: gqe_train_step), and the loop's ``optimizer.step()`` finds nothing left to do."""
        self.model = model
        self.optimizer = optimizer if isinstance(optimizer, FusedAdam) else None
        self.native_optimizer = optimizer if isinstance(optimizer, (FusedAdam, FusedSGD)) else None    # (what the native loop can step with)
        self.items = []
        self._full = {}           # mode -> rows of graph.full_lists[mode] (1-chain negatives)

    def begin(self):
        self.items = []
        # the reference's negatives are drawn on the ``random`` module's generator: held natively for the iteration's batches
        # (sampler.PyRandomStream), handed back in ``finish``
        from .sampler import PyRandomStream
        self._stream = PyRandomStream().__enter__()

    def _close_stream(self):
        st, self._stream = getattr(self, "_stream", None), None
        if st is not None:
            st.__exit__()

    def add_window(self, formula, pool, lo, hi, weight, hard):
        """``add(formula, pool[lo:hi], ...)`` without one interpreter step per query: the rows of the list come from a cache
        built on first use, and the negatives are the reference's draw — ``random.choice`` per query, model.py:113-120 —
        replayed on the ``random`` module's own generator by one native call (sampler.py_random_choices): the same values, the
        same state afterwards."""
        m = self.model
        stream = getattr(self, "_stream", None)
        if stream is None:                       # (add_window outside begin / finish)
            from .sampler import py_random_choices
        else:
            py_random_choices = stream.choices
        if "inter" not in formula.query_type and hard:
            raise Exception("Hard negative examples can only be used with intersection queries")
        rows = m.pool_rows(formula, pool)
        n = hi - lo
        if formula.query_type == "1-chain" and not hard:
            mode = formula.target_mode
            if mode not in self._full:
                self._full[mode] = m.enc.rows(m.graph.full_lists[mode], mode)
            full = self._full[mode]
            neg = full[py_random_choices(np.full(n, len(full), dtype=np.int64))]
        else:
            csr = rows.lists(m, hard)
            if csr is None:       # (a query without negatives: the reference's own exception, from its own code path)
                return self.add(formula, pool[lo:hi], weight, hard)
            ptr, flat = csr
            neg = flat[ptr[lo:hi] + py_random_choices(ptr[lo + 1:hi + 1] - ptr[lo:hi])]
        self.items.append((formula, rows.target[lo:hi], neg, rows.anchors[:, lo:hi], weight, 1.0))

    def give_random(self):
        """Whatever else draws from ``random`` inside an iteration (an evaluation, this class's own ``add``) has to see the generator
        where the batches so far left it ..."""
        st = getattr(self, "_stream", None)
        if st is not None:
            st.give()

    def take_random(self):
        """... and the batches after it continue from where IT left the generator."""
        st = getattr(self, "_stream", None)
        if st is not None:
            st.take()

    def add(self, formula, queries, weight, hard):
        m = self.model
        self.give_random()
        negatives = reference_negative_nodes(m.graph, formula, queries, hard)     # the reference's draw, call for call
        self.take_random()
        target, anchors = m._rows(formula, queries, [q.target_node for q in queries])
        self.items.append((formula, target, m.enc.rows(negatives, formula.target_mode), anchors, weight, 1.0))

    def finish(self):
        self._close_stream()
        if self.optimizer is not None:
            losses = self.model.train_step(self.items, self.optimizer)
        else:
            losses, _, _ = self.model.margin_step(self.items)
        return float(losses[-1].item())

    # ---- the loop itself, natively: runs of iterations between two points where run_train has to look (include/gqe.h,
    # "reference streams") ----
    def native_loop(self, train_queries, batch_size, path_weight, inter_weight):
        """A ``_NativeLoop`` over these training queries, or None where the native feeder cannot stand in for the loop: no
        FusedAdam, a query without the negatives its type needs (the reference raises on it: that exception has to come from
        the per-batch path), more batches per iteration than one launch carries.  Built once per (dictionary, batch size)."""
        if getattr(self, "native_optimizer", None) is None or os.environ.get("GQE_RUN_TRAIN_NATIVE", "1") == "0":
            return None
        key = (id(train_queries), batch_size, path_weight, inter_weight)
        if getattr(self, "_native_key", None) != key:
            self._native_key = key
            try:
                self._native = _NativeLoop(self, train_queries, batch_size, path_weight, inter_weight)
            except _NativeUnsupported:
                self._native = None
        return self._native


class _NativeUnsupported(Exception):
    pass


class _NativeLoop(object):
    """train_helpers.py:40-107 between two events (phase switch, validation, end) as ONE library call per run of iterations:
    formula draws on ``np.random``'s generator, negatives on ``random``'s, packing, gqe_train_step — gqe_feeder_run with reference
    streams.  What Python keeps: the schedule's decisions, the moving average and the log lines (made from the run's loss
    history), validation."""

    MAX_RUN = 4096

    def __init__(self, executor, train_queries, batch_size, path_weight, inter_weight):
        m = executor.model
        pools_by_type, mode_rows = {}, {}
        from .tensorize import table_key
        # the loop draws a 1-chain batch first whatever the dictionary's order; the other types follow in its order
        order = ["1-chain"] + [qt for qt in train_queries if qt != "1-chain"]
        n_batches = 1 + sum(2 if "inter" in qt else 1 for qt in order[1:])
        if n_batches > 16 or "1-chain" not in train_queries:
            raise _NativeUnsupported()
        for qt in order:
            entries = []
            for formula, pool in train_queries[qt].items():
                if len(pool) == 0:
                    raise _NativeUnsupported()
                rows = m.pool_rows(formula, pool)
                neg = hard = None
                if qt == "1-chain":
                    mode = formula.target_mode
                    if mode not in executor._full:
                        executor._full[mode] = m.enc.rows(m.graph.full_lists[mode], mode)
                    mode_rows[table_key(mode)] = executor._full[mode]
                else:
                    neg = rows.lists(m, False)
                    if neg is None:
                        raise _NativeUnsupported()
                    if "inter" in qt:
                        hard = rows.lists(m, True)
                        if hard is None:
                            raise _NativeUnsupported()
                entries.append((m.plan(formula), rows.target, rows.anchors, neg, hard))
            if not entries:
                raise _NativeUnsupported()
            pools_by_type[qt] = entries
        self.executor, self.model = executor, m
        self.batches_full = n_batches
        m.engine.reserve(n_batches * batch_size, n_batches)       # (the per-batch path grows the workspace step by step)
        self.sgd = isinstance(executor.native_optimizer, FusedSGD)
        self.feeder = m.engine.make_reference_feeder(pools_by_type, mode_rows, batch_size, path_weight, inter_weight, sgd=self.sgd)

    def close(self):
        if self.feeder is not None:
            self.model.engine.feeder_destroy(self.feeder)
            self.feeder = None

    def run(self, first_iteration, n, all_types):
        """Iterations [first_iteration, first_iteration + n): the list of their losses (floats)."""
        import random
        from .sampler import np_state_restore, np_state_words
        m, opt = self.model, self.executor.native_optimizer
        betas, eps = (opt.betas, opt.eps) if not self.sgd else ((0.0, 0.0), 0.0)
        np_state, np_rest = np_state_words()
        version, words, gauss = random.getstate()
        py_state = np.array(words, dtype=np.uint32)
        try:
            hist = m.engine.reference_feeder_run(self.feeder, np_state, py_state, first_iteration, n, not all_types,
                                                 opt.lr, betas, eps)
            out = hist[:, self.batches_full if all_types else 1].cpu().numpy()
        finally:
            # (whatever the run consumed is consumed: the generators continue from where it left them)
            np_state_restore(np_state, np_rest)
            random.setstate((version, tuple(py_state.tolist()), gauss))
        return [float(x) for x in out]


class EagerExecutor(object):
    """torch.optim compatibility: one ``margin_loss`` per batch, one backward on the weighted sum."""

    def __init__(self, model):
        self.model = model
        self.total = None

    def begin(self):
        self.total = None

    def add(self, formula, queries, weight, hard):
        term = self.model.margin_loss(formula, queries, hard_negatives=hard)
        if weight != 1.0:
            term = weight * term
        self.total = term if self.total is None else self.total + term

    def finish(self):
        value = self.total.item()
        self.total.backward()
        return value


# ---- evaluation + logging ----------------------------------------------------------------------------------------
def evaluate(model, queries, iteration, logger, by_type=False):
    """AUC (one negative per query) and percentile (all negatives) per query type, hard-negative variants for the
    intersection types; returns {type[+"hard"]: AUC} and logs the reference's lines (train_helpers.py:28,34)."""
    scores = {}
    for query_type, one_neg in queries["one_neg"].items():
        variants = [(False, query_type, "")]
        if "inter" in query_type:
            variants.append((True, query_type + "hard", "Hard-"))
        for hard, key, prefix in variants:
            auc, per_relation = eval_auc_queries(one_neg, model, hard_negatives=hard)
            perc = eval_perc_queries(queries["full_neg"][query_type], model, hard_negatives=hard)
            scores[key] = auc
            logger.info("{:s}{:s} val AUC: {:f} val perc {:f}; iteration: {:d}".format(prefix, query_type, auc, perc, iteration))
            if by_type:
                for rels, value in per_relation.items():
                    logger.info(str(rels) + "\t" + str(value))
    return scores


def _macro(scores):
    return float(np.mean(list(scores.values())))


# ---- the loop ----------------------------------------------------------------------------------------------------
def native_run_length(first, max_iter, all_types, losses_seen, max_burn_in, val_every, max_run):
    """How many iterations from ``first`` on need nothing from ``run_train`` but their losses: up to (and including) the next
    iteration behind which validation runs (``i >= val_every and i % val_every == 0``), not into the iteration that finds
    ``max_burn_in`` losses of the edges-only phase (it switches phases), not past ``max_iter``, at most ``max_run``.  The caller has
    checked that ``first`` itself neither switches nor stops."""
    n = min(max_iter - first, max_run)
    if not all_types:
        n = min(n, max_burn_in - losses_seen)
    if val_every > 0:
        due = max(val_every, -(-first // val_every) * val_every)
        n = min(n, due - first + 1)
    return n


def run_train(model, optimizer, train_queries, val_queries, test_queries, logger,
              max_burn_in=100000, batch_size=512, log_every=100, val_every=1000, tol=1e-6,
              max_iter=int(10e7), inter_weight=0.005, path_weight=0.01, model_file=None):
    executor = FusedExecutor(model, optimizer) if isinstance(optimizer, _FusedOptimizer) else EagerExecutor(model)
    if isinstance(executor, EagerExecutor) and hasattr(model, "engine"):
        import warnings
        warnings.warn("run_train with a torch.optim optimiser keeps the reference's call sequence (margin_loss -> backward -> step: one "
                      "launch per batch, dense gradients, torch's own Adam kernels); graphqembed_amd.model.FusedAdam(model, lr=...) / "
                      "FusedSGD run the same schedule as one library call per iteration, natively between validations", stacklevel=2)

    give_random = getattr(executor, "give_random", lambda: None)
    take_random = getattr(executor, "take_random", lambda: None)

    def add(window, weight, hard):
        formula, pool, lo, hi = window
        if hasattr(executor, "add_window"):
            executor.add_window(formula, pool, lo, hi, weight, hard)
        else:
            executor.add(formula, pool[lo:hi], weight, hard)
    plateau = Plateau()        # ``tol`` is accepted but, as in the reference (its convergence test is called with the
                               # defaults, train_helpers.py:52,73), not used
    average = LossAverage()
    all_types = False          # phase 2: every query type, not only edges
    score_at_switch = None
    iteration, nxt = -1, 0
    native_for = getattr(executor, "native_loop", lambda *a: None)
    while nxt < max_iter:
        iteration = nxt
        # Runs of iterations in which nothing has to be decided here — no phase switch, no convergence stop, validation at most
        # behind the last one — go to the native loop as ONE call (``_NativeLoop``: the same draws on the same generators, the
        # same batches, the same steps); the moving average and the log lines are made from the run's loss history.
        native = native_for(train_queries, batch_size, path_weight, inter_weight)
        switching = (not all_types) and (plateau.reached() or average.count >= max_burn_in)
        stopping = all_types and plateau.reached()
        if native is not None and not switching and not stopping and not model._touched:
            n = native_run_length(nxt, max_iter, all_types, average.count, max_burn_in, val_every, native.MAX_RUN)
            optimizer.zero_grad()
            for k, loss in enumerate(native.run(nxt, n, all_types)):
                smoothed = average.add(loss)
                if (nxt + k) % log_every == 0:
                    logger.info("Iter: {:d}; ema_loss: {:f}".format(nxt + k, smoothed))
            nxt += n
            iteration = nxt - 1
            if iteration >= val_every and iteration % val_every == 0:
                scores = evaluate(model, val_queries, iteration, logger)
                plateau.add(_macro(scores) if all_types else scores["1-chain"])
            continue
        nxt += 1
        optimizer.zero_grad()
        executor.begin()
        try:
            add(draw_window(train_queries["1-chain"], iteration, batch_size), 1.0, False)
            if not all_types and (plateau.reached() or average.count >= max_burn_in):
                logger.info("Edge converged at iteration {:d}".format(iteration - 1))
                logger.info("Testing at edge conv...")
                give_random()
                score_at_switch = _macro(evaluate(model, test_queries, iteration, logger))
                take_random()
                all_types = True
                plateau.reset()
                average.reset()
                if model_file is not None:
                    torch.save(model.state_dict(), model_file + "-edge_conv")
            for spec in iteration_plan(train_queries, all_types, path_weight, inter_weight):
                add(draw_window(train_queries[spec.query_type], iteration, batch_size), spec.weight, spec.hard)
            if all_types and plateau.reached():
                logger.info("Fully converged at iteration {:d}".format(iteration))
                break
            smoothed = average.add(executor.finish())
            optimizer.step()
            if iteration % log_every == 0:
                logger.info("Iter: {:d}; ema_loss: {:f}".format(iteration, smoothed))
            if iteration >= val_every and iteration % val_every == 0:
                scores = evaluate(model, val_queries, iteration, logger)
                plateau.add(_macro(scores) if all_types else scores["1-chain"])
        finally:
            # ``begin`` moved the ``random`` module's state into a native stream; whatever ends the iteration — its ``finish``, the
            # convergence stop, the reference's exception for hard negatives on a chain query, an interrupt — hands it back
            getattr(executor, "_close_stream", lambda: None)()
    native = getattr(executor, "_native", None)
    if native is not None:      # (its pinned buffers and events go with the run)
        native.close()
        executor._native, executor._native_key = None, None
    final = evaluate(model, test_queries, iteration, logger)
    logger.info("Test macro-averaged val: {:f}".format(_macro(final)))
    if score_at_switch is not None:
        logger.info("Improvement from edge conv: {:f}".format((_macro(final) - score_at_switch) / score_at_switch))
    return final


# ---- the reference's helper names, for callers that import them ----------------------------------------------------
run_eval = evaluate


def run_batch(train_queries, enc_dec, iter_count, batch_size, hard_negatives=False):
    """One batch's mean margin loss as a tensor (eager path), on the batch the schedule would pick."""
    formula, queries = draw_batch(train_queries, iter_count, batch_size)
    return enc_dec.margin_loss(formula, queries, hard_negatives=hard_negatives)


def check_conv(vals, window=2, tol=1e-6):
    p = Plateau(window, tol)
    p.scores = list(vals)
    return p.reached()


def update_loss(loss, losses, ema_loss, ema_alpha=0.01):
    losses.append(loss)
    avg = LossAverage(ema_alpha)
    avg.value = ema_loss
    return losses, avg.add(loss)
