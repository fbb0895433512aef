"""Training / evaluation harness (netquery/train_helpers.py:5-107).

``run_train`` keeps the reference's schedule: phase 1 trains 1-chain (edge) batches until
"edge convergence" (or ``max_burn_in`` iterations), phase 2 adds every other query type with
``path_weight`` (chains) / ``inter_weight`` (intersections, once with regular and once with
hard negatives); one optimiser step per iteration; EMA loss; periodic validation; the same
log lines.  The difference is mechanical: the (formula, query-slice) batches of an iteration
are collected first and run through ONE grouped fused forward/backward launch
(``QueryEncoderDecoder.margin_step``), then one fused optimiser pass — instead of up to 11
eager autograd graphs, one backward and a dense torch Adam.

With a ``torch.optim`` optimiser the reference's eager flow (``margin_loss`` ->
``loss.backward()`` -> ``optimizer.step()``) is used instead (compatibility path).
"""
from __future__ import annotations

import numpy as np
import torch

from .model import _FusedOptimizer
from .tensorize import reference_negative_nodes
from .utils import eval_auc_queries, eval_perc_queries


def check_conv(vals, window=2, tol=1e-6):
    if len(vals) < 2 * window:
        return False
    return np.mean(vals[-window:]) - np.mean(vals[-2 * window:-window]) < tol


def update_loss(loss, losses, ema_loss, ema_alpha=0.01):
    losses.append(loss)
    ema_loss = loss if ema_loss is None else (1 - ema_alpha) * ema_loss + ema_alpha * loss
    return losses, ema_loss


def run_eval(model, queries, iteration, logger, by_type=False):
    vals = {}

    def _by_rel(rel_aucs):
        for rels, auc in rel_aucs.items():
            logger.info(str(rels) + "\t" + str(auc))
    for query_type in queries["one_neg"]:
        auc, rel_aucs = eval_auc_queries(queries["one_neg"][query_type], model)
        perc = eval_perc_queries(queries["full_neg"][query_type], model)
        vals[query_type] = auc
        logger.info("{:s} val AUC: {:f} val perc {:f}; iteration: {:d}".format(query_type, auc, perc, iteration))
        if by_type:
            _by_rel(rel_aucs)
        if "inter" in query_type:
            auc, rel_aucs = eval_auc_queries(queries["one_neg"][query_type], model, hard_negatives=True)
            perc = eval_perc_queries(queries["full_neg"][query_type], model, hard_negatives=True)
            logger.info("Hard-{:s} val AUC: {:f} val perc {:f}; iteration: {:d}".format(query_type, auc, perc, iteration))
            if by_type:
                _by_rel(rel_aucs)
            vals[query_type + "hard"] = auc
    return vals


def select_batch(train_queries, iter_count, batch_size):
    """Which formula and which slice ``run_batch`` trains on (train_helpers.py:96-105):
    formula drawn ∝ its number of queries with ``np.random.multinomial``; the slice walks the
    formula's query list with wrap-around, so B varies in [1, batch_size]."""
    formulas = list(train_queries.keys())
    num = np.array([float(len(train_queries[f])) for f in formulas])
    formula = formulas[int(np.argmax(np.random.multinomial(1, num / num.sum())))]
    n = len(train_queries[formula])
    start = (iter_count * batch_size) % n
    end = min(((iter_count + 1) * batch_size) % n, n)
    end = n if end <= start else end
    return formula, start, end


def run_batch(train_queries, enc_dec, iter_count, batch_size, hard_negatives=False):
    """Eager compatibility path: returns the batch's mean margin loss (a tensor)."""
    formula, start, end = select_batch(train_queries, iter_count, batch_size)
    return enc_dec.margin_loss(formula, train_queries[formula][start:end], hard_negatives=hard_negatives)


def _collect(train_queries, model, iter_count, batch_size, weight, hard_negatives=False):
    """Fused path: the same batch as ``run_batch`` would train on, as index arrays."""
    formula, start, end = select_batch(train_queries, iter_count, batch_size)
    queries = train_queries[formula][start:end]
    neg_nodes = reference_negative_nodes(model.graph, formula, queries, hard_negatives)
    target, anchors = model._rows(formula, queries, [q.target_node for q in queries])
    return (formula, target, model.enc.rows(neg_nodes, formula.target_mode), anchors, weight, 1.0)


def run_train(model, optimizer, train_queries, val_queries, test_queries, logger,
              max_burn_in=100000, batch_size=512, log_every=100, val_every=1000, tol=1e-6,
              max_iter=int(10e7), inter_weight=0.005, path_weight=0.01, model_file=None):
    fused = isinstance(optimizer, _FusedOptimizer)
    edge_conv = False
    ema_loss = None
    vals = []
    losses = []
    conv_test = None
    i = -1
    for i in range(max_iter):
        optimizer.zero_grad()
        if fused:
            items = [_collect(train_queries["1-chain"], model, i, batch_size, 1.0)]
        else:
            loss = run_batch(train_queries["1-chain"], model, i, batch_size)
        if not edge_conv and (check_conv(vals) or len(losses) >= max_burn_in):
            logger.info("Edge converged at iteration {:d}".format(i - 1))
            logger.info("Testing at edge conv...")
            conv_test = run_eval(model, test_queries, i, logger)
            conv_test = np.mean(list(conv_test.values()))
            edge_conv = True
            losses = []
            ema_loss = None
            vals = []
            if model_file is not None:
                torch.save(model.state_dict(), model_file + "-edge_conv")

        if edge_conv:
            for query_type in train_queries:
                if query_type == "1-chain":
                    continue
                if "inter" in query_type:
                    if fused:
                        items.append(_collect(train_queries[query_type], model, i, batch_size, inter_weight))
                        items.append(_collect(train_queries[query_type], model, i, batch_size, inter_weight, True))
                    else:
                        loss += inter_weight * run_batch(train_queries[query_type], model, i, batch_size)
                        loss += inter_weight * run_batch(train_queries[query_type], model, i, batch_size, hard_negatives=True)
                else:
                    if fused:
                        items.append(_collect(train_queries[query_type], model, i, batch_size, path_weight))
                    else:
                        loss += path_weight * run_batch(train_queries[query_type], model, i, batch_size)
            if check_conv(vals):
                logger.info("Fully converged at iteration {:d}".format(i))
                break

        if fused:
            dev_losses, _, _ = model.margin_step(items)
            loss_value = float(dev_losses[-1].item())
        else:
            loss_value = loss.item()
            loss.backward()
        losses, ema_loss = update_loss(loss_value, losses, ema_loss)
        optimizer.step()

        if i % log_every == 0:
            logger.info("Iter: {:d}; ema_loss: {:f}".format(i, ema_loss))

        if i >= val_every and i % val_every == 0:
            v = run_eval(model, val_queries, i, logger)
            if edge_conv:
                vals.append(np.mean(list(v.values())))
            else:
                vals.append(v["1-chain"])

    v = run_eval(model, test_queries, i, logger)
    logger.info("Test macro-averaged val: {:f}".format(np.mean(list(v.values()))))
    if conv_test is not None:
        logger.info("Improvement from edge conv: {:f}".format((np.mean(list(v.values())) - conv_test) / conv_test))
    return v
