"""Factories and evaluation (netquery/utils.py).

  get_encoder / get_metapath_decoder / get_intersection_decoder   utils.py:93-150
  eval_auc_queries                                                utils.py:35-67
  eval_perc_queries                                               utils.py:70-91
  setup_logging                                                   utils.py:152-167

The two eval functions keep the reference's protocol (which negatives are drawn, with
which seed, in which order, how scores are grouped) but push each formula's chunk
through ``QueryEncoderDecoder.forward`` on the GPU.
"""
from __future__ import annotations

import logging
import os
import random

import numpy as np
import torch

from .decoders import (BilinearDiagMetapathDecoder, BilinearMetapathDecoder, SetIntersection,
                       SimpleSetIntersection, TransEMetapathDecoder)
from .encoders import DirectEncoder


def get_encoder(depth, graph, out_dims, feature_modules, cuda=True, node_maps=None, bags=None):
    if depth < 0 or depth > 3:
        raise Exception("Depth must be between 0 and 3 (inclusive)")
    if depth != 0:
        raise Exception("only the depth-0 DirectEncoder is on the MI355X fast path "
                        "(GraphSAGE-style encoders of netquery/encoders.py:47-129 are out of scope)")
    # (a graph from reddit_data.load_graph carries what the reference's feature closure knows: the id -> row maps and the posts' word lists)
    if node_maps is None:
        node_maps = getattr(graph, "node_maps", None)
    if bags is None:
        bags = getattr(graph, "bags", None)
    return DirectEncoder(graph.features, feature_modules, node_maps=node_maps, bags=bags)


def get_metapath_decoder(graph, out_dims, decoder):
    if decoder == "bilinear":
        return BilinearMetapathDecoder(graph.relations, out_dims)
    if decoder == "transe":
        return TransEMetapathDecoder(graph.relations, out_dims)
    if decoder == "bilinear-diag":
        return BilinearDiagMetapathDecoder(graph.relations, out_dims)
    raise Exception("Metapath decoder not recognized.")


def get_intersection_decoder(graph, out_dims, decoder):
    if decoder == "mean":
        return SetIntersection(out_dims, out_dims, agg_func=torch.mean)
    if decoder == "mean-simple":
        return SimpleSetIntersection(agg_func=torch.mean)
    if decoder == "min":
        return SetIntersection(out_dims, out_dims, agg_func=torch.min)
    if decoder == "min-simple":
        return SimpleSetIntersection(agg_func=torch.min)
    raise Exception("Intersection decoder not recognized.")


def _auc(labels, scores):
    """Area under the ROC curve = P(score_pos > score_neg) + 0.5 P(tie) (rank statistic);
    what sklearn.metrics.roc_auc_score returns for binary labels."""
    labels = np.asarray(labels)
    scores = np.asarray(scores, dtype=np.float64)
    order = np.argsort(scores, kind="mergesort")
    ranks = np.empty(len(scores), dtype=np.float64)
    s = scores[order]
    i = 0
    while i < len(s):                      # average ranks over ties
        j = i
        while j + 1 < len(s) and s[j + 1] == s[i]:
            j += 1
        ranks[order[i:j + 1]] = 0.5 * (i + j) + 1.0
        i = j + 1
    n_pos = float(labels.sum())
    n_neg = float(len(labels) - n_pos)
    if n_pos == 0 or n_neg == 0:
        raise ValueError("Only one class present in y_true. ROC AUC score is not defined in that case.")
    return (ranks[labels == 1].sum() - n_pos * (n_pos + 1) / 2.0) / (n_pos * n_neg)


def _percentile_of_score(a, score):
    """scipy.stats.percentileofscore(a, score) with the default kind='rank'."""
    a = np.asarray(a)
    n = len(a)
    if n == 0 or score != score or np.isnan(a).any():       # nan_policy 'propagate'
        return np.nan
    left = np.count_nonzero(a < score)
    right = np.count_nonzero(a <= score)
    return (left + right + (1 if right > left else 0)) * 50.0 / n


def _chunks(formula_queries, batch_size):
    for offset in range(0, len(formula_queries), batch_size):
        yield formula_queries[offset:offset + batch_size]


def eval_auc_queries(test_queries, enc_dec, batch_size=1000, hard_negatives=False, seed=0, on_device=None):
    """Overall and per-formula ROC AUC with ONE random negative per query; the negative
    draw replays the reference's ``random.seed(seed)`` / ``random.choice`` sequence.
    ``on_device`` (default: whenever the model has an ``engine``): the scores never leave the GPU — the pair counts
    behind each AUC are accumulated there (gqe_auc_pair_counts) and one number per AUC is read back."""
    if on_device is None:
        on_device = hasattr(enc_dec, "engine")
    predictions, labels, formula_aucs = [], [], {}
    pos_all, neg_all = [], []
    random.seed(seed)
    cached = on_device and hasattr(enc_dec, "pool_rows") and os.environ.get("GQE_EVAL_CACHED", "1") != "0"
    deferred = []
    for formula in test_queries:
        f_labels, f_preds, f_pos, f_neg = [], [], [], []
        if cached and len(test_queries[formula]):
            # the rows of this list are looked up once (model.pool_rows) and every later validation works on arrays: the negatives
            # are the reference's draw — random.choice per query, batch by batch — replayed natively on ``random``'s generator
            # (sampler.py_random_choices: the same values, the same state afterwards); scored below, several formulas per launch
            rows = enc_dec.pool_rows(formula, test_queries[formula])
            csr = rows.lists(enc_dec, hard_negatives)
            if csr is not None:
                from .sampler import py_random_choices
                ptr, flat = csr
                pick = py_random_choices(ptr[1:] - ptr[:-1])       # (the chunks draw in order: one call over the whole list is the same sequence)
                deferred.append((formula, rows, flat[ptr[:-1] + pick]))
                formula_aucs[formula] = None                       # (keeps the dictionary's order)
                continue
        for batch in _chunks(test_queries[formula], batch_size):
            if hard_negatives:
                negatives = [random.choice(q.hard_neg_samples) for q in batch]
            else:
                negatives = [random.choice(q.neg_samples) for q in batch]
            scores = enc_dec.forward(formula, batch + batch, [q.target_node for q in batch] + negatives)
            if on_device:
                f_pos.append(scores[:len(batch)])
                f_neg.append(scores[len(batch):])
                continue
            f_labels.extend([1] * len(batch) + [0] * len(negatives))
            f_preds.extend(scores.detach().cpu().tolist())
        if on_device:
            import torch
            if not f_pos:        # a formula with no queries: the reference's roc_auc_score raises the same ValueError (utils.py:63)
                raise ValueError("Only one class present in y_true. ROC AUC score is not defined in that case.")
            p, n = torch.cat(f_pos), torch.cat(f_neg)
            formula_aucs[formula] = enc_dec.engine.auc(p, n)
            pos_all.append(p)
            neg_all.append(n)
            continue
        formula_aucs[formula] = _auc(f_labels, np.nan_to_num(f_preds))
        labels.extend(f_labels)
        predictions.extend(f_preds)
    if on_device:
        import torch
        if deferred:
            # up to 16 formulas (<= batch_size queries each) per forward launch; every AUC's pair count lands in ONE device array
            # that is read back once
            eng = enc_dec.engine
            items, owner = [], []
            for k, (formula, rows, neg) in enumerate(deferred):
                for lo in range(0, rows.n, batch_size):
                    hi = min(lo + batch_size, rows.n)
                    items.append((formula, np.concatenate([rows.target[lo:hi], neg[lo:hi]]),
                                  np.concatenate([rows.anchors[:, lo:hi], rows.anchors[:, lo:hi]], axis=1)))
                    owner.append((k, hi - lo))
            parts = [[[], []] for _ in deferred]
            for g in range(0, len(items), 16):
                scores = enc_dec.score_batches(items[g:g + 16])
                off = 0
                for k, n in owner[g:g + 16]:
                    parts[k][0].append(scores[off:off + n])
                    parts[k][1].append(scores[off + n:off + 2 * n])
                    off += 2 * n
            counts = torch.zeros(len(deferred), dtype=torch.int64, device=eng.device)
            norms = []
            for k, (formula, rows, neg) in enumerate(deferred):
                p, n = torch.cat(parts[k][0]), torch.cat(parts[k][1])
                norms.append(eng.auc_pair_count(p, n, counts, k))
                pos_all.append(p)
                neg_all.append(n)
            for (formula, _, _), c, z in zip(deferred, counts.cpu().tolist(), norms):
                formula_aucs[formula] = float(c) / z
        if not pos_all:
            raise ValueError("Only one class present in y_true. ROC AUC score is not defined in that case.")
        return enc_dec.engine.auc(torch.cat(pos_all), torch.cat(neg_all)), formula_aucs
    return _auc(labels, np.nan_to_num(predictions)), formula_aucs


def eval_perc_queries(test_queries, enc_dec, batch_size=1000, hard_negatives=False, fused=None):
    """Mean percentile rank of the true target among ALL stored negatives of its query.
    ``fused`` (default: whenever the model offers ``candidate_percentiles``): score each query against its
    candidate list [target] + negatives in one fused evaluation launch instead of repeating the query per
    negative as the reference does (same scores, the query side is computed once) and rank on the device: one
    percentile per query is read back."""
    perc_scores = []
    if fused is None:
        fused = hasattr(enc_dec, "candidate_percentiles")
    cached = fused and hasattr(enc_dec, "pool_rows") and os.environ.get("GQE_EVAL_CACHED", "1") != "0"
    pending, on_dev = [], []      # cached lists: candidate batches of several formulas share a launch, one read-back at the end

    def flush():
        if pending:
            on_dev.append(enc_dec.candidate_percentiles_rows(pending))
            del pending[:]
    for formula in test_queries:
        if cached and len(test_queries[formula]):
            rows = enc_dec.pool_rows(formula, test_queries[formula])
            csr = rows.lists(enc_dec, hard_negatives)
            if csr is not None:
                nptr, flat = csr
                for lo in range(0, rows.n, batch_size):
                    hi = min(lo + batch_size, rows.n)
                    # candidate list of query i: its target, then its negatives (CSR over table rows)
                    ptr = (nptr[lo:hi + 1] - nptr[lo] + np.arange(hi - lo + 1)).astype(np.int32)
                    cand = np.empty(int(ptr[-1]), dtype=np.int32)
                    is_target = np.zeros(len(cand), dtype=bool)
                    is_target[ptr[:-1]] = True
                    cand[is_target] = rows.target[lo:hi]
                    cand[~is_target] = flat[nptr[lo]:nptr[hi]]
                    pending.append((formula, rows.anchors[:, lo:hi], ptr, cand))
                    if len(pending) == 16 or sum(len(p[3]) for p in pending) > (1 << 20):
                        flush()
                continue
        flush()                   # (the per-Query path below appends to perc_scores directly: keep the order)
        if on_dev:
            import torch
            perc_scores.extend(torch.cat(on_dev).detach().cpu().tolist())
            del on_dev[:]
        for batch in _chunks(test_queries[formula], batch_size):
            lists = [q.hard_neg_samples if hard_negatives else q.neg_samples for q in batch]
            lengths = [len(l) for l in lists]
            if fused:
                perc = enc_dec.candidate_percentiles(formula, batch, [[q.target_node] + list(l) for q, l in zip(batch, lists)])
                perc_scores.extend(perc.detach().cpu().tolist())
                continue
            negatives = [n for l in lists for n in l]
            rep = [q for q, k in zip(batch, lengths) for _ in range(k)]
            scores = enc_dec.forward(formula, batch + rep, [q.target_node for q in batch] + negatives)
            scores = scores.detach().cpu().numpy()
            neg_scores, cum = scores[len(batch):], 0
            for i, k in enumerate(lengths):
                perc_scores.append(_percentile_of_score(neg_scores[cum:cum + k], scores[i]))
                cum += k
    flush()
    if on_dev:
        import torch
        perc_scores.extend(torch.cat(on_dev).detach().cpu().tolist())
    return np.mean(perc_scores)


def setup_logging(log_file, console=True):
    logging.basicConfig(level=logging.INFO, format="%(asctime)s - %(levelname)s - %(message)s",
                        filename=log_file, filemode="w")
    if console:
        handler = logging.StreamHandler()
        handler.setLevel(logging.INFO)
        handler.setFormatter(logging.Formatter("%(asctime)s - %(levelname)s - %(message)s"))
        logging.getLogger("").addHandler(handler)
    return logging
