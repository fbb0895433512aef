#!/usr/bin/env python3
"""ORACLE tooling — generate golden vectors by running the REFERENCE ITSELF.

Runs only in the build container (needs /root/reference); never on the GPU box.
Nothing from the reference is copied into the repo: the script makes a scratch
copy under a temp dir, applies the purely syntactic ``lib2to3`` transform plus
the two one-line torch-API patches of SURVEY.md §8c / Appendix A, imports it,
drives ``QueryEncoderDecoder.margin_loss`` / ``forward`` / ``run_train`` /
``eval_*_queries`` on a small seeded synthetic graph and writes only DATA
(inputs and the reference's outputs) to ``tests/golden/*.npz``.

    python oracle/make_golden.py            # regenerates every fixture

Fixture schema (all index arrays are TABLE ROWS = node_maps[mode][node] + 1):
  tables_d<D>.npz                  ``enc.feat-<mode>.weight``
  model_<dec>_<inter>_d<D>.npz     ``param/<state_dict key>`` (decoder params) and,
     per case C (``1-chain`` .. ``3-chain_inter`` and ``<type>.hard``):
       C/meta (json: type, rels, hard, margin)  C/target[B] C/neg[B] C/anchors[k,B]
       C/pos[B] C/negscore[B] C/loss            C/grad/<key> (dense, touched keys)
       C/adam/neg[3,B] C/adam/loss[3] C/adam/delta/<key> (p_after_3_steps - p_0)
  train_<dec>_<inter>_d<D>.npz     5 ``run_train`` iterations (2 burn-in + 3 full):
       it<i>/n, it<i>/b<j>/{meta,target,neg,anchors,loss}, it<i>/loss,
       delta/<key> (final - initial), touched/<key> = per-tensor Adam step count
  eval_<dec>_<inter>_d<D>.npz      eval_auc_queries / eval_perc_queries captures.
  adam1_<dec>_<inter>_d32.npz      per case: grad/<key> (the reference's dense gradient) and after/<key> (the parameter after
     ONE torch.optim.Adam step on that gradient).
  trainlong_<dec>_<inter>_d{32,128}.npz  ``run_train`` for 400 iterations (100 edge-only + 300 with every type, batch 64, validation every
     100 iterations) on the d=32 world, per seed S of the run (``seed_all(S)`` right before ``run_train``):
       s<S>/log (json list: every line the reference logged: ema_loss every 20 iterations, the val AUC / val perc lines of the
       edge-convergence evaluation, of the three validations and of the final test, the macro average and the improvement),
       s<S>/loss[400] (the iteration losses handed to update_loss), s<S>/n_batches[400],
       s<S>/sig_full[400], s<S>/sig_rows[400] (CRC32 of the iteration's batches: tests/golden_utils.batch_signature),
       s<S>/touched/<key> (per-tensor Adam step counts at the end);  param/<key> (initial decoder parameters, shared).
     The evaluation queries of these runs are tests/golden/queries_long_test.pkl (serialize() tuples, up to 96 per type).
  trainlong-reddit_<dec>_<inter>_d32.npz  the same on the Reddit-shaped world (graphqembed_amd.data_utils.make_reddit_tiny; param/* holds
     every table incl. the word table, bag/post/{ptr,ids} the posts' words; queries: queries_reddit_long.pkl).
  trainlong-sgd_<dec>_<inter>_d32.npz  the same on the d = 32 world with torch.optim.SGD(lr 0.1, momentum 0) (bio/train.py:59-60).
  reddit_<dec>_<inter>_d{32,128}.npz  Reddit-shaped world (post features = nn.EmbeddingBag mean over word ids):
       param/* (all tables incl. the word table enc.feat-post.weight), bag/post/{ptr,ids} (CSR of the posts'
       word ids; a post's index row = its bag index), cases as in model_*.npz.
"""
import json
import logging
import os
import random
import shutil
import subprocess
import sys
import tempfile

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)

_2TO3_FILES = ["graph.py", "model.py", "encoders.py", "decoders.py", "utils.py",
               "train_helpers.py", "data_utils.py", "aggregators.py", "bio/data_utils.py"]


def import_reference():
    """Scratch py3 copy of the reference, put on sys.path."""
    tmp = tempfile.mkdtemp(prefix="gqe_oracle_")
    shutil.copytree(os.path.join(REFERENCE, "netquery"), os.path.join(tmp, "netquery"))
    subprocess.check_call(["chmod", "-R", "u+w", tmp])
    files = [os.path.join(tmp, "netquery", f) for f in _2TO3_FILES]
    code = ("from lib2to3.main import main; import sys; "
            "sys.exit(main('lib2to3.fixes', ['-w', '-n'] + sys.argv[1:]))")
    subprocess.check_call([sys.executable, "-W", "ignore", "-c", code] + files,
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    subprocess.check_call(["sed", "-i", "s/if type(combined) == tuple:/if isinstance(combined, tuple):/",
                           os.path.join(tmp, "netquery", "decoders.py")])       # patch #1
    subprocess.check_call(["sed", "-i", r"s/loss.data\[0\]/loss.item()/",
                           os.path.join(tmp, "netquery", "train_helpers.py")])  # patch #2
    sys.path.insert(0, tmp)
    return tmp


def seed_all(s):
    random.seed(s)
    np.random.seed(s)
    torch.manual_seed(s)


class World(object):
    """The tiny synthetic graph in the reference's own objects."""

    def __init__(self, d):
        from graphqembed_amd.data_utils import (make_synthetic_graph, make_node_maps,
                                                BIO_TINY_SIZES, BIO_TINY_EDGES_PER_KIND)
        from netquery.graph import Graph
        self.d = d
        self.relations, self.adj_lists, self.node_ids = make_synthetic_graph(
            BIO_TINY_SIZES, edges_per_kind=BIO_TINY_EDGES_PER_KIND, seed=0)
        self.node_maps = make_node_maps(self.node_ids)
        seed_all(1000 + d)
        # what bio/data_utils.py:16-21 builds
        self.feature_modules = {m: torch.nn.Embedding(len(self.node_maps[m]) + 1, d) for m in self.relations}
        for m in self.relations:
            self.feature_modules[m].weight.data.normal_(0, 1. / d)
        self._init_tables = {m: self.feature_modules[m].weight.data.clone() for m in self.relations}
        fm, nm = self.feature_modules, self.node_maps
        self.features = lambda nodes, mode: fm[mode](
            torch.autograd.Variable(torch.LongTensor([nm[mode][n] for n in nodes]) + 1))
        self.out_dims = {m: d for m in self.relations}
        self.graph = Graph(self.features, self.out_dims, self.relations, self.adj_lists)

    def rows(self, nodes, mode):
        return np.asarray([self.node_maps[mode][n] + 1 for n in nodes], dtype=np.int32)

    def reset_tables(self):
        for m in self.relations:
            self.feature_modules[m].weight.data.copy_(self._init_tables[m])
            self.feature_modules[m].weight.grad = None

    def build_model(self, dec, inter, seed=7):
        from netquery.utils import get_encoder, get_metapath_decoder, get_intersection_decoder
        from netquery.model import QueryEncoderDecoder
        self.reset_tables()
        seed_all(seed)
        enc = get_encoder(0, self.graph, self.out_dims, self.feature_modules, False)
        pdec = get_metapath_decoder(self.graph, self.out_dims, dec)
        idec = get_intersection_decoder(self.graph, self.out_dims, inter)
        return QueryEncoderDecoder(self.graph, enc, pdec, idec)


def sample_queries(world, n2=600, n3=1200):
    """Train-style queries (1 stored negative) + edge queries, grouped by formula."""
    from netquery.graph import Query
    from collections import defaultdict
    seed_all(11)
    qs = world.graph.sample_queries(2, n2, 1, verbose=False)
    qs += world.graph.sample_queries(3, n3, 1, verbose=False)
    edges = world.graph.get_all_edges(seed=3)[:600]
    qs += [Query(("1-chain", e), None, None, keep_graph=True) for e in edges]
    by = defaultdict(lambda: defaultdict(list))
    for q in qs:
        by[q.formula.query_type][q.formula].append(q)
    return by


def sample_test_queries(world, per_type=24):
    """Eval-style queries: a ``one_neg`` split and a ``full_neg`` split (many negatives)."""
    from netquery.graph import Query
    from collections import defaultdict
    out = {"one_neg": defaultdict(lambda: defaultdict(list)), "full_neg": defaultdict(lambda: defaultdict(list))}
    seed_all(21)
    for split, neg_max in (("one_neg", 1), ("full_neg", 8)):
        qs = world.graph.sample_queries(2, 6 * per_type, neg_max, verbose=False)
        qs += world.graph.sample_queries(3, 12 * per_type, neg_max, verbose=False)
        edges = world.graph.get_all_edges(seed=5)[:per_type]
        qs += [Query(("1-chain", e), world.graph.get_negative_edge_samples(e, neg_max), None, neg_max + 1, keep_graph=True) for e in edges]
        counts = defaultdict(int)
        for q in qs:
            t = q.formula.query_type
            if counts[t] >= per_type:
                continue
            if "inter" in t and (q.hard_neg_samples is None or len(q.hard_neg_samples) == 0):
                continue
            counts[t] += 1
            out[split][t][q.formula].append(q)
    return out


def rels_to_json(rels):
    return [rels_to_json(r) if isinstance(r[0], tuple) else list(r) for r in rels]


def anchors_rows(world, formula, queries):
    return np.stack([world.rows([q.anchor_nodes[i] for q in queries], formula.anchor_modes[i])
                     for i in range(len(formula.anchor_modes))])


class Spy(object):
    """Records what margin_loss / forward were called with and what they returned."""

    def __init__(self, model):
        self.model = model
        self.calls = []
        self.in_margin = False
        self.margin_calls = []
        self._fwd = model.forward
        self._ml = model.margin_loss
        model.forward = self.forward
        model.margin_loss = self.margin_loss

    def forward(self, formula, queries, source_nodes):
        out = self._fwd(formula, queries, source_nodes)
        self.calls.append({"formula": formula, "queries": list(queries), "nodes": list(source_nodes),
                           "scores": out.detach().numpy().copy(), "in_margin": self.in_margin})
        return out

    def margin_loss(self, formula, queries, hard_negatives=False, margin=1):
        self.in_margin = True
        n0 = len(self.calls)
        loss = self._ml(formula, queries, hard_negatives=hard_negatives, margin=margin)
        self.in_margin = False
        pos, neg = self.calls[n0], self.calls[n0 + 1]
        self.margin_calls.append({"formula": formula, "queries": list(queries), "hard": hard_negatives,
                                  "margin": margin, "neg_nodes": neg["nodes"], "pos": pos["scores"],
                                  "neg": neg["scores"], "loss": float(loss.item())})
        return loss


def batch_record(world, mc, prefix, out):
    f, qs = mc["formula"], mc["queries"]
    out[prefix + "/meta"] = json.dumps({"type": f.query_type, "rels": rels_to_json(f.rels),
                                        "hard": bool(mc["hard"]), "margin": mc["margin"]})
    out[prefix + "/target"] = world.rows([q.target_node for q in qs], f.target_mode)
    out[prefix + "/neg"] = world.rows(mc["neg_nodes"], f.target_mode)
    out[prefix + "/anchors"] = anchors_rows(world, f, qs)
    out[prefix + "/loss"] = np.float64(mc["loss"])


def state_np(model):
    return {k: v.detach().numpy().copy() for k, v in model.state_dict().items()}


def pick_formula(by_formula, qtype, B):
    cands = sorted(by_formula[qtype].items(), key=lambda kv: (-len(kv[1]), str(kv[0])))
    f, qs = cands[0]
    return f, qs[:B]


def gen_model_cases(world, by_formula, dec, inter, B, cases=None, do_adam=True):
    d = world.d
    out = {}
    model = world.build_model(dec, inter)
    p0 = state_np(model)
    for k, v in p0.items():
        if not k.startswith("enc."):
            out["param/" + k] = v
    types = ["1-chain", "2-chain", "3-chain", "2-inter", "3-inter", "3-inter_chain", "3-chain_inter"]
    for qtype in types:
        for hard in ((False, True) if "inter" in qtype else (False,)):
            case = qtype + (".hard" if hard else "")
            if cases is not None and case not in cases:
                continue
            model = world.build_model(dec, inter)
            spy = Spy(model)
            formula, queries = pick_formula(by_formula, qtype, B)
            seed_all(31)
            model.zero_grad()
            loss = model.margin_loss(formula, queries, hard_negatives=hard)
            loss.backward()
            mc = spy.margin_calls[-1]
            batch_record(world, mc, case, out)
            out[case + "/pos"] = mc["pos"]
            out[case + "/negscore"] = mc["neg"]
            for k, p in model.named_parameters():
                if p.grad is not None:
                    out[case + "/grad/" + k] = p.grad.detach().numpy().copy()
            if not do_adam:
                continue
            # three Adam steps on the same query batch (negatives re-drawn by the reference)
            model = world.build_model(dec, inter)
            spy = Spy(model)
            opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=0.01)
            seed_all(37)
            negs, losses = [], []
            for _ in range(3):
                opt.zero_grad()
                loss = model.margin_loss(formula, queries, hard_negatives=hard)
                loss.backward()
                opt.step()
                negs.append(world.rows(spy.margin_calls[-1]["neg_nodes"], formula.target_mode))
                losses.append(spy.margin_calls[-1]["loss"])
            out[case + "/adam/neg"] = np.stack(negs)
            out[case + "/adam/loss"] = np.asarray(losses, dtype=np.float64)
            p3 = state_np(model)
            for k, p in model.named_parameters():
                if p.grad is not None:
                    out[case + "/adam/delta/" + k] = (p3[k].astype(np.float64) - p0[k].astype(np.float64)).astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "model_%s_%s_d%d.npz" % (dec, inter, d)), **out)
    return p0


class ListLogger(object):
    def __init__(self):
        self.lines = []

    def info(self, msg):
        self.lines.append(msg)


def gen_train_case(world, by_formula, test_queries, dec, inter, B):
    """run_train for 5 iterations: 2 burn-in (1-chain only) then 3 with every type."""
    import netquery.train_helpers as th
    d = world.d
    model = world.build_model(dec, inter)
    p0 = state_np(model)
    spy = Spy(model)
    opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=0.01)
    iter_losses = []
    orig_update = th.update_loss

    def update_spy(loss, losses, ema_loss, ema_alpha=0.01):
        iter_losses.append((float(loss), len(spy.margin_calls)))
        return orig_update(loss, losses, ema_loss, ema_alpha)
    th.update_loss = update_spy
    train_queries = {t: dict(by_formula[t]) for t in
                     ["1-chain", "2-chain", "3-chain", "2-inter", "3-inter", "3-inter_chain", "3-chain_inter"]}
    logger = ListLogger()
    seed_all(41)
    try:
        th.run_train(model, opt, train_queries, test_queries, test_queries, logger,
                     max_burn_in=2, batch_size=B, log_every=1, val_every=1000, max_iter=5)
    finally:
        th.update_loss = orig_update
    out = {}
    start = 0
    for i, (loss, end) in enumerate(iter_losses):
        calls = spy.margin_calls[start:end]
        out["it%d/n" % i] = np.int32(len(calls))
        out["it%d/loss" % i] = np.float64(loss)
        for j, mc in enumerate(calls):
            batch_record(world, mc, "it%d/b%d" % (i, j), out)
        start = end
    p1 = state_np(model)
    for k in p1:
        out["delta/" + k] = (p1[k].astype(np.float64) - p0[k].astype(np.float64)).astype(np.float32)
    names = {id(p): k for k, p in model.named_parameters()}
    for p, st in opt.state.items():
        out["touched/" + names[id(p)]] = np.int32(int(st["step"]))
    out["log"] = json.dumps(logger.lines)
    for k, v in p0.items():
        if not k.startswith("enc."):
            out["param/" + k] = v
    np.savez_compressed(os.path.join(OUT, "train_%s_%s_d%d.npz" % (dec, inter, d)), **out)


def gen_eval_case(world, test_queries, dec, inter):
    """eval_auc_queries / eval_perc_queries (utils.py:35-91) on a fresh model."""
    from netquery.utils import eval_auc_queries, eval_perc_queries
    d = world.d
    model = world.build_model(dec, inter)
    p0 = state_np(model)
    out = {}
    for k, v in p0.items():
        if not k.startswith("enc."):
            out["param/" + k] = v
    spy = Spy(model)
    n = 0
    summary = {}
    for qtype in sorted(test_queries["one_neg"].keys()):
        for hard in ((False, True) if "inter" in qtype else (False,)):
            tag = qtype + (".hard" if hard else "")
            c0 = len(spy.calls)
            auc, rel_aucs = eval_auc_queries(test_queries["one_neg"][qtype], model, hard_negatives=hard)
            c1 = len(spy.calls)
            perc = eval_perc_queries(test_queries["full_neg"][qtype], model, hard_negatives=hard)
            c2 = len(spy.calls)
            summary[tag] = {"auc": float(auc), "perc": float(perc),
                            "auc_calls": list(range(n, n + c1 - c0)),
                            "perc_calls": list(range(n + c1 - c0, n + c2 - c0)),
                            "formula_aucs": [[str(f), float(a)] for f, a in rel_aucs.items()]}
            for call in spy.calls[c0:c2]:
                f, qs = call["formula"], call["queries"]
                pre = "call%d" % n
                out[pre + "/meta"] = json.dumps({"type": f.query_type, "rels": rels_to_json(f.rels)})
                out[pre + "/target"] = world.rows(call["nodes"], f.target_mode)
                out[pre + "/anchors"] = anchors_rows(world, f, qs)
                out[pre + "/scores"] = call["scores"]
                n += 1
    out["summary"] = json.dumps(summary)
    np.savez_compressed(os.path.join(OUT, "eval_%s_%s_d%d.npz" % (dec, inter, d)), **out)


def dump_queries(world, by_formula, test_queries):
    """The sampled Query objects as data (serialize() tuples) so host-side tests can
    rebuild the same query sets without the reference."""
    import pickle
    train = {t: [q.serialize() if q.query_graph is not None else None for f in by_formula[t] for q in by_formula[t][f]]
             for t in by_formula}
    test = {s: [q.serialize() for t in test_queries[s] for f in test_queries[s][t] for q in test_queries[s][t][f]]
            for s in test_queries}
    with open(os.path.join(OUT, "queries_tiny.pkl"), "wb") as f:
        pickle.dump({"train": train, "test": test}, f, protocol=2)


class RedditWorld(object):
    """Tiny Reddit-shaped graph: users / communities are nn.Embedding tables indexed by id + 1, posts are an
    nn.EmbeddingBag (mean) over each post's word ids — what reddit/data_utils_new.py:143-182 (load_graph)
    builds.  That module cannot be imported here (it imports gensim / spacy at module level), so its
    feature closure (lines 162-169, CPU branch) is restated below; everything downstream (Graph,
    DirectEncoder, decoders, QueryEncoderDecoder, optim) is the imported reference."""

    def __init__(self, d, n_user=80, n_post=120, n_comm=12, n_words=60, seed=0):
        from netquery.graph import Graph
        from graphqembed_amd.data_utils import make_reddit_tiny
        self.d = d
        # (the graph itself lives in the package, so that tests can rebuild it without the reference: same draws, same order)
        self.relations, self.adj_lists, self.post_words_np = make_reddit_tiny(n_user, n_post, n_comm, n_words, seed)
        self.RELATIONS = self.relations
        self.post_ids = list(range(n_post))
        post_words = {p: torch.LongTensor(w) for p, w in self.post_words_np.items()}
        seed_all(2000 + d)
        self.feature_modules = {"post": torch.nn.EmbeddingBag(n_words, d), "user": torch.nn.Embedding(n_user + 1, d),
                                "community": torch.nn.Embedding(n_comm + 1, d)}
        for m in self.feature_modules:
            self.feature_modules[m].weight.data.normal_(0, 1. / d)
        self._init_tables = {m: self.feature_modules[m].weight.data.clone() for m in self.feature_modules}
        fm = self.feature_modules

        def _feature_func(nodes, mode):                    # restated: reddit/data_utils_new.py:162-169
            if mode != "post":
                return fm[mode](torch.autograd.Variable(torch.LongTensor(nodes) + 1))
            offsets = np.concatenate(([0], np.cumsum([post_words[post].size()[0] for post in nodes[:-1]])))
            return fm[mode](torch.autograd.Variable(torch.cat([post_words[post] for post in nodes])),
                            torch.autograd.Variable(torch.LongTensor(offsets)))
        self.features = _feature_func
        self.out_dims = {m: d for m in self.relations}
        self.graph = Graph(self.features, self.out_dims, self.relations, self.adj_lists)
        self.bag_ptr = np.concatenate(([0], np.cumsum([len(self.post_words_np[p]) for p in self.post_ids]))).astype(np.int32)
        self.bag_ids = np.concatenate([self.post_words_np[p] for p in self.post_ids]).astype(np.int32)

    def rows(self, nodes, mode):
        """posts -> bag index (position in post_ids); users / communities -> id + 1."""
        if mode == "post":
            return np.asarray(nodes, dtype=np.int32)
        return np.asarray(nodes, dtype=np.int32) + 1

    reset_tables = World.reset_tables
    build_model = World.build_model


def gen_reddit_cases(d, B, combos=(("bilinear-diag", "min"), ("bilinear", "mean"), ("transe", "mean-simple")),
                     hard_types=("2-inter", "3-chain_inter"), do_adam=True):
    """margin_loss / backward / 3 Adam steps on the Reddit-shaped world (EmbeddingBag post features)."""
    from collections import defaultdict
    from netquery.graph import Query
    world = RedditWorld(d)
    seed_all(51)
    qs = world.graph.sample_queries(2, 500, 1, verbose=False) + world.graph.sample_queries(3, 1000, 1, verbose=False)
    qs += [Query(("1-chain", e), None, None, keep_graph=True) for e in world.graph.get_all_edges(seed=7)[:500]]
    by = defaultdict(lambda: defaultdict(list))
    for q in qs:
        by[q.formula.query_type][q.formula].append(q)
    for dec, inter in combos:
        out = {"bag/post/ptr": world.bag_ptr, "bag/post/ids": world.bag_ids}
        model = world.build_model(dec, inter)
        p0 = state_np(model)
        for k, v in p0.items():
            out["param/" + k] = v
        for qtype in ["1-chain", "2-chain", "3-chain", "2-inter", "3-inter", "3-inter_chain", "3-chain_inter"]:
            # prefer formulas that involve posts on both sides, then the best populated
            cands = sorted(by[qtype].items(), key=lambda kv: (-(str(kv[0]).count("post")), -len(kv[1]), str(kv[0])))
            formula, queries = cands[0][0], cands[0][1][:B]
            for hard in ((False, True) if qtype in hard_types else (False,)):
                case = qtype + (".hard" if hard else "")
                model = world.build_model(dec, inter)
                spy = Spy(model)
                seed_all(61)
                model.zero_grad()
                loss = model.margin_loss(formula, queries, hard_negatives=hard)
                loss.backward()
                mc = spy.margin_calls[-1]
                batch_record(world, mc, case, out)
                out[case + "/pos"] = mc["pos"]
                out[case + "/negscore"] = mc["neg"]
                for k, p in model.named_parameters():
                    if p.grad is not None:
                        out[case + "/grad/" + k] = p.grad.detach().numpy().copy()
                if not do_adam:
                    continue
                model = world.build_model(dec, inter)
                spy = Spy(model)
                opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=0.01)
                seed_all(67)
                negs, losses = [], []
                for _ in range(3):
                    opt.zero_grad()
                    loss = model.margin_loss(formula, queries, hard_negatives=hard)
                    loss.backward()
                    opt.step()
                    negs.append(world.rows(spy.margin_calls[-1]["neg_nodes"], formula.target_mode))
                    losses.append(spy.margin_calls[-1]["loss"])
                out[case + "/adam/neg"] = np.stack(negs)
                out[case + "/adam/loss"] = np.asarray(losses, dtype=np.float64)
                p3 = state_np(model)
                for k, p in model.named_parameters():
                    if p.grad is not None:
                        out[case + "/adam/delta/" + k] = (p3[k].astype(np.float64) - p0[k].astype(np.float64)).astype(np.float32)
        np.savez_compressed(os.path.join(OUT, "reddit_%s_%s_d%d.npz" % (dec, inter, d)), **out)
        print("reddit", dec, inter, d, flush=True)


def gen_adam1_case(world, by_formula, dec, inter, B, cases=("2-chain", "3-inter.hard", "3-chain_inter")):
    """One torch.optim.Adam step (lr 0.01, torch defaults) from the reference's own gradient: per case the dense gradient
    of every touched tensor and the parameters AFTER the step -> adam1_<dec>_<inter>_d<D>.npz.  Lets the device optimiser
    be checked in isolation: golden gradient in -> one step -> golden parameters out."""
    out = {}
    for case in cases:
        qtype, hard = (case[:-5], True) if case.endswith(".hard") else (case, False)
        model = world.build_model(dec, inter)
        formula, queries = pick_formula(by_formula, qtype, B)
        opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=0.01)
        seed_all(31)
        opt.zero_grad()
        loss = model.margin_loss(formula, queries, hard_negatives=hard)
        loss.backward()
        for k, p in model.named_parameters():
            if p.grad is not None:
                out[case + "/grad/" + k] = p.grad.detach().numpy().copy()
        opt.step()
        after = state_np(model)
        for k, p in model.named_parameters():
            if p.grad is not None:
                out[case + "/after/" + k] = after[k]
    np.savez_compressed(os.path.join(OUT, "adam1_%s_%s_d%d.npz" % (dec, inter, world.d)), **out)


def gen_trainlong_case(world, by_formula, test_queries, dec, inter, seeds, B=64, max_burn_in=100, max_iter=400, tag=None, sgd_lr=None):
    """The reference's run_train (train_helpers.py:40-93) over both phases with validations on the way, once per seed: what it
    logged, every iteration's loss and a checksum of every iteration's batches -> trainlong_<dec>_<inter>_d32.npz."""
    import netquery.train_helpers as th
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from golden_utils import batch_signature
    out = {"meta": json.dumps({"batch_size": B, "max_burn_in": max_burn_in, "max_iter": max_iter, "log_every": 20, "val_every": 100,
                               "lr": 0.01 if sgd_lr is None else sgd_lr, "optimizer": "adam" if sgd_lr is None else "sgd", "seeds": list(seeds)})}
    for seed in seeds:
        model = world.build_model(dec, inter)
        if seed == seeds[0]:
            for k, v in state_np(model).items():
                if tag is not None or not k.startswith("enc."):      # (a tagged world ships its own tables: no tables_d<D>.npz)
                    out["param/" + k] = v
        spy = Spy(model)
        if sgd_lr is None:
            opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=0.01)
        else:                                                   # bio/train.py:59-60: --opt sgd
            opt = torch.optim.SGD([p for p in model.parameters() if p.requires_grad], lr=sgd_lr, momentum=0)
        iter_losses = []
        orig_update = th.update_loss

        def update_spy(loss, losses, ema_loss, ema_alpha=0.01):
            iter_losses.append((float(loss), len(spy.margin_calls)))
            return orig_update(loss, losses, ema_loss, ema_alpha)
        th.update_loss = update_spy
        train_queries = {t: dict(by_formula[t]) for t in
                         ["1-chain", "2-chain", "3-chain", "2-inter", "3-inter", "3-inter_chain", "3-chain_inter"]}
        logger = ListLogger()
        seed_all(seed)
        try:
            th.run_train(model, opt, train_queries, test_queries, test_queries, logger,
                         max_burn_in=max_burn_in, batch_size=B, log_every=20, val_every=100, max_iter=max_iter)
        finally:
            th.update_loss = orig_update
        assert len(iter_losses) == max_iter
        start, sig_full, sig_rows, nb = 0, [], [], []
        for loss, end in iter_losses:
            full = rows = 0
            for mc in spy.margin_calls[start:end]:
                f, qs = mc["formula"], mc["queries"]
                t = world.rows([q.target_node for q in qs], f.target_mode)
                ng = world.rows(mc["neg_nodes"], f.target_mode)
                a = anchors_rows(world, f, qs)
                full = batch_signature(full, f.query_type, rels_to_json(f.rels), t, ng, a)
                rows = batch_signature(rows, f.query_type, None, t, ng, a)
            sig_full.append(full); sig_rows.append(rows); nb.append(end - start)
            start = end
        pre = "s%d/" % seed
        out[pre + "log"] = json.dumps(logger.lines)
        out[pre + "loss"] = np.asarray([l for l, _ in iter_losses], dtype=np.float64)
        out[pre + "n_batches"] = np.asarray(nb, dtype=np.int32)
        out[pre + "sig_full"] = np.asarray(sig_full, dtype=np.uint32)
        out[pre + "sig_rows"] = np.asarray(sig_rows, dtype=np.uint32)
        names = {id(p): k for k, p in model.named_parameters()}
        for p, st in opt.state.items():
            if "step" in st:                                    # (SGD without momentum keeps no state)
                out[pre + "touched/" + names[id(p)]] = np.int32(int(st["step"]))
        print("trainlong", dec, inter, "seed", seed, logger.lines[-2], flush=True)
    if tag is not None and hasattr(world, "bag_ptr"):
        out["bag/post/ptr"], out["bag/post/ids"] = world.bag_ptr, world.bag_ids
    name = "trainlong_%s_%s_d%d.npz" % (dec, inter, world.d) if tag is None else "trainlong-%s_%s_%s_d%d.npz" % (tag, dec, inter, world.d)
    np.savez_compressed(os.path.join(OUT, name), **out)


def sample_long_test_queries(world, by_formula, per_type=96):
    """The evaluation sets of the long runs.  The tiny graph is uniformly random, so queries the model never trained on score at
    chance whatever the trainer does; an evaluation that can tell a working trainer from a broken one has to look at what training
    fits.  ``one_neg`` (AUC) therefore takes the first ``per_type`` TRAINING queries of every type (their one stored negative; the
    1-chain queries are the first training edges with one sampled negative each), ``full_neg`` (percentile) keeps freshly sampled
    queries with up to 8 negatives as in sample_test_queries."""
    from netquery.graph import Query
    from collections import defaultdict
    out = sample_test_queries(world, per_type=per_type)
    one = defaultdict(lambda: defaultdict(list))
    seed_all(23)
    for t in by_formula:
        n = 0
        for f, qs in by_formula[t].items():
            for q in qs:
                if n >= per_type:
                    break
                if t == "1-chain":
                    e = q.query_graph[1]
                    q = Query(("1-chain", e), world.graph.get_negative_edge_samples(e, 1), None, 2, keep_graph=True)
                elif "inter" in t and not q.hard_neg_samples:
                    continue
                one[t][f].append(q)
                n += 1
    out["one_neg"] = one
    return out


def gen_round6():
    """Fixtures added in round 6: trainlong_<dec>_<inter>_d32.npz (400 iterations of the reference's run_train; three seeds for the
    headline decoder pair — their spread is the band the device run is held to — one for the two other families) and the larger
    evaluation query set those runs validate on (queries_long_test.pkl)."""
    import pickle
    world = World(32)
    by_formula = sample_queries(world)          # the training queries of queries_tiny.pkl (same seeds)
    test_queries = sample_long_test_queries(world, by_formula, per_type=96)
    test = {s: [q.serialize() for t in test_queries[s] for f in test_queries[s][t] for q in test_queries[s][t][f]] for s in test_queries}
    with open(os.path.join(OUT, "queries_long_test.pkl"), "wb") as f:
        pickle.dump({"test": test}, f, protocol=2)
    gen_trainlong_case(world, by_formula, test_queries, "bilinear-diag", "min", (41, 42, 43))
    gen_trainlong_case(world, by_formula, test_queries, "bilinear", "mean", (41,))
    gen_trainlong_case(world, by_formula, test_queries, "transe", "min-simple", (41,))
    # ... and the headline decoder pair at d = 128 — the dimension whose kernels carry the split step's riders (the timed path):
    # the same graph, the same query sets (they do not depend on d), tables_d128.npz
    world128 = World(128)
    by128 = sample_queries(world128)
    test128 = sample_long_test_queries(world128, by128, per_type=96)
    same = {s: [q.serialize() for t in test128[s] for f in test128[s][t] for q in test128[s][t][f]] for s in test128}
    assert same == test, "the evaluation queries of the d = 128 run differ from queries_long_test.pkl"
    gen_trainlong_case(world128, by128, test128, "bilinear-diag", "min", (41,))
    gen_trainlong_case(world128, by128, test128, "bilinear", "mean", (41,))       # BASELINE config 4's decoder pair (MFMA hops)
    # ... and a Reddit-shaped world (posts = nn.EmbeddingBag means over word rows): the bag path inside the loop.  Its graph is
    # graphqembed_amd.data_utils.make_reddit_tiny; its query sets travel as serialize() tuples (queries_reddit_long.pkl)
    rw = RedditWorld(32)
    rby = sample_queries(rw, n2=500, n3=1000)
    rtest = sample_long_test_queries(rw, rby, per_type=64)
    rtrain = {t: [q.serialize() for f in rby[t] for q in rby[t][f]] for t in rby}
    rtest_ser = {s: [q.serialize() for t in rtest[s] for f in rtest[s][t] for q in rtest[s][t][f]] for s in rtest}
    with open(os.path.join(OUT, "queries_reddit_long.pkl"), "wb") as f:
        pickle.dump({"train": rtrain, "test": rtest_ser}, f, protocol=2)
    gen_trainlong_case(rw, rby, rtest, "bilinear-diag", "min", (41,), tag="reddit")
    # ... and --opt sgd (bio/train.py:59-60: torch.optim.SGD, momentum 0) on the d = 32 world
    gen_trainlong_case(world, by_formula, test_queries, "bilinear-diag", "min", (41,), tag="sgd", sgd_lr=0.1)


def gen_round2():
    """Fixtures added in round 2 (each builds its own worlds, so the files above are unaffected):
      eval_bilinear-diag_min_d128.npz   eval_auc_queries / eval_perc_queries at the BASELINE dimension
      adam1_<dec>_<inter>_d32.npz       golden gradient -> one Adam step -> golden parameters
      reddit_<dec>_<inter>_d128.npz     Reddit-shaped world at d=128, hard negatives for every intersection type"""
    world = World(128)
    sample_queries(world)
    gen_eval_case(world, sample_test_queries(world), "bilinear-diag", "min")
    print("eval bilinear-diag min 128", flush=True)
    world = World(32)
    by_formula = sample_queries(world)
    for dec, inter in (("bilinear-diag", "min"), ("bilinear", "mean"), ("transe", "min-simple")):
        gen_adam1_case(world, by_formula, dec, inter, 23)
        print("adam1", dec, inter, flush=True)
    gen_reddit_cases(128, 40, combos=(("bilinear-diag", "min"),),
                     hard_types=("2-inter", "3-inter", "3-inter_chain", "3-chain_inter"), do_adam=False)


def main():
    os.makedirs(OUT, exist_ok=True)
    tmp = import_reference()
    logging.disable(logging.CRITICAL)
    if "--round6-only" in sys.argv:          # the round-6 fixtures alone (bit-identical to what the full run writes)
        try:
            gen_round6()
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
        return
    if "--round2-only" in sys.argv:          # the round-2 fixtures alone (bit-identical to what the full run writes)
        try:
            gen_round2()
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
        return
    try:
        decs = ["bilinear-diag", "transe", "bilinear"]
        inters = ["min", "mean", "min-simple", "mean-simple"]
        meta = {"torch": torch.__version__, "numpy": np.__version__, "reference": "williamleif/graphqembed @ /root/reference"}
        all32 = [(a, b, None, True) for a in decs for b in inters]
        some128 = [("bilinear-diag", "min", ["1-chain", "2-chain", "3-chain", "2-inter", "3-inter", "3-inter_chain",
                                             "3-chain_inter", "3-inter.hard"], False),
                   ("bilinear", "mean", ["3-chain", "3-inter_chain", "3-chain_inter.hard"], False),
                   ("transe", "min-simple", ["2-chain", "3-inter", "3-chain_inter"], False)]
        for d, combos, B in ((32, all32, 23), (128, some128, 40)):
            world = World(d)
            by_formula = sample_queries(world)
            test_queries = sample_test_queries(world)
            if d == 32:
                dump_queries(world, by_formula, test_queries)
            tables = None
            for dec, inter, cases, do_adam in combos:
                p0 = gen_model_cases(world, by_formula, dec, inter, B, cases, do_adam)
                tables = {k: v for k, v in p0.items() if k.startswith("enc.")}
                print("model", dec, inter, d, flush=True)
            np.savez_compressed(os.path.join(OUT, "tables_d%d.npz" % d), **tables)
            for dec, inter in ([("bilinear-diag", "min"), ("bilinear", "mean"), ("transe", "min-simple")] if d == 32
                               else [("bilinear-diag", "min")]):
                gen_train_case(world, by_formula, test_queries, dec, inter, B)
                if d == 32:
                    gen_eval_case(world, test_queries, dec, inter)
                print("train/eval", dec, inter, d, flush=True)
        gen_reddit_cases(32, 23)
        gen_round2()
        gen_round6()
        with open(os.path.join(OUT, "META.json"), "w") as f:
            json.dump(meta, f, indent=1)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
