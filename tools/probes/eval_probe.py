#!/usr/bin/env python3
"""What run_train's validation costs through the reference-shaped API: train_helpers.evaluate (eval_auc_queries + eval_perc_queries per
query type, hard variants for intersections) on a bio-synth-sized world, query lists sampled by the native sampler.
python tools/probes/eval_probe.py [queries per type] [negatives per query]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from graphqembed_amd import data_utils, train_helpers, utils
from graphqembed_amd.graph import Graph, Query
from graphqembed_amd.model import QueryEncoderDecoder
from graphqembed_amd.sampler import NativeSampler
n_q = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
n_neg = int(sys.argv[2]) if len(sys.argv) > 2 else 100
d = 128
rel, adj, ids = data_utils.make_synthetic_graph(data_utils.BIO_SYNTH_SIZES, seed=0)
node_maps = data_utils.make_node_maps(ids)
dims = {m: d for m in rel}
graph = Graph(None, dims, rel, adj)
feats = {m: torch.nn.Embedding(len(node_maps[m]) + 1, d) for m in rel}
enc = utils.get_encoder(0, graph, dims, feats, True, node_maps=node_maps)
model = QueryEncoderDecoder(graph, enc, utils.get_metapath_decoder(graph, dims, "bilinear-diag"), utils.get_intersection_decoder(graph, dims, "min"))
sampler = NativeSampler(graph, node_maps)
val = {"one_neg": {}, "full_neg": {}}
for k, t in enumerate(["2-chain", "3-chain", "2-inter", "3-inter", "3-inter_chain"]):
    by = data_utils.group_by_formula(sampler.sample(n_q, q_type=t, neg_sample_max=n_neg, seed=k, threads=8).to_queries(keep_graph=False))[t]
    val["one_neg"][t] = by
    val["full_neg"][t] = by


class Quiet(object):
    def info(self, m):
        pass


for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    scores = train_helpers.evaluate(model, val, 0, Quiet())
    torch.cuda.synchronize()
    print("evaluate: %.3f s for %d queries per type x 5 types, <= %d negatives per query; AUCs %s" % (
        time.perf_counter() - t0, n_q, n_neg, {k: round(v, 4) for k, v in scores.items()}), flush=True)
