#!/usr/bin/env python3
"""Where a fed iteration's time goes on the host (gqe_feeder_host_seconds): bench.py's api_path world (run_train's native runs with
the reference's random streams replayed) — sampling + packing per iteration, the whole host share, the wall time per iteration.
python tools/probes/feeder_host_probe.py"""
import os, random, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from graphqembed_amd import data_utils, train_helpers, utils
from graphqembed_amd.graph import Graph, Query
from graphqembed_amd.model import FusedAdam, QueryEncoderDecoder
from graphqembed_amd.sampler import NativeSampler
d, B = 128, 512
rel, adj, ids = data_utils.make_synthetic_graph(data_utils.BIO_SYNTH_SIZES, seed=0)
node_maps = data_utils.make_node_maps(ids)
dims = {m: d for m in rel}
graph = Graph(None, dims, rel, adj)
feats = {m: torch.nn.Embedding(len(node_maps[m]) + 1, d) for m in rel}
enc = utils.get_encoder(0, graph, dims, feats, True, node_maps=node_maps)
model = QueryEncoderDecoder(graph, enc, utils.get_metapath_decoder(graph, dims, "bilinear-diag"), utils.get_intersection_decoder(graph, dims, "min"),
                            max_queries=9 * B, max_batches=9)
sampler = NativeSampler(graph, node_maps)
train = {"1-chain": dict(data_utils.group_by_formula([Query(("1-chain", e), None, None) for e in graph.get_all_edges(seed=0)[:60000]])["1-chain"])}
for k, t in enumerate(["2-chain", "3-chain", "2-inter", "3-inter", "3-inter_chain"]):
    by = data_utils.group_by_formula(sampler.sample(40000, q_type=t, neg_sample_max=20, seed=k, threads=8).to_queries(keep_graph=False))[t]
    train[t] = {f: by[f] for f in sorted(by, key=lambda f: -len(by[f]))[:6]}
opt = FusedAdam(model, lr=0.01)
ex = train_helpers.FusedExecutor(model, opt)
loop = train_helpers._NativeLoop(ex, train, B, 0.01, 0.005)
random.seed(0); np.random.seed(0)
loop.run(0, 200, True)
torch.cuda.synchronize()
b0, r0 = model.engine.feeder_host_seconds(loop.feeder)
q0 = model.engine.feeder_queries(loop.feeder)
t0 = time.perf_counter()
n = 2000
it = 200
while it < 200 + n:
    loop.run(it, 500, True)
    it += 500
torch.cuda.synchronize()
wall = time.perf_counter() - t0
b1, r1 = model.engine.feeder_host_seconds(loop.feeder)
q = model.engine.feeder_queries(loop.feeder) - q0
print("per iteration: wall %.1f us | inside gqe_feeder_run %.1f us | sampling + packing %.1f us | %.1f queries | %.1f M queries/s"
      % (wall / n * 1e6, (r1 - r0) / n * 1e6, (b1 - b0) / n * 1e6, q / n, q / wall / 1e6))
