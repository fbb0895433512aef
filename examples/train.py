#!/usr/bin/env python3
"""The reference's training script (netquery/bio/train.py) on this package: the same arguments, the same calls, FusedAdam / FusedSGD in
place of torch.optim's.  ``--data_dir`` = the reference's pickles (graph_data.pkl, train_edges.pkl, val_queries_2.pkl, ...);
without it a synthetic graph of the Bio data's shape (graphqembed_amd.data_utils.make_synthetic_graph) with query lists drawn by the
native sampler — no data set ships with this repository.

    python examples/train.py --max_iter 20000 --val_every 5000          # synthetic, ~2 s of training on one MI355X
    python examples/train.py --data_dir ./bio_data --cuda               # the reference's invocation

Prints the wall time of each stage; the model file is a ``state_dict`` with the reference's key names."""
import os, random, sys, time
from argparse import ArgumentParser
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from graphqembed_amd import data_utils
from graphqembed_amd.data_utils import load_queries_by_formula, load_test_queries_by_formula
from graphqembed_amd.graph import Graph, Query
from graphqembed_amd.model import FusedAdam, FusedSGD, QueryEncoderDecoder
from graphqembed_amd.train_helpers import run_train
from graphqembed_amd.utils import get_encoder, get_intersection_decoder, get_metapath_decoder, setup_logging

parser = ArgumentParser()
parser.add_argument("--embed_dim", type=int, default=128)
parser.add_argument("--data_dir", type=str, default=None)
parser.add_argument("--lr", type=float, default=0.01)
parser.add_argument("--depth", type=int, default=0)
parser.add_argument("--batch_size", type=int, default=512)
parser.add_argument("--max_iter", type=int, default=100000000)
parser.add_argument("--max_burn_in", type=int, default=1000000)
parser.add_argument("--val_every", type=int, default=5000)
parser.add_argument("--tol", type=float, default=0.0001)
parser.add_argument("--cuda", action="store_true", help="accepted for compatibility: the hot path always runs on the GPU")
parser.add_argument("--log_dir", type=str, default="./")
parser.add_argument("--model_dir", type=str, default="./")
parser.add_argument("--decoder", type=str, default="bilinear")
parser.add_argument("--inter_decoder", type=str, default="mean")
parser.add_argument("--opt", type=str, default="adam")
parser.add_argument("--seed", type=int, default=0)
parser.add_argument("--synthetic_queries", type=int, default=40000, help="synthetic data: queries sampled per query type")
args = parser.parse_args()
if args.depth != 0:
    raise SystemExit("only the DirectEncoder (depth 0) is on the accelerated path")
random.seed(args.seed); np.random.seed(args.seed); torch.manual_seed(args.seed)
clock = time.perf_counter
t0 = clock()
if args.data_dir:
    rels, adj_lists, node_ids = data_utils.load_graph_data(args.data_dir)        # (the pickle's third element: the id list of every mode)
    node_maps = data_utils.make_node_maps(node_ids)                              # bio/data_utils.py:11-13
else:
    rels, adj_lists, node_ids = data_utils.make_synthetic_graph(data_utils.BIO_SYNTH_SIZES, seed=args.seed)
    node_maps = data_utils.make_node_maps(node_ids)
out_dims = {mode: args.embed_dim for mode in rels}
graph = Graph(None, out_dims, rels, adj_lists)
feature_modules = {m: torch.nn.Embedding(len(node_maps[m]) + 1, args.embed_dim) for m in rels}     # bio/data_utils.py:14-17
for f in feature_modules.values():
    f.weight.data.normal_(0, 1.0 / args.embed_dim)
print("graph: %.1f s" % (clock() - t0)); t0 = clock()

if args.data_dir:
    train_queries = load_queries_by_formula(args.data_dir + "/train_edges.pkl")
    val_queries = load_test_queries_by_formula(args.data_dir + "/val_edges.pkl")
    test_queries = load_test_queries_by_formula(args.data_dir + "/test_edges.pkl")
    for i in range(2, 4):
        train_queries.update(load_queries_by_formula(args.data_dir + "/train_queries_{:d}.pkl".format(i)))
        for held, name in ((val_queries, "val"), (test_queries, "test")):
            more = load_test_queries_by_formula(args.data_dir + "/{:s}_queries_{:d}.pkl".format(name, i))
            held["one_neg"].update(more["one_neg"])
            held["full_neg"].update(more["full_neg"])
else:
    from graphqembed_amd.sampler import NativeSampler
    sampler = NativeSampler(graph, node_maps)
    n = args.synthetic_queries
    edges = graph.get_all_edges(seed=args.seed)

    # 1-chain training queries carry no negatives: model.py:113-114 draws them from the whole mode
    train_queries = {"1-chain": dict(data_utils.group_by_formula([Query(("1-chain", e), None, None) for e in edges[:3 * n // 2]])["1-chain"])}
    val_queries = {"one_neg": {}, "full_neg": {}}
    test_queries = {"one_neg": {}, "full_neg": {}}
    for k, t in enumerate(["2-chain", "3-chain", "2-inter", "3-inter", "3-inter_chain"]):
        # (query_lists: the sampled queries as lists that ARE row arrays — no Query object per sample; .to_queries() would build them)
        by = sampler.sample(n, q_type=t, neg_sample_max=20, seed=args.seed + k, threads=8).query_lists()[t]
        keep = sorted(by, key=lambda f: -len(by[f]))[:8]
        train_queries[t] = {f: by[f][:-100] for f in keep if len(by[f]) > 200}
        val = {f: by[f][-100:-50] for f in train_queries[t]}
        tst = {f: by[f][-50:] for f in train_queries[t]}
        val_queries["one_neg"][t] = val_queries["full_neg"][t] = val
        test_queries["one_neg"][t] = test_queries["full_neg"][t] = tst
print("queries: %.1f s (%d training queries)" % (clock() - t0, sum(len(q) for by in train_queries.values() for q in by.values()))); t0 = clock()

enc = get_encoder(args.depth, graph, out_dims, feature_modules, True, node_maps=node_maps)
dec = get_metapath_decoder(graph, out_dims, args.decoder)
inter_dec = get_intersection_decoder(graph, out_dims, args.inter_decoder)
enc_dec = QueryEncoderDecoder(graph, enc, dec, inter_dec)
optimizer = FusedSGD(enc_dec, lr=args.lr) if args.opt == "sgd" else FusedAdam(enc_dec, lr=args.lr)
name = "{data:s}-{depth:d}-{embed_dim:d}-{lr:f}-{decoder:s}-{inter_decoder:s}".format(
    data=(args.data_dir or "synthetic").strip().split("/")[-1], depth=args.depth, embed_dim=args.embed_dim, lr=args.lr, decoder=args.decoder,
    inter_decoder=args.inter_decoder)
logger = setup_logging(args.log_dir + "/" + name + ".log")
print("model: %.1f s" % (clock() - t0)); t0 = clock()
if not args.data_dir:       # no held-out 1-chain lists with negatives in the synthetic set: the edge phase is bounded by max_burn_in alone
    args.max_burn_in = min(args.max_burn_in, max(1, args.val_every - 1))
run_train(enc_dec, optimizer, train_queries, val_queries, test_queries, logger, max_burn_in=args.max_burn_in, batch_size=args.batch_size,
          val_every=args.val_every, max_iter=args.max_iter, model_file=args.model_dir + "/" + name + ".model")
torch.cuda.synchronize()
print("run_train: %.1f s" % (clock() - t0))
torch.save(enc_dec.state_dict(), args.model_dir + "/" + name + ".model")
