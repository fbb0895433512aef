#!/usr/bin/env python3
"""The reference's Reddit training script (netquery/reddit/new_train.py:1-80) on this package: the same arguments and defaults, the
same file names, the same calls — ``load_graph`` from graphqembed_amd.reddit_data, FusedAdam / FusedSGD in place of torch.optim's.

    python examples/train_reddit.py --data_dir ./gaming_graph --cuda          # the reference's invocation
    python examples/train_reddit.py --data_dir ./gaming_flat --flat           # a directory written by tools/convert_data.py --reddit
    python examples/train_reddit.py --make_synthetic /tmp/reddit_synth && python examples/train_reddit.py --data_dir /tmp/reddit_synth --max_iter 2000 --val_every 500 --max_burn_in 500

``--data_dir`` holds adj_lists.pkl / rels.pkl / post_words.pkl, train_edges.pkl, {val,test}_edges-split.pkl, train_queries_{2,3}.pkl and
{val,test}_queries_{2,3}-clean.pkl (new_train.py:29-46).  Two differences from the reference script, both so that a run can be
bounded: ``--batch_size`` and ``--max_iter`` are handed to ``run_train`` (the reference parses them and then calls run_train with its
defaults, new_train.py:78).  ``--make_synthetic DIR`` writes a small data set of that layout (a random Reddit-shaped graph; no data set
ships with this repository) and exits."""
import os, random, sys, time
from argparse import ArgumentParser
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from graphqembed_amd import reddit_data
from graphqembed_amd.data_utils import load_queries_by_formula, load_test_queries_by_formula
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
parser.add_argument("--flat", action="store_true", help="--data_dir was written by tools/convert_data.py --reddit")
parser.add_argument("--make_synthetic", type=str, default=None, metavar="DIR")
args = parser.parse_args()
if args.make_synthetic:
    reddit_data_dir = args.make_synthetic
    from graphqembed_amd.reddit_data import write_synthetic_dataset as write_reddit_dataset
    print(write_reddit_dataset(reddit_data_dir, seed=args.seed))
    raise SystemExit(0)
if not args.data_dir:
    raise SystemExit("--data_dir is required (or --make_synthetic DIR to write a small data set first)")
if args.depth != 0:
    raise SystemExit("only the DirectEncoder (depth 0) is on the accelerated path")
random.seed(args.seed); np.random.seed(args.seed); torch.manual_seed(args.seed)
clock = time.perf_counter
t0 = clock()
print("Loading graph data..")
if args.flat:
    from graphqembed_amd import flatdata
    graph, feature_modules, flat = reddit_data.load_flat_graph(args.data_dir, args.embed_dim)
    ext = ".npz"
    load_train = lambda path: flatdata.load_queries_by_formula(path, flat)
    load_test = lambda path: flatdata.load_test_queries_by_formula(path, flat)
else:
    graph, feature_modules = reddit_data.load_graph(args.data_dir, args.embed_dim, cuda=args.cuda)
    ext = ".pkl"
    load_train, load_test = load_queries_by_formula, load_test_queries_by_formula
out_dims = {mode: args.embed_dim for mode in graph.relations}
print("graph: %.1f s" % (clock() - t0)); t0 = clock()

print("Loading edge data..")
train_queries = load_train(args.data_dir + "/train_edges" + ext)
val_queries = load_test(args.data_dir + "/val_edges-split" + ext)
test_queries = load_test(args.data_dir + "/test_edges-split" + ext)
print("Loading query data..")
for i in range(2, 4):
    train_queries.update(load_train(args.data_dir + "/train_queries_{:d}".format(i) + ext))
    for held, name in ((val_queries, "val"), (test_queries, "test")):
        more = load_test(args.data_dir + "/{:s}_queries_{:d}-clean".format(name, i) + ext)
        held["one_neg"].update(more["one_neg"])
        held["full_neg"].update(more["full_neg"])
print("queries: %.1f s (%d training queries)" % (clock() - t0, sum(len(q) for by in train_queries.values() for q in by.values()))); t0 = clock()

enc = get_encoder(args.depth, graph, out_dims, feature_modules, args.cuda)
dec = get_metapath_decoder(graph, out_dims, args.decoder)
inter_dec = get_intersection_decoder(graph, out_dims, args.inter_decoder)
enc_dec = QueryEncoderDecoder(graph, enc, dec, inter_dec)
optimizer = FusedSGD(enc_dec, lr=args.lr) if args.opt == "sgd" else FusedAdam(enc_dec, lr=args.lr)
name = "{data:s}-{depth:d}-{embed_dim:d}-{lr:f}-{decoder:s}-{inter_decoder:s}".format(
    data=args.data_dir.strip().rstrip("/").split("/")[-1], depth=args.depth, embed_dim=args.embed_dim, lr=args.lr, decoder=args.decoder,
    inter_decoder=args.inter_decoder)
logger = setup_logging(args.log_dir + "/" + name + ".log")
model_file = args.model_dir + "/" + name + ".model"      # (the reference names its model file *.log as well, new_train.py:69-75, in another directory)
print("model: %.1f s" % (clock() - t0)); t0 = clock()
run_train(enc_dec, optimizer, train_queries, val_queries, test_queries, logger, max_burn_in=args.max_burn_in, batch_size=args.batch_size,
          val_every=args.val_every, max_iter=args.max_iter, model_file=model_file)
torch.cuda.synchronize()
print("run_train: %.1f s" % (clock() - t0))
torch.save(enc_dec.state_dict(), model_file)
