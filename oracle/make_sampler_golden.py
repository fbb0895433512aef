#!/usr/bin/env python3
"""ORACLE tooling — golden data for the query SAMPLER, produced by running the reference's own
``netquery.graph.Graph`` (imported as in make_golden.py; build container only, writes DATA only):

  tests/golden/sampler_ref.json
    "type_counts": {"2": {type: n}, "3": {type: n}}   accepted query types of Graph.sample_queries(arity, 4000, 1)
    "queries": [{"graph": <query graph, relations as lists>, "negs": [...], "hard": [...] | null}, ...]
        the FULL negative / hard-negative node sets Graph.get_negative_samples returns for 40 sampled queries of
        every type (not the sub-sampled lists a Query keeps)

    python oracle/make_sampler_golden.py
"""
import collections
import json
import logging
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.make_golden import OUT, World, import_reference, seed_all  # noqa: E402


def listify(x):
    return [listify(y) for y in x] if isinstance(x, (tuple, list)) else x


def main():
    tmp = import_reference()
    logging.disable(logging.CRITICAL)
    try:
        world = World(32)
        g = world.graph
        out = {"type_counts": {}, "queries": []}
        for arity in (2, 3):
            seed_all(100 + arity)
            qs = g.sample_queries(arity, 4000, 1, verbose=False)
            out["type_counts"][str(arity)] = dict(collections.Counter(q.formula.query_type for q in qs))
        seed_all(7)
        for qt in ("2-chain", "3-chain", "2-inter", "3-inter", "3-inter_chain", "3-chain_inter"):
            got = 0
            while got < 40:
                q = g.sample_query_subgraph_bytype(qt)
                if q is None:
                    continue
                negs, hard = g.get_negative_samples(q)
                if negs is None or ("inter" in qt and hard is None):
                    continue
                out["queries"].append({"graph": listify(q), "negs": sorted(negs), "hard": None if hard is None else sorted(hard)})
                got += 1
        with open(os.path.join(OUT, "sampler_ref.json"), "w") as f:
            json.dump(out, f)
        print({k: v for k, v in out["type_counts"].items()}, len(out["queries"]))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
