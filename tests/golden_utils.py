"""Helpers to read tests/golden/*.npz (schema: oracle/make_golden.py docstring)."""
import glob
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def to_rels(x):
    return tuple(to_rels(y) if isinstance(y[0], list) else tuple(y) for y in x)


def model_files(d=None):
    out = []
    for p in sorted(glob.glob(os.path.join(GOLDEN, "model_*.npz"))):
        _, dec, inter, dd = os.path.basename(p)[:-4].split("_")
        if d is None or int(dd[1:]) == d:
            out.append((p, dec, inter, int(dd[1:])))
    return out


def load_tables(d):
    z = np.load(os.path.join(GOLDEN, "tables_d%d.npz" % d))
    return {k: z[k] for k in z.files}


def load_params(z, d):
    """tables + decoder params of one fixture file -> {state_dict key: array}."""
    params = load_tables(d)
    for k in z.files:
        if k.startswith("param/"):
            params[k[6:]] = z[k]
    return params


def case_names(z):
    return sorted(set(k.split("/")[0] for k in z.files if k.endswith("/meta") and not k.startswith("it")))


def load_case(z, case):
    meta = json.loads(str(z[case + "/meta"]))
    out = {"type": meta["type"], "rels": to_rels(meta["rels"]), "hard": meta.get("hard", False),
           "margin": meta.get("margin", 1)}
    for name in ("target", "neg", "anchors", "pos", "negscore", "loss", "scores"):
        if case + "/" + name in z.files:
            out[name] = z[case + "/" + name]
    pre = case + "/grad/"
    out["grads"] = {k[len(pre):]: z[k] for k in z.files if k.startswith(pre)}
    pre = case + "/adam/delta/"
    out["adam_delta"] = {k[len(pre):]: z[k] for k in z.files if k.startswith(pre)}
    if case + "/adam/neg" in z.files:
        out["adam_neg"] = z[case + "/adam/neg"]
        out["adam_loss"] = z[case + "/adam/loss"]
    return out


def reddit_files():
    out = []
    for p in sorted(glob.glob(os.path.join(GOLDEN, "reddit_*.npz"))):
        _, dec, inter, dd = os.path.basename(p)[:-4].split("_")
        out.append((p, dec, inter, int(dd[1:])))
    return out


def load_reddit_params(z, with_bags=True):
    """All parameters of a reddit_*.npz fixture (+ the oracle's bag registry under '__bags__')."""
    params = {k[6:]: z[k] for k in z.files if k.startswith("param/")}
    if with_bags:
        params["__bags__"] = {"post": (z["bag/post/ptr"], z["bag/post/ids"])}
    return params


def eval_files():
    out = []
    for p in sorted(glob.glob(os.path.join(GOLDEN, "eval_*.npz"))):
        _, dec, inter, dd = os.path.basename(p)[:-4].split("_")
        out.append((p, dec, inter, int(dd[1:])))
    return out


def adam1_files():
    out = []
    for p in sorted(glob.glob(os.path.join(GOLDEN, "adam1_*.npz"))):
        _, dec, inter, dd = os.path.basename(p)[:-4].split("_")
        out.append((p, dec, inter, int(dd[1:])))
    return out


def eval_calls(z):
    """The forward calls an eval fixture recorded: [(type, rels, target rows, anchor rows [k, n], scores)]."""
    n = 0
    while "call%d/meta" % n in z.files:
        meta = json.loads(str(z["call%d/meta" % n]))
        yield n, meta["type"], to_rels(meta["rels"]), z["call%d/target" % n], z["call%d/anchors" % n], z["call%d/scores" % n]
        n += 1
