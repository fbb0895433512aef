"""Helpers to read tests/golden/*.npz (schema: oracle/make_golden.py docstring)."""
import glob
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def to_rels(x):
    return tuple(to_rels(y) if isinstance(y[0], list) else tuple(y) for y in x)


def rels_to_lists(rels):
    return [rels_to_lists(r) if isinstance(r[0], (tuple, list)) else list(r) for r in rels]


def batch_signature(crc, query_type, rels_json, target, neg, anchors):
    """Running CRC32 over an iteration's batches (trainlong_*.npz: sig_full with the formula's relations as
    oracle/make_golden.py's rels_to_json writes them, sig_rows with ``rels_json=None``): type name, relations, then the int32
    target / negative / anchor rows."""
    import zlib
    crc = zlib.crc32(query_type.encode(), crc)
    if rels_json is not None:
        crc = zlib.crc32(json.dumps(rels_json).encode(), crc)
    for a in (target, neg, anchors):
        crc = zlib.crc32(np.ascontiguousarray(a, dtype=np.int32).tobytes(), crc)
    return crc & 0xffffffff


def parse_train_log(lines):
    """What run_train logged (train_helpers.py:19-38, 52-55, 80-93) as data: {"iters": [(iteration, ema_loss)],
    "evals": [{tag: (auc, perc)} per evaluation, in order, with "iteration"], "edge_conv": iteration or None,
    "macro": float or None, "improvement": float or None}."""
    out = {"iters": [], "evals": [], "edge_conv": None, "macro": None, "improvement": None}
    cur, cur_it = None, None
    for l in lines:
        if " val AUC: " in l:
            tag, rest = l.split(" val AUC: ")
            auc, rest = rest.split(" val perc ")
            perc, it = rest.split("; iteration: ")
            if cur is None or int(it) != cur_it or tag in cur:
                cur, cur_it = {}, int(it)
                out["evals"].append({"iteration": cur_it, "scores": cur})
            cur[tag] = (float(auc), float(perc))
            continue
        cur = None
        if l.startswith("Iter: "):
            a, b = l.split("; ema_loss: ")
            out["iters"].append((int(a[6:]), float(b)))
        elif l.startswith("Edge converged at iteration "):
            out["edge_conv"] = int(l.rsplit(" ", 1)[1])
        elif l.startswith("Test macro-averaged val: "):
            out["macro"] = float(l.rsplit(" ", 1)[1])
        elif l.startswith("Improvement from edge conv: "):
            out["improvement"] = float(l.rsplit(" ", 1)[1])
    return out


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
