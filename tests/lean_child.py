"""Child process of tests/test_gpu_lean.py: one grouped margin launch (every query type, ragged batches, hub rows) on the toy
schema, 16- and 8-wave shapes; prints the losses, the scores and the table / dense gradients as a .npz path.  The parent runs it
twice — with and without GQE_NO_LEAN=1 (read once per process by the launcher, csrc/gqe_fused.h) — and compares."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gpu_utils import TOY_FORMULAS, TOY_KINDS, TOY_SIZES, engine_from_params, plan_for, random_params, read_arena, toy_batch
from graphqembed_amd.tensorize import pack_margin_batches

out = {}
for dec, inter, d, B in (("bilinear-diag", "min", 128, 37), ("bilinear", "mean", 64, 21), ("transe", "min-simple", 128, 19),
                         ("bilinear-diag", "min", 128, 1200)):        # (the last one: 525 tiles, the 8-wave COMPACT shape)
    rng = np.random.RandomState(17 + d + B)
    params = random_params(rng, d, dec, inter, TOY_SIZES, TOY_KINDS)
    eng = engine_from_params(params, d, dec, inter, max_queries=7 * max(B, 64))
    items = []
    for j, qtype in enumerate(TOY_FORMULAS):
        t, g, a = toy_batch(rng, qtype, B - j, hub=(j % 2 == 0))
        items.append((plan_for(eng, qtype, TOY_FORMULAS[qtype]), t, g, a, [1.0, 0.01, 0.01, 0.005, 0.005, 0.5, 2.0][j % 7], 1.0))
    descs, idx, n = pack_margin_batches(items)
    losses, pos, neg = eng.margin_fwd_bwd(descs, idx, n, want_scores=True)
    tag = "%s_%s_%d_%d" % (dec, inter, d, B)
    out[tag + "_losses"] = losses.cpu().numpy()
    out[tag + "_pos"] = pos.cpu().numpy()
    out[tag + "_neg"] = neg.cpu().numpy()
    for k, v in read_arena(eng, eng.grads).items():
        out[tag + "_g_" + k] = v
    eng.close()
np.savez(sys.argv[1], **out)
