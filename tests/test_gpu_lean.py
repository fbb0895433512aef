"""The LEAN instantiations of the backward kernels (csrc/gqe_fused.h: launches without EmbeddingBag roles, fetched rows or a
debug profile run the same source compiled without those branches) against the plain ones: same inputs, two processes
(GQE_NO_LEAN is read once per process), same results up to the last bits.  Everything a tile computes is deterministic — the scores, the per-batch
hinge sums, every row's gradient contribution (lists are summed in list order, which both kernels build identically up to the
order of concurrent pushes); the relation vectors' and matrices' gradients are float atomics: compared to a few ulps of their
largest element."""
import os, subprocess, sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _run(tmp_path, name, env_extra):
    out = str(tmp_path / (name + ".npz"))
    env = dict(os.environ)
    env.pop("GQE_NO_LEAN", None)
    env.update(env_extra)
    r = subprocess.run([sys.executable, os.path.join(HERE, "lean_child.py"), out], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return dict(np.load(out))


def test_lean_and_plain_kernels_agree(tmp_path):
    lean = _run(tmp_path, "lean", {})
    plain = _run(tmp_path, "plain", {"GQE_NO_LEAN": "1"})
    assert sorted(lean) == sorted(plain) and len(lean) > 40
    for k in sorted(lean):
        a, b = lean[k], plain[k]
        if k.endswith(("_pos", "_neg")):
            # scores: no atomics on their path — but two instantiations are two compilations (fused multiply-adds are contracted
            # where the scheduler finds them): cosines agree to the last bit or two
            np.testing.assert_allclose(a, b, rtol=0, atol=3e-7, err_msg=k)
        elif k.endswith("_losses"):
            np.testing.assert_allclose(a, b, rtol=2e-6, atol=1e-7, err_msg=k)   # per-tile hinge sums, summed per batch
        else:
            scale = max(float(np.abs(b).max()), 1e-12)
            assert float(np.abs(a - b).max()) <= 2e-5 * scale, (k, float(np.abs(a - b).max()), scale)
