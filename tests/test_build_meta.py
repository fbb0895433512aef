"""What the built library contains (no GPU needed): the code objects inside graphqembed_amd/libgqe.so carry the registers
and scratch of every kernel.  The hot kernels must not use scratch — a fused tile that spills (the d = 256 kernels sit
at the 128-VGPR limit of a 16-wave workgroup) loses the registers-for-the-whole-kernel design, and a spilling guarded
variant once lost lanes of a gradient."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
LIB = os.path.join(ROOT, "graphqembed_amd", "libgqe.so")


@pytest.fixture(scope="module")
def kernels():
    if not os.path.exists(LIB):
        import __graft_entry__ as g
        g.build()
    from kernel_meta import kernel_metadata
    ks = kernel_metadata(LIB)
    assert len(ks) > 100, "expected the fused / optimiser / scoring kernels of every variant, found %d" % len(ks)
    return ks


def test_sixteen_wave_full_kernels_use_no_scratch(kernels):
    """d = 64 / 128 / 256 (FULL) tiles of 16 waves, forward and backward, every decoder and both intersection kinds."""
    from kernel_meta import fused_variant
    seen = 0
    for k in kernels:
        v = fused_variant(k["name"])
        if not v:
            continue
        dec, mlp, nc, full, bwd, fw = v
        if fw == 16 and full:
            seen += 1
            assert k["scratch"] == 0 and k["vgpr_spill"] == 0, "%s: %d B scratch, %d spilled VGPRs" % (k["name"][:60], k["scratch"], k["vgpr_spill"])
            assert k["vgpr"] <= 128, k
    assert seen == 3 * 2 * 3 * 2, seen     # DEC x MLP x NC in {1, 2, 4} x {fwd, bwd}


def test_streaming_and_gemm_kernels_use_no_scratch(kernels):
    for k in kernels:
        n = k["name"]
        if n.startswith("_Z14gqe_opt_kernel") or n.startswith("_Z20gqe_pair_gemm_kernel") or "gqe_eval_score_kernel" in n or "gqe_rows_kernel" in n:
            assert k["scratch"] == 0, (n[:60], k["scratch"])
        if n.startswith("_Z14gqe_opt_kernelILi0E"):   # Adam pass: 8 waves per SIMD needs <= 64 VGPRs
            assert k["vgpr"] <= 64, (n[:60], k["vgpr"])


def test_guarded_diag_and_transe_kernels_do_not_spill(kernels):
    """bilinear-diag / TransE at any d <= 192 the dispatcher can select (d % 64 != 0: the 8-wave kernels with 256 VGPRs per
    lane above d = 64, the 16-wave ones below): spilling variants of these once corrupted a relation gradient (d = 144)
    and faulted (d = 96).  The d in (192, 256) and the full-Bilinear guarded kernels still spill a few registers; the
    parity matrix runs them (tests/test_gpu_parity.py::test_random_schema_vs_oracle: every multiple of 16 up to 256)."""
    from kernel_meta import fused_variant
    checked = 0
    for k in kernels:
        v = fused_variant(k["name"])
        if not v:
            continue
        dec, mlp, nc, full, bwd, fw = v
        if dec in (0, 1) and ((fw == 16 and nc == 1) or (fw == 8 and nc in (2, 3))):
            checked += 1
            assert k["vgpr_spill"] == 0, (k["name"][:60], k["vgpr_spill"])
    assert checked >= 2 * 2 * 2 * 4, checked
    # the shapes that are no longer built: guarded d in (64, 128) on 16 waves
    assert not any(fused_variant(k["name"]) and fused_variant(k["name"])[2:4] == (2, 0) and fused_variant(k["name"])[5] == 16 for k in kernels)
