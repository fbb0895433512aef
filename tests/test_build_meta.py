"""What the built library contains (no GPU needed): the code objects inside graphqembed_amd/libgqe.so carry the registers
and scratch of every kernel, and the build keeps the compiler's device assembly (csrc/obj/*/*.s).

Round 2 saw two fused-kernel variants that spill registers corrupt results (d = 144: lanes >= 16 of a gradient lost; d = 96
/ 80: a memory fault).  Round 3 found the cause with rocgdb on the faulting wave: the register allocator
placed a live-range-split copy and spill stores at the top of the JOIN block of a guarded store, ahead of the
`s_or_b64 exec` that ends the guarded region, so they ran with lanes >= 16 masked off.  Hence two rules, checked here for
EVERY instantiation the dispatcher can select for a configuration gqe_create accepts:
  * no spilled VGPRs;
  * no allocator-inserted instruction ahead of an EXEC restore (tools/exec_check.py on the kept assembly).
Every (decoder, intersection, d % 16 == 0) configuration is accepted (gqe_dim_supported) and has spill-free kernels."""
import ctypes
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
LIB = os.path.join(ROOT, "graphqembed_amd", "libgqe.so")
DIMS = list(range(16, 257, 16))
INTERS = {"min": 0, "mean": 1, "min-simple": 2, "mean-simple": 3}


@pytest.fixture(scope="module")
def kernels():
    if not os.path.exists(LIB):
        import __graft_entry__ as g
        g.build()
    from kernel_meta import kernel_metadata
    ks = kernel_metadata(LIB)
    assert len(ks) > 100, "expected the fused / optimiser / scoring kernels of every variant, found %d" % len(ks)
    return ks


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(LIB):
        import __graft_entry__ as g
        g.build()
    from graphqembed_amd.engine import load_library
    return load_library()


def selectable(lib):
    """{(DEC, MLP, NC, FULL, BWD, FW)} the dispatcher can run for the configurations gqe_create accepts, with the
    (decoder, inter, d) that reach each."""
    out = {}
    for dec in (0, 1, 2):
        for name, inter in INTERS.items():
            for d in DIMS:
                if not lib.gqe_dim_supported(dec, inter, d):
                    continue
                for tiles in (1, 100000):                   # few tiles / more tiles than GQE_FW8_MIN_TILES
                    v = (ctypes.c_int32 * 3)()
                    assert lib.gqe_debug_fused_variant(dec, d, tiles, v) == 0
                    for bwd in (0, 1):
                        out.setdefault((dec, 1 if inter < 2 else 0, v[0], v[1], bwd, v[2]), []).append((dec, name, d))
    return out


def test_supported_configurations(lib):
    """Every multiple of 16 up to 256 with every decoder / intersection pair (the reference takes any --embed_dim,
    bio/train.py:13); nothing else."""
    for dec in (0, 1, 2):
        for inter in INTERS.values():
            for d in DIMS:
                assert lib.gqe_dim_supported(dec, inter, d), (dec, inter, d)
    assert not lib.gqe_dim_supported(0, 0, 24) and not lib.gqe_dim_supported(0, 0, 272) and not lib.gqe_dim_supported(1, 2, 0)
    assert not lib.gqe_dim_supported(3, 0, 128) and not lib.gqe_dim_supported(-1, 0, 128)


def test_no_unreachable_fused_kernels_are_built(kernels, lib):
    """Every gqe_fused_kernel instantiation in the library is one the dispatcher can select (round 3 shipped 14 spilling
    variants that nothing could reach)."""
    from kernel_meta import fused_variant
    built = set(fused_variant(k["name"]) for k in kernels if fused_variant(k["name"]))
    sel = set(selectable(lib))
    assert built == sel, (sorted(built - sel), sorted(sel - built))


def test_every_selectable_fused_kernel_is_free_of_spills(kernels, lib):
    from kernel_meta import fused_variant
    built = {}
    for k in kernels:       # (a variant may exist twice: the plain and the LEAN instantiation — both have to be clean)
        if fused_variant(k["name"]):
            built.setdefault(fused_variant(k["name"]), []).append(k)
    sel = selectable(lib)
    assert len(sel) >= 60, len(sel)
    for v, cfgs in sorted(sel.items()):
        assert v in built, "the dispatcher selects %r (%r) but the library does not contain it" % (v, cfgs[0])
        for k in built[v]:
            assert k["vgpr_spill"] == 0, "gqe_fused_kernel<DEC=%d, MLP=%d, NC=%d, FULL=%d, BWD=%d, FW=%d> spills %d VGPRs; reached by %r" % (v + (k["vgpr_spill"], cfgs[0]))
            assert k["vgpr"] <= (128 if v[5] == 16 else 256), (v, k["vgpr"])


def test_fused_kernels_use_no_scratch(kernels, lib):
    """Every fused kernel the dispatcher can select — every decoder, both intersection kinds, every accepted d, 16- and 8-wave
    shapes, forward and backward: no scratch at all (a spilled register or a dynamically indexed local array would show here;
    round 3's TransE 8-wave d = 128 kernels carried such an array, 40 B, without a spill)."""
    from kernel_meta import fused_variant
    sel = selectable(lib)
    seen = 0
    for k in kernels:
        v = fused_variant(k["name"])
        if not v or v not in sel:
            continue
        seen += 1
        assert k["scratch"] == 0 and k["vgpr_spill"] == 0, "%s: %d B scratch, %d spilled VGPRs" % (k["name"][:60], k["scratch"], k["vgpr_spill"])
        assert k["vgpr"] <= (128 if v[5] == 16 else 256), k
    # DEC x MLP x (16-wave NC / FULL variants + the 8-wave shape) x {fwd, bwd} + the LEAN instantiations of the FULL backward kernels
    assert seen >= len(sel) and seen >= 3 * 2 * (7 + 1) * 2, seen


def test_lean_instantiations_exist_where_the_launcher_picks_them(kernels):
    """gqe_fused_kernel<..., LEAN = true> (no EmbeddingBag role, no fetched rows, no profile: gqe_fused.h) is built for exactly
    the backward kernels of the straight-line dims (FULL), next to the plain instantiation."""
    from kernel_meta import fused_lean, fused_variant
    lean = set(fused_variant(k["name"]) for k in kernels if fused_lean(k["name"]) == 1)
    plain = set(fused_variant(k["name"]) for k in kernels if fused_lean(k["name"]) == 0)
    assert lean and lean <= plain
    assert lean == set(v for v in plain if v[3] == 1 and v[4] == 1), sorted(lean ^ set(v for v in plain if v[3] == 1 and v[4] == 1))
    # LEAN == 2: the row-sharded launch (fetched rows as a compile-time fact) — the 16-wave kernels of the same set
    sharded = set(fused_variant(k["name"]) for k in kernels if fused_lean(k["name"]) == 2)
    assert sharded == set(v for v in lean if v[5] == 16), sorted(sharded ^ set(v for v in lean if v[5] == 16))


def test_streaming_and_gemm_kernels_use_no_scratch(kernels):
    for k in kernels:
        n = k["name"]
        if n.startswith("_Z14gqe_opt_kernel") or n.startswith("_Z20gqe_pair_gemm_kernel") or "gqe_eval_score_kernel" in n or "gqe_rows_kernel" in n:
            assert k["scratch"] == 0, (n[:60], k["scratch"])
        if n.startswith("_Z14gqe_opt_kernelILi0E"):   # Adam pass: 8 waves per SIMD needs <= 64 VGPRs
            assert k["vgpr"] <= 64, (n[:60], k["vgpr"])
    # the pass that carries the deferred pair GEMM's units (gqe_set_deferred_gemm): the units must not cost the streaming chunks
    # their occupancy — <= 64 registers in total (accumulators included), no LDS beyond the pass's own segment table, no spill
    ride = [k for k in kernels if k["name"].startswith("_Z19gqe_opt_gemm_kernel")]
    assert len(ride) == 2, [k["name"][:40] for k in ride]
    for k in ride:
        assert k["vgpr"] + k["agpr"] <= 64 and k["scratch"] == 0 and k["vgpr_spill"] == 0 and k["lds"] <= 1024, k
    for name in ("gqe_matstep_kernel", "gqe_retile_kernel"):
        ks = [k for k in kernels if name in k["name"]]
        assert len(ks) == 1 and ks[0]["scratch"] == 0, (name, ks)


def test_no_allocator_code_ahead_of_an_exec_restore(lib):
    """tools/exec_check.py on the assembly the build kept: the placement that corrupted the round-2 variants must not occur in
    any kernel the dispatcher can select (it may in instantiations that are compiled but unreachable — they are listed)."""
    import exec_check
    files = exec_check.default_files()
    assert len(files) >= 13, "the build keeps one device assembly file per translation unit under csrc/obj/ (found %d)" % len(files)
    found = exec_check.check_files(files)
    sel = selectable(lib)
    names = {"gqe_fused_kernel<DEC=%d, MLP=%d, NC=%d, FULL=%d, BWD=%d, FW=%d>" % v for v in sel}
    bad = sorted(set(k for _, k, _, _, _ in found if k in names or not k.startswith("gqe_fused_kernel<")))
    assert not bad, "allocator-inserted code runs before EXEC is restored in: %r" % bad


def test_the_check_recognises_the_round_two_miscompile():
    """The block rocgdb stopped in (d = 80, gqe_fused_kernel<0, 1, 2, FULL=0, BWD=1, FW=16> of commit 8dea5d1^), verbatim."""
    import exec_check
    asm = """
_Z16gqe_fused_kernelILi0ELb1ELi2ELb0ELb1ELi16EEvX:
.LBB6_688:
	s_or_b64 exec, exec, s[0:1]
	scratch_store_dwordx2 off, v[82:83], off offset:140 ; 8-byte Folded Spill
	s_and_saveexec_b64 s[0:1], s[4:5]
	s_cbranch_execz .LBB6_690
; %bb.689:
	global_store_dword v[2:3], v0, off offset:256
.LBB6_690:                              ; %_Z15tile_to_scratchILi2EEvRK7TileEnviPKf.exit1503
	v_writelane_b32 v127, s50, 38
	v_mov_b32_e32 v124, v52
	s_mov_b64 s[12:13], s[94:95]
	scratch_store_dwordx2 off, v[20:21], off offset:220 ; 8-byte Folded Spill
	s_waitcnt vmcnt(4)
	scratch_store_dword off, v9, off offset:152 ; 4-byte Folded Spill
	s_or_b64 exec, exec, s[0:1]
	v_mov_b32_e32 v0, 0
"""
    found = exec_check.check_text(asm)
    assert [f[2] for f in found] == ["v_mov_b32_e32 v124, v52", "scratch_store_dwordx2 off, v[20:21], off offset:220", "scratch_store_dword off, v9, off offset:152"]
    # the same region with the restore where it belongs is clean, and so is a guarded BODY that ends in its own restore
    good = asm.replace("\tv_writelane_b32 v127, s50, 38\n", "\ts_or_b64 exec, exec, s[0:1]\n\tv_writelane_b32 v127, s50, 38\n", 1)
    good = good.replace("\ts_or_b64 exec, exec, s[0:1]\n\tv_mov_b32_e32 v0, 0\n", "\tv_mov_b32_e32 v0, 0\n")
    assert exec_check.check_text(good) == []
