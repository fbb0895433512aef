#!/usr/bin/env python3
"""Static check of the compiler's gfx950 assembly for the miscompile behind round 2's "spilling kernels corrupt results"
finding (root-caused in round 3 with rocgdb on a faulting wave):

    .LBB6_688:
        s_or_b64 exec, exec, s[0:1]
        s_and_saveexec_b64 s[0:1], s[4:5]        ; exec := the lanes of a guarded store (j < d: lanes 0..15 at d = 80)
        s_cbranch_execz .LBB6_690
    ; %bb.689:
        global_store_dword v[2:3], v0, off offset:256
    .LBB6_690:                                   ; JOIN block of the guarded region
        v_writelane_b32 v127, s50, 38
        v_mov_b32_e32 v124, v52                  ; <- live-range-split COPY inserted by the register allocator ...
        scratch_store_dwordx2 off, v[20:21], ... ; <- ... and VGPR SPILL stores ...
        s_or_b64 exec, exec, s[0:1]              ; <- ... AHEAD of the instruction that restores EXEC

The copy and the spill stores run with the narrowed EXEC of the region that has just ended, so lanes that were inactive
there (lanes >= 16) keep stale register / stack contents: v124 (lane * 4, an address term) held floats in lanes 16..63 and
the next full-EXEC store faulted; a spilled gradient partial lost its lanes >= 16.  The allocator only needs to WANT a
split or a spill at such a join — register pressure — so the pattern shows up in the guarded (d % 64 != 0) kernels that sit
at their VGPR limit; source-level undefined behaviour is not involved.

The check: in every join block of every kernel (targets of `s_cbranch_execz` skip branches, and the `; %bb.N:` fall-through
blocks left when a skip branch is removed), no spill store / reload and no VGPR-to-VGPR copy may precede the block's first
`s_or_b64 exec, exec, s[..]`.

    python tools/exec_check.py [file.s ...]      default: every *gfx950*.s under graphqembed_amd/csrc/obj (make keeps them)
Exit code 1 and one line per finding if a kernel is affected."""
import glob
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BLOCK = re.compile(r"^(\.LBB\d+_\d+:|; %bb\.\d+:)")
FUNC = re.compile(r"^([A-Za-z_][\w$.]*):\s*(;.*)?$")
END_CF = re.compile(r"^s_or_b64 exec, exec, (s\[\d+:\d+\]|vcc)")
EXEC_FREE = re.compile(r"^(s_|v_readlane_b32|v_writelane_b32|;|\.)")   # SALU / SMEM / waitcnt, the SGPR <-> VGPR-lane moves, comments, directives


SKIP = re.compile(r"^s_cbranch_execz (\.LBB\d+_\d+)")
ARTIFACT = re.compile(r"Folded Spill|Folded Reload")
COPY = re.compile(r"^v_(?:mov_b32_e32|mov_b64_e32|accvgpr_write_b32|accvgpr_read_b32) (?:v|a)(?:\d+|\[\d+:\d+\]), (?:v|a)(?:\d+|\[\d+:\d+\])\s*$")


def check_text(text, name="<asm>"):
    """[(function, block label, offending instruction, line number)] for one assembly file.

    JOIN blocks are the targets of the `s_cbranch_execz` skip branches (and the `; %bb.N:` fall-through blocks that remain when
    the skip branch of a short region is removed late).  In such a block, ahead of its first `s_or_b64 exec, exec, sX`, the
    register allocator's own instructions — spill stores / reloads (the assembly marks them `Folded Spill` / `Folded Reload`)
    and plain VGPR-to-VGPR copies (live-range splits) — are findings: they would run under the narrowed EXEC of the region
    that has just ended.  (Other vector instructions in front of a restore belong to region bodies that tail duplication
    merged with their restore; they are meant to run narrowed.)"""
    lines = text.splitlines()
    findings = []
    # pass 1: skip-branch targets per function
    targets, func = {}, None
    for raw in lines:
        m = FUNC.match(raw)
        if m and not raw.startswith((".L", ";")):
            func = m.group(1)
            targets[func] = set()
            continue
        mk = SKIP.match(raw.strip())
        if mk and func is not None:
            targets[func].add(mk.group(1))
    func, block, lead, in_lead, is_join = None, None, [], False, False
    prev = ""                  # the previous instruction (a fall-through block right behind a saveexec / its skip branch is the BODY)
    for ln, raw in enumerate(lines, 1):
        line = raw.strip()
        if not line:
            continue
        m = FUNC.match(raw)
        if m and not raw.startswith((".L", ";")):
            func, block, in_lead = m.group(1), "entry", False
            continue
        if BLOCK.match(raw):
            block = line.split(":")[0]
            body = prev.startswith(("s_cbranch_exec", "s_and_saveexec", "s_or_saveexec", "s_andn2_saveexec", "s_xor_saveexec"))
            is_join = block in targets.get(func, ()) or (block.startswith("; %bb") and not body)
            lead, in_lead = [], True
            continue
        if not line.startswith((";", ".")):
            prev = line
        if func is None or not in_lead:
            continue
        if END_CF.match(line):
            if is_join:
                for ins, l0 in lead:
                    findings.append((func, block, ins, l0))
            in_lead = False
            continue
        if EXEC_FREE.match(line):
            continue
        code = line.split(";")[0].strip()
        if ARTIFACT.search(line) or COPY.match(code):
            lead.append((code, ln))
        if len(lead) > 64:
            in_lead = False
    return findings


def kernel_of(func):
    m = re.match(r"_Z16gqe_fused_kernelILi(\d)ELb(\d)ELi(\d)ELb(\d)ELb(\d)ELi(\d+)EE", func)
    return ("gqe_fused_kernel<DEC=%s, MLP=%s, NC=%s, FULL=%s, BWD=%s, FW=%s>" % m.groups()) if m else func[:60]


def check_files(paths):
    out = []
    for p in paths:
        with open(p, errors="replace") as f:
            for (func, block, ins, ln) in check_text(f.read(), p):
                out.append((p, kernel_of(func), block, ins, ln))
    return out


def default_files():
    return sorted(glob.glob(os.path.join(ROOT, "graphqembed_amd", "csrc", "obj", "**", "*gfx950*.s"), recursive=True))


if __name__ == "__main__":
    files = sys.argv[1:] or default_files()
    if not files:
        raise SystemExit("no assembly found: build first (the Makefile keeps the device assembly under csrc/obj/)")
    found = check_files(files)
    kernels = sorted(set((p, k) for p, k, _, _, _ in found))
    for p, k, b, ins, ln in found[:200]:
        print("%s:%d  %s  block %s: `%s` runs before EXEC is restored" % (os.path.relpath(p, ROOT), ln, k, b, ins))
    print("%d file(s), %d finding(s) in %d kernel(s)" % (len(files), len(found), len(kernels)))
    sys.exit(1 if found else 0)
