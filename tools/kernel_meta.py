#!/usr/bin/env python3
"""Registers / scratch of every gfx950 kernel INSIDE the built libgqe.so, read from the code objects' metadata notes (no
recompilation): the library's .hip_fatbin holds one clang offload bundle per translation unit, each bundle an ELF for
gfx950 whose NT_AMDGPU_METADATA note (msgpack) lists the kernels.
    python tools/kernel_meta.py [path/to/libgqe.so]     ->  name, VGPRs, scratch bytes per lane, spilled VGPRs / SGPRs"""
import os, re, struct, sys

MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _code_objects(blob):
    for m in re.finditer(re.escape(MAGIC), blob):
        p = m.start()
        (n,) = struct.unpack_from("<Q", blob, p + 24)
        q = p + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, q)
            q += 24
            triple = blob[q:q + tl].decode()
            q += tl
            if "gfx950" in triple and size:
                yield blob[p + off:p + off + size]


def _notes(elf):
    """(type, desc) of every note in the ELF64's SHT_NOTE sections."""
    assert elf[:4] == b"\x7fELF" and elf[4] == 2
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum = struct.unpack_from("<HH", elf, 0x3A)
    for i in range(shnum):
        sh = shoff + i * shentsize
        sh_type, = struct.unpack_from("<I", elf, sh + 4)
        if sh_type != 7:
            continue
        off, size = struct.unpack_from("<QQ", elf, sh + 0x18)
        p, end = off, off + size
        while p + 12 <= end:
            namesz, descsz, ntype = struct.unpack_from("<III", elf, p)
            p += 12
            p += (namesz + 3) & ~3
            yield ntype, elf[p:p + descsz]
            p += (descsz + 3) & ~3


def kernel_metadata(path):
    import msgpack
    out = []
    blob = open(path, "rb").read()
    for co in _code_objects(blob):
        for ntype, desc in _notes(co):
            if ntype != 32:   # NT_AMDGPU_METADATA
                continue
            meta = msgpack.unpackb(desc, raw=False, strict_map_key=False)
            for k in meta.get("amdhsa.kernels", []):
                out.append({"name": k[".name"], "vgpr": k.get(".vgpr_count"), "agpr": k.get(".agpr_count", 0),
                            "sgpr": k.get(".sgpr_count"), "scratch": k.get(".private_segment_fixed_size", 0),
                            "vgpr_spill": k.get(".vgpr_spill_count", 0), "sgpr_spill": k.get(".sgpr_spill_count", 0),
                            "lds": k.get(".group_segment_fixed_size", 0)})
    return out


def fused_variant(name):
    """(DEC, MLP, NC, FULL, BWD, FW) of a mangled gqe_fused_kernel name, or None.  (The LEAN instantiations — the seventh
    template argument, fused_lean — are variants of the same dispatcher choice: the launcher picks them by what the launch
    carries, not by the configuration.)"""
    m = re.match(r"_Z16gqe_fused_kernelILi(\d)ELb(\d)ELi(\d)ELb(\d)ELb(\d)ELi(\d+)EL[bi]\dEE", name)
    return tuple(int(x) for x in m.groups()) if m else None


def fused_lean(name):
    """The LEAN template argument of a mangled gqe_fused_kernel name — 0 plain, 1 lean, 2 lean row-sharded (None: not one)."""
    m = re.match(r"_Z16gqe_fused_kernelILi\dELb\dELi\dELb\dELb\dELi\d+EL[bi](\d)EE", name)
    return int(m.group(1)) if m else None


if __name__ == "__main__":
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "graphqembed_amd", "libgqe.so")
    ks = kernel_metadata(path)
    print("%-78s %5s %7s %6s %6s" % ("kernel", "VGPR", "scratch", "vspill", "sspill"))
    for k in sorted(ks, key=lambda k: k["name"]):
        v = fused_variant(k["name"])
        label = ("gqe_fused_kernel<DEC=%d, MLP=%d, NC=%d, FULL=%d, BWD=%d, FW=%d%s>" % (v + (", LEAN=%d" % fused_lean(k["name"]) if fused_lean(k["name"]) else "",))) if v else k["name"][:78]
        print("%-78s %5s %7s %6s %6s" % (label, k["vgpr"], k["scratch"], k["vgpr_spill"], k["sgpr_spill"]))
