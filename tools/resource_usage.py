#!/usr/bin/env python3
"""Per-kernel register / scratch / occupancy table of the gfx950 build (hipcc -Rpass-analysis=kernel-resource-usage).
    python tools/resource_usage.py [DEC MLP] [--all]   # fused variants of one (decoder, MLP) pair (default 0 1) [+ gqe_kernels.hip]
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "graphqembed_amd", "csrc")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-munsafe-fp-atomics", "-Rpass-analysis=kernel-resource-usage"]


def report(src, defs):
    p = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + defs + ["-c", src, "-o", "/dev/null"], cwd=CSRC, stderr=subprocess.PIPE,
                       stdout=subprocess.PIPE, universal_newlines=True)
    rows, cur = [], None
    for line in p.stderr.splitlines():
        m = re.search(r"remark:\s+(?:Function )?Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}
            rows.append(cur)
            continue
        m = re.search(r"remark:\s+([A-Za-z /\[\]]+): (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    return rows


def demangle(name):
    try:
        return subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], stdout=subprocess.PIPE, universal_newlines=True).stdout.strip()
    except OSError:
        return name


def main():
    dec, mlp = (sys.argv[1], sys.argv[2]) if len(sys.argv) > 2 else ("0", "1")
    rows = []
    for fw in ("16", "8"):
        rows += report("gqe_fused_inst.hip", ["-DGQE_DEC=" + dec, "-DGQE_MLP=" + mlp, "-DGQE_FW=" + fw])
    if "--all" in sys.argv:
        rows += report("gqe_kernels.hip", [])
    print("%-78s %5s %5s %8s %5s %7s %7s" % ("kernel", "VGPR", "AGPR", "scratch", "occ", "vspill", "sspill"))
    for r in rows:
        nm = re.sub(r"\(.*", "", demangle(r["name"])).replace("void ", "")
        print("%-78s %5d %5d %8d %5d %7d %7d" % (nm[:78], r.get("VGPRs", -1), r.get("AGPRs", -1), r.get("ScratchSize [bytes/lane]", -1),
                                              r.get("Occupancy [waves/SIMD]", -1), r.get("VGPRs Spill", -1), r.get("SGPRs Spill", -1)))


if __name__ == "__main__":
    main()
