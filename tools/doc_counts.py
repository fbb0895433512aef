#!/usr/bin/env python3
"""The counts DESIGN.md quotes, generated instead of typed: tests per marker (pytest --collect-only), golden fixtures, C-ABI symbols.
    python tools/doc_counts.py            # print the block
    python tools/doc_counts.py --write    # rewrite the block between <!-- counts:begin --> and <!-- counts:end --> in DESIGN.md"""
import glob, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def collected(marker):
    out = subprocess.run([sys.executable, "-m", "pytest", "tests", "--collect-only", "-q", "-m", marker], cwd=ROOT, capture_output=True, text=True).stdout
    m = re.search(r"(\d+)(?:/\d+)? tests? collected|(\d+) selected", out) or re.search(r"(\d+) tests? collected", out)
    n = [l for l in out.splitlines() if "::" in l]
    return len(n)


def main():
    gpu, cpu = collected("gpu"), collected("not gpu")
    fixtures = len(glob.glob(os.path.join(ROOT, "tests", "golden", "*.npz")))
    header = open(os.path.join(ROOT, "include", "gqe.h")).read() + open(os.path.join(ROOT, "include", "gqe_sampler.h")).read()
    symbols = len(set(re.findall(r"^\s*(?:int|int64_t|const char\*|void)\s+(gqe_\w+)\s*\(", header, re.M)))
    block = ("<!-- counts:begin (tools/doc_counts.py --write) -->\n"
             "%d `-m gpu` tests (parity through the C ABI on the MI355X) + %d `-m \"not gpu\"` tests (oracle vs golden fixtures, host logic, "
             "library symbols, 2–4-rank gloo); %d reference-generated `.npz` fixtures under `tests/golden/`; %d exported `gqe_*` entry points "
             "declared in `include/`.\n<!-- counts:end -->" % (gpu, cpu, fixtures, symbols))
    if "--write" in sys.argv:
        p = os.path.join(ROOT, "DESIGN.md")
        s = open(p).read()
        s2 = re.sub(r"<!-- counts:begin.*?<!-- counts:end -->", lambda m: block, s, flags=re.S)
        if s2 == s and "counts:begin" not in s:
            raise SystemExit("DESIGN.md has no counts block")
        open(p, "w").write(s2)
    print(block)


if __name__ == "__main__":
    main()
