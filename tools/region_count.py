#!/usr/bin/env python3
"""Static instruction counts of the hot regions, each compiled as a kernel of its own
(tools/ubench/region_kernels.hip).  No GPU needed: hipcc -S for gfx950 and a tally per kernel.
Straight-line regions: the static count is the per-call dynamic count.  (k_stage_only is the
common prologue; subtract it.)

    python tools/region_count.py [-D MACRO ...] [--asm out.s]
"""
import argparse, collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tools", "ubench", "region_kernels.hip")
sys.path.insert(0, os.path.join(ROOT, "tools"))
from isa_profile import classify


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-D", action="append", default=[])
    ap.add_argument("--asm", default="")
    args = ap.parse_args()
    with tempfile.TemporaryDirectory() as td:
        asm = args.asm or os.path.join(td, "r.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-S",
                               "--cuda-device-only", "-o", asm, SRC] + ["-D" + d for d in args.D], stderr=subprocess.DEVNULL)
        text = open(asm, errors="replace").read().split("\n")
    cur = None
    counts = collections.OrderedDict()
    for ln in text:
        m = re.match(r"^(k_\w+):", ln)
        if m:
            cur = m.group(1); counts[cur] = collections.Counter(); continue
        s = ln.strip()
        if s.startswith(".Lfunc_end"):
            cur = None
        if cur is None or not s or s.startswith((".", ";")) or s.endswith(":"):
            continue
        op = s.split()[0]
        counts[cur][classify(op)] += 1
        counts[cur]["all"] += 1
        if op.startswith("v_div_fixup"): counts[cur]["div"] += 1
        if op.startswith("v_cndmask"): counts[cur]["cnd"] += 1
        if op.startswith("v_cmp"): counts[cur]["cmp"] += 1
    base = counts.get("k_stage_only", collections.Counter())
    cols = ("all", "valu", "f64", "trans", "salu", "lds", "vmem", "div", "cnd", "cmp")
    print("%-18s" % "kernel (minus prologue)" + "".join("%7s" % c for c in cols))
    for k, c in counts.items():
        uses_stage = k not in ("k_ray_setup", "k_rng", "k_div", "k_sqrt", "k_normalize", "k_stage_only")
        print("%-18s" % k + "".join("%7d" % (c[x] - (base[x] if uses_stage else 0)) for x in cols))


if __name__ == "__main__":
    main()
