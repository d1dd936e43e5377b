#!/usr/bin/env python3
"""Static instruction attribution for a kernel of libssx_hip.so (no GPU needed).

Compiles csrc/ssx_api.hip to gfx950 assembly with line info (-g -S), walks one kernel's body and
attributes every instruction to the source function whose line range contains its .loc line
(inlined callees keep their own lines, so ssx_fmath.h work shows up under ssx_fm_* / ssx_acosf ...).
Prints instruction counts per function and class (VALU / transcendental / f64 / SALU / LDS / VMEM),
and the static count of IEEE division / sqrt expansions.  Static counts are not dynamic counts: the
quad loop of pass 1 runs n_quads times per trace, pass 2 once per candidate -- the table is for
comparing builds (did this change remove instructions from region X?), not for timing.

    python tools/isa_profile.py [--kernel ssx_render_kernel] [--lines N] [-D MACRO ...]
"""
import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "simple_spectral_amd", "csrc", "ssx_api.hip")

FUNC_RE = re.compile(r"^\s*(?:template\s*<[^>]*>\s*)?(?:static\s+|extern\s+\"C\"\s+|SSX_FM_FN\s+|__device__\s+|__global__\s+|__forceinline__\s+|inline\s+|__host__\s+)*"
                     r"[A-Za-z_][\w:<>\*&\s]*?\b([A-Za-z_]\w*)\s*\([^;{}]*\)\s*(?:const\s*)?\{\s*(?://.*)?$")


def function_ranges(path):
    """[(first_line, last_line, name)] of top-level-ish function bodies, by brace matching."""
    out = []
    try:
        lines = open(path, errors="replace").read().split("\n")
    except OSError:
        return out
    i = 0
    while i < len(lines):
        m = FUNC_RE.match(lines[i])
        if m and m.group(1) not in ("if", "for", "while", "switch", "return", "sizeof"):
            depth = 0
            j = i
            started = False
            while j < len(lines):
                code = re.sub(r"//.*", "", lines[j])
                depth += code.count("{") - code.count("}")
                if "{" in code:
                    started = True
                if started and depth <= 0:
                    break
                j += 1
            out.append((i + 1, j + 1, m.group(1)))
            i = j + 1
        else:
            i += 1
    return out


def classify(op):
    if op.startswith(("v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log", "v_sin", "v_cos")):
        return "trans"
    if op.startswith("v_") and ("_f64" in op):
        return "f64"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "vmem"
    return "other"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", default="ssx_render_kernel")
    ap.add_argument("--lines", type=int, default=0, help="also print the N hottest source lines")
    ap.add_argument("-D", action="append", default=[])
    ap.add_argument("--keep", default="", help="write the assembly here")
    args = ap.parse_args()

    hipcc = "/opt/rocm/bin/hipcc"
    with tempfile.TemporaryDirectory() as td:
        asm = args.keep or os.path.join(td, "k.s")
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-g", "-S", "--cuda-device-only",
               "-o", asm, SRC] + ["-D" + d for d in args.D]
        subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
        text = open(asm, errors="replace").read().split("\n")

    files = {}
    for ln in text:
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', ln)
        if m:
            files[int(m.group(1))] = os.path.join(m.group(2), m.group(3)) if m.group(3) else m.group(2)
    ranges = {}

    def func_of(fid, line):
        path = files.get(fid, "?")
        if path not in ranges:
            ranges[path] = function_ranges(path if os.path.isabs(path) else os.path.join(ROOT, path))
        best = None
        for a, b, name in ranges[path]:
            if a <= line <= b and (best is None or a >= best[0]):
                best = (a, b, name)
        return best[2] if best else os.path.basename(path) + ":?"

    start = None
    for i, ln in enumerate(text):
        if ln.startswith(args.kernel + ":"):
            start = i
            break
    if start is None:
        sys.exit("kernel %s not found" % args.kernel)
    per_func = collections.defaultdict(collections.Counter)
    per_line = collections.Counter()
    cur = (0, 0)
    total = collections.Counter()
    for ln in text[start + 1:]:
        s = ln.strip()
        if s.startswith(".Lfunc_end"):
            break
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", s)
        if m:
            cur = (int(m.group(1)), int(m.group(2)))
            continue
        if not s or s.startswith((".", ";")) or s.endswith(":"):
            continue
        op = s.split()[0]
        c = classify(op)
        f = func_of(*cur)
        per_func[f][c] += 1
        per_func[f]["all"] += 1
        if op.startswith("v_div_fixup"):
            per_func[f]["div"] += 1
            total["div"] += 1
        if op.startswith(("v_cndmask", "v_cmp", "v_min", "v_max", "v_med3")):   # the 4.3-cycle select/compare kind
            per_func[f]["sel"] += 1
            total["sel"] += 1
        total[c] += 1
        total["all"] += 1
        per_line[(os.path.basename(files.get(cur[0], "?")), cur[1])] += 1
    cols = ("all", "valu", "f64", "trans", "salu", "lds", "vmem", "div", "sel")
    print("%-28s" % args.kernel + "".join("%8s" % c for c in cols))
    for f, cnt in sorted(per_func.items(), key=lambda kv: -kv[1]["all"]):
        print("%-28s" % f[:28] + "".join("%8d" % cnt[c] for c in cols))
    print("%-28s" % "TOTAL" + "".join("%8d" % total[c] for c in cols))
    if args.lines:
        print("\nhottest source lines (static instructions):")
        for (f, l), n in per_line.most_common(args.lines):
            print("  %-18s %5d  %d" % (f, l, n))


if __name__ == "__main__":
    main()
