"""The no-GPU analysis tools keep working on the tree as it is (they parse the kernel source by text and the disassembly by inline chains:
an edit of csrc/ssx_kernels.hip can break them silently)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_isa_census_prices_the_path_kernel_and_agrees_with_the_counters():
    """tools/isa_census.py: compiles the path kernel with line tables, attributes every instruction to a region through its inline chain,
    weights with the committed lane-occupancy counters and prices with the committed issue rates.  Its dynamic VALU instruction count has
    to stay within 4 % of what the hardware counted (profiles/r05/pmc_summary.csv: SQ_INSTS_VALU per launch / loop iterations)."""
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        import pytest
        pytest.skip("no hipcc")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_census.py"), "--rates", os.path.join(ROOT, "profiles", "r05", "valu_rates.log"),
                          "--lanestat", os.path.join(ROOT, "profiles", "r05", "lanestat.log")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    m = re.search(r"model: (\d+) VALU wave-instructions and (\d+) issue cycles per loop iteration", out.stdout)
    p = re.search(r"pair-aware model .*?: (\d+) cycles per iteration", out.stdout)
    assert m and p, out.stdout[:2000]
    dyn, single, pair = int(m.group(1)), int(m.group(2)), int(p.group(1))
    valu = None
    for ln in open(os.path.join(ROOT, "profiles", "r05", "pmc_summary.csv")):
        if ln.startswith("ssx_render_kernel_cornell,SQ_INSTS_VALU,"):
            valu = float(ln.strip().split(",")[-1])
    iters = None
    for ln in open(os.path.join(ROOT, "profiles", "r05", "lanestat.log")):
        if ln.startswith("iteration: lanes with a path"):
            iters = float(ln.split()[5]) * 4.0            # the counters are of a 64-spp render, the PMC run of the 256-spp bench workload
    measured = valu / iters
    assert abs(dyn - measured) / measured < 0.04, (dyn, measured)
    assert 0.7 * single < pair < single                   # the pair-aware price is below the sum of the single prices, and not by an order of magnitude
    for region in ("trace primary: pass 1", "trace primary: pass 2 trip", "light: sphtri_make acos/sin (binary64)", "fold: level step (2 ways)", "loop: refill"):
        assert region in out.stdout, region
