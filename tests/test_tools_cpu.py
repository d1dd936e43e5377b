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


def test_bench_helpers_replay_stamp_and_n1_reference(tmp_path, monkeypatch):
    """bench.py's bookkeeping that needs no GPU: a replayed traffic figure carries the hash of the kernel sources it was taken on and is
    marked stale when they differ from the tree's; the N = 1 reference of the efficiency field names its source."""
    import json
    import types
    sys.path.insert(0, ROOT)
    import bench
    src_id = bench.kernel_source_id()
    assert len(src_id) == 16 and src_id == bench.kernel_source_id()
    key = "cornell-srgb 512 spp256 obs1931 gpus2"
    tj = tmp_path / "traffic.json"
    args = types.SimpleNamespace(scene="cornell-srgb", res=512, spp=256, observer=1931, no_pmc=False)
    monkeypatch.setenv("SSX_BENCH_TRAFFIC_JSON", str(tj))
    # nothing recorded for this key: no figure, but still a stamp saying why nothing was measured
    tj.write_text(json.dumps({}))
    traffic, detail, source = bench.measured_traffic(args, 2)
    assert traffic is None and "REPLAYED" in source and detail["replayed"]["why_not_measured"].startswith("N > 1")
    # recorded on these sources: not stale; recorded on others: stale
    tj.write_text(json.dumps({key: 123, key + " detail": {"kernel_source_id": src_id}}))
    traffic, detail, source = bench.measured_traffic(args, 2)
    assert traffic == 123 and detail["replayed"]["stale"] is False and detail["replayed"]["this_build_kernel_source_id"] == src_id
    tj.write_text(json.dumps({key: 123, key + " detail": {"kernel_source_id": "0" * 16}}))
    assert bench.measured_traffic(args, 2)[1]["replayed"]["stale"] is True
    tj.write_text(json.dumps({key: 123, key + " detail": {}}))
    assert "unknown" in str(bench.measured_traffic(args, 2)[1]["replayed"]["stale"])
    # the N = 1 reference of the efficiency field: only a figure taken on THIS build's kernel sources counts
    monkeypatch.delenv("SSX_BENCH_N1_VALUE", raising=False)
    n1 = bench.n1_reference("f" * 16)                      # sources nobody has measured: refused, with the newest other figure named
    assert n1["value"] is None and "no N = 1 figure" in n1["refused"] and "newest other" in n1["refused"]
    rec = tmp_path / "BENCH_r99.json"
    rec.write_text(json.dumps({"parsed": {"n_gpus": 1, "value": 3210.5, "config": {"workload": "cornell-srgb 512x512 spp=256/GPU (total spp 256)", "kernel_source_id": "f" * 16}}}))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    n1 = bench.n1_reference("f" * 16)
    assert n1["value"] == 3210.5 and "BENCH_r99.json" in n1["source"] and not n1.get("refused")
    assert bench.n1_reference("e" * 16)["value"] is None   # the same record is stale for other sources
    monkeypatch.setenv("SSX_BENCH_N1_VALUE", "3000")
    assert bench.n1_reference("e" * 16)["value"] == 3000.0
    monkeypatch.delenv("SSX_BENCH_N1_VALUE")
    monkeypatch.setattr(bench, "ROOT", ROOT)
    # the design's byte model the line quotes next to the counters
    assert 440 < bench.design_bytes_per_sample("cornell-srgb", levels=4.07) < 500      # 470.8: the counters read 1.23 x the path kernel's share of it
    assert abs(bench.algorithmic_bytes_per_sample_8d("cornell-srgb", 256) - (16.0 / 256 + 3.8)) < 1e-9


def test_runtime_preload_reads_sonames_from_the_elf_files():
    """simple_spectral_amd/_capi.py maps torch's bundled HIP runtime ahead of libssx_hip.so only when its SONAME is the one the library
    needs: the ELF dynamic sections it reads that from."""
    sys.path.insert(0, ROOT)
    from simple_spectral_amd import _capi, build
    if not os.path.exists(build.HIP_LIB):
        import pytest
        pytest.skip("library not built")
    soname, needed = _capi._elf_dynamic(build.HIP_LIB)
    assert any(n.startswith("libamdhip64.so") for n in needed) and "libstdc++.so.6" in needed
    soname, needed = _capi._elf_dynamic(build.HOST_LIB)
    assert "libz.so.1" in needed or any(n.startswith("libz") for n in needed)
    assert _capi._elf_dynamic(__file__) == (None, [])          # not an ELF file
    assert _capi._elf_dynamic("/nonexistent") == (None, [])
