"""bench.py's N>1 branch end to end: `python bench.py --gpus 2` run plainly spawns
its two ranks itself, each renders its tile share, the framebuffers are summed on rank 0, one JSON line
comes out.  SSX_BENCH_TEST_ONE_GPU=1 lets both ranks share device 0 (reduce over gloo), so this runs on
a 1-GPU box; on the driver's 8-GPU node the same code path uses one GPU per rank and RCCL."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle_lib as ol

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("launcher,world", [("self-spawn", 2), ("torch.distributed.run", 2), ("self-spawn", 8)])
def test_bench_two_ranks_json_line_and_image(tmp_path, launcher, world):
    """(world 8: the driver's largest run in miniature -- eight ranks, tile rows rotated, every rank's record in the line)"""
    dump = str(tmp_path / "img.npy")
    env = dict(os.environ, SSX_BENCH_TEST_ONE_GPU="1", SSX_BENCH_DUMP=dump)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    args = ["--gpus", str(world), "--steps", "2", "--warmup", "1", "--res", "64", "--spp", "4", "--texture", "test-img.png", "--no-cpu-baseline"]
    if launcher == "self-spawn":
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1", "--master-port", "29653",
               os.path.join(ROOT, "bench.py")] + args
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == world and line["scaling"] == "weak" and line["unit"] == "Msamples/s" and line["value"] > 0
    assert line["steps"] == 2 and line["warmup"] == 1 and "roofline" in line and "cpu_baseline" not in line
    assert line["config"]["workload"].startswith("cornell-srgb 64x64 spp=4/GPU (total spp %d)" % (4 * world))
    # the line ties its value to a checked image and says who took part
    assert line["check"]["differing_floats"] == 0 and line["check"]["tiles"] >= 4 and line["check"]["spp"] == 4 * world
    assert [x["rank"] for x in line["ranks"]] == list(range(world)) and all(x["ms_per_step"] > 0 and x["path_ms"] > 0 and x["tiles_owned"] == 64 // world for x in line["ranks"])
    assert len({x["pid"] for x in line["ranks"]}) == world and all(0.0 <= x["units_parked_frac"] <= 1.0 for x in line["ranks"])
    d = line["distributed"]
    assert d["evidence_error"] is None
    assert d["backend"] == "gloo" and d["world_size"] == world and d["overlap"]["pixels_nonzero_on_more_than_one_rank"] == 0 and d["overlap"]["pixels_nonzero_on_some_rank"] > 0
    assert d["devices_distinct"] is False            # both ranks on device 0 here (SSX_BENCH_TEST_ONE_GPU): the driver's 8-GPU run must say True, or bench.py refuses
    # the efficiency field divides only by an N = 1 figure taken on THIS build's kernel sources; anything else in the tree is refused by name
    e = line["efficiency_vs_n1_reference"]
    if e["n1_value"] is None:
        assert e["value"] is None and "no N = 1 figure taken on this build's kernel sources" in e["refused"]
    else:
        assert abs(e["value"] - line["value"] / (world * e["n1_value"])) < 1e-3 and e["kernel_source_id"] == line["config"]["kernel_source_id"]
    assert all(x["device_scratch_total"] > 0 and x["device_scratch_bytes"]["log_bytes"] > 0 for x in line["ranks"])
    assert "REPLAYED" in (line["roofline"]["traffic_source"] or "REPLAYED") or line["roofline"]["traffic"] is None
    img = np.load(dump)
    ref = ol.Oracle("cornell-srgb", texture="test-img.png").render(64, 64, 4 * world)   # total spp = 4 per GPU x world
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))


@pytest.mark.gpu
def test_bench_refuses_a_line_when_two_ranks_share_a_device(tmp_path):
    """Without SSX_BENCH_TEST_ONE_GPU a 2-rank run on a 1-GPU box puts both ranks on device 0 over RCCL: either RCCL refuses (duplicate
    device) or bench.py's own device check does -- in no case is a bench line printed."""
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "SSX_BENCH_TEST_ONE_GPU", "SSX_BENCH_FORCE_DIST"):
        env.pop(k, None)
    env["NCCL_DEBUG"] = "WARN"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--res", "64", "--spp", "4", "--texture", "test-img.png", "--no-cpu-baseline"]
    try:
        r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=240)
    except subprocess.TimeoutExpired:
        pytest.skip("RCCL hung on the duplicate device instead of refusing it")
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


@pytest.mark.gpu
def test_bench_one_rank_through_rccl(tmp_path):
    """the RCCL branch executed for real on the 1-GPU box -- `--gpus 1` with
    SSX_BENCH_FORCE_DIST=1 initialises the nccl (= RCCL) process group with world size 1 on this device and
    runs the framebuffer reduce on the DEVICE buffer inside the timed loop, next to libssx_hip.so in one
    process (one HIP runtime: simple_spectral_amd/_capi.py).  The combined image equals the oracle's."""
    dump = str(tmp_path / "img.npy")
    env = dict(os.environ, SSX_BENCH_FORCE_DIST="1", SSX_BENCH_DUMP=dump)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "SSX_BENCH_TEST_ONE_GPU"):
        env.pop(k, None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--res", "64", "--spp", "8",
           "--texture", "test-img.png", "--no-cpu-baseline", "--no-pmc"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["check"]["differing_floats"] == 0 and line["distributed"]["backend"] == "nccl" and line["distributed"]["world_size"] == 1
    assert len(line["ranks"]) == 1 and line["ranks"][0]["device_uuid"] != "" and line["distributed"]["devices_distinct"] is True
    assert line["n_gpus"] == 1 and "RCCL reduce (world size 1" in line["config"]["parallelism"]
    assert line["value"] > 0 and line["value_device_resident"] > 0 and line["ms_per_step_device_resident"] > 0 and line["value_host_inclusive"] == line["value"]
    img = np.load(dump)
    ref = ol.Oracle("cornell-srgb", texture="test-img.png").render(64, 64, 8)
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))


@pytest.mark.gpu
def test_bench_measures_its_hbm_traffic_in_the_run(tmp_path):
    """with rocprofv3 on the box the default N = 1 run measures roofline.traffic itself (two --pmc passes over a child
    run) instead of replaying a file; the line says which, and carries the generate kernel's bytes too."""
    import shutil
    if not (shutil.which("rocprofv3") or os.path.exists("/opt/rocm/bin/rocprofv3")):
        pytest.skip("no rocprofv3")
    tj = str(tmp_path / "traffic.json")
    env = dict(os.environ, SSX_BENCH_TRAFFIC_JSON=tj)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "SSX_BENCH_TEST_ONE_GPU", "SSX_BENCH_FORCE_DIST", "SSX_BENCH_NO_PMC"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--res", "256", "--spp", "32", "--no-cpu-baseline", "--update-traffic"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    rf = line["roofline"]
    if "REPLAYED" in (rf["traffic_source"] or "REPLAYED"):
        pytest.skip("counter passes unavailable on this box: %s" % ((rf.get("traffic_detail") or {}).get("replayed") or {}).get("why_not_measured"))
    assert rf["traffic_source"].startswith("measured in this run")
    samples = 256 * 256 * 32
    per_sample = rf["traffic"] / samples
    assert 300.0 < per_sample < 900.0, per_sample                 # the design's own bytes are ~425 B per sample: counters within 2x of them
    d = rf["traffic_detail"]
    assert d["bytes_per_launch"]["generate"] > 48 * samples * 0.8 and d["launches"]["path"] >= 3 and d["renders"] == d["launches"]["path"] and d["kernel_source_id"]
    assert json.load(open(tj))["cornell-srgb 256 spp32 obs1931 gpus1"] == rf["traffic"]
    assert line["check"]["differing_floats"] == 0


@pytest.mark.gpu
@pytest.mark.parametrize("world", [1, 2])
def test_bench_dry_run_reports_the_multi_gpu_plumbing(world):
    """`bench.py --dist-dry-run` goes through the process group, a timed framebuffer reduce, every rank's share of the
    workload at 1/16 of the samples and the C++ host's RCCL combine as a probe (ssx_rccl_probe: ncclCommInitAll over the visible devices), and
    REPORTS what it found in one JSON line -- here on one GPU (world 2: both ranks on device 0, gloo)."""
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "SSX_BENCH_TEST_ONE_GPU", "SSX_BENCH_FORCE_DIST"):
        env.pop(k, None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if world == 1:
        env["SSX_BENCH_FORCE_DIST"] = "1"      # RCCL with world size 1
    else:
        env["SSX_BENCH_TEST_ONE_GPU"] = "1"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--dist-dry-run", "--res", "128", "--spp", "32", "--texture", "test-img.png"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["dry_run"] is True and d["n_gpus"] == world and len(d["ranks"]) == world
    assert d["collectives"]["all_reduce_ones"] == float(world) and len(d["collectives"]["framebuffer_reduce_ms"]) == 3
    for x in d["ranks"]:
        assert "device_error" not in x and "render_error" not in x, x
        assert x["free_bytes"] > 0 and 0 < x["share_pixels_nonzero"] <= 128 * 128 // world and x["share_render_ms"] > 0   # (alpha is 0 where a camera ray leaves the open box)
        assert x["sample_bytes_needed"] > 0 and x["scratch_after_upload"]["log_bytes"] > 0
    p = d["rccl_probe_single_process"]
    assert p["visible_devices"] >= 1 and p["returned"] == p["devices_reduced_ok"] == p["visible_devices"] and p["rccl"] == "ok" and p["reduce_elements_wrong"] == 0
