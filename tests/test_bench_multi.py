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
@pytest.mark.parametrize("launcher", ["self-spawn", "torch.distributed.run"])
def test_bench_two_ranks_json_line_and_image(tmp_path, launcher):
    dump = str(tmp_path / "img.npy")
    env = dict(os.environ, SSX_BENCH_TEST_ONE_GPU="1", SSX_BENCH_DUMP=dump)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    args = ["--gpus", "2", "--steps", "2", "--warmup", "1", "--res", "64", "--spp", "4", "--texture", "test-img.png", "--no-cpu-baseline"]
    if launcher == "self-spawn":
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29653",
               os.path.join(ROOT, "bench.py")] + args
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["unit"] == "Msamples/s" and line["value"] > 0
    assert line["steps"] == 2 and line["warmup"] == 1 and "roofline" in line and "cpu_baseline" not in line
    assert line["config"]["workload"].startswith("cornell-srgb 64x64 spp=4/GPU (total spp 8)")
    img = np.load(dump)
    ref = ol.Oracle("cornell-srgb", texture="test-img.png").render(64, 64, 8)   # total spp = 4 per GPU x 2
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))


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
           "--texture", "test-img.png", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and "RCCL reduce (world size 1" in line["config"]["parallelism"]
    assert line["value"] > 0 and line["value_device_resident"] > 0 and line["ms_per_step_device_resident"] > 0 and line["value_host_inclusive"] == line["value"]
    img = np.load(dump)
    ref = ol.Oracle("cornell-srgb", texture="test-img.png").render(64, 64, 8)
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))
