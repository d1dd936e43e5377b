"""The C++ host: CLI argument behaviour (CPU) and an end-to-end render through the binary (GPU)."""
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as ol
from simple_spectral_amd import build as sbuild

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "simple-spectral")


@pytest.fixture(scope="module")
def cli():
    sbuild.build_host()
    assert os.path.exists(CLI)
    return CLI


def run(cli, *args):
    return subprocess.run([cli] + list(args), cwd=ROOT, capture_output=True, text=True)


def test_cli_argument_errors_mirror_the_reference(cli):
    r = run(cli)
    assert r.returncode == 255 and "Required argument `--scene`/`-s` not found!" in r.stderr and "Simple Spectral" in r.stdout
    r = run(cli, "--scene=nope", "-w=8", "-h=8", "-spp=1", "-o=/tmp/x.png")
    assert r.returncode == 255 and 'Unrecognized scene "nope"' in r.stderr  # src/main.cpp:92-101
    r = run(cli, "-s=cornell", "-w=0", "-h=8", "-spp=1", "-o=/tmp/x.png")
    assert r.returncode == 255 and "Invalid width or height!" in r.stderr   # src/main.cpp:106-112
    r = run(cli, "-s=cornell", "-w=8", "-h=8", "-spp=x", "-o=/tmp/x.png")
    assert r.returncode == 255 and "Invalid number of samples!" in r.stderr
    r = run(cli, "-s=cornell", "-w=8", "-h=8", "-spp=1", "-io=1", "-o=/tmp/x.png")
    assert r.returncode == 255 and "does not take a value" in r.stderr     # src/main.cpp:130-136
    r = run(cli, "-s=cornell", "-w=8", "-h=8", "-spp=1")
    assert r.returncode == 255 and "Required argument `--output`/`-o` not found!" in r.stderr


@pytest.mark.gpu
def test_cli_renders_the_same_image_as_the_oracle(cli, tmp_path):
    from PIL import Image
    out = str(tmp_path / "o.png")
    r = run(cli, "--scene=cornell-srgb", "--width=40", "-h=24", "--samples=6", "--output=" + out, "extra-arg",
            "--texture=data/scenes/test-img.png", "--seed=9")
    assert r.returncode == 0, r.stderr
    assert "Render completed in" in r.stdout and 'ignoring extraneous argument' in r.stderr and '"extra-arg"' in r.stderr
    o = ol.Oracle("cornell-srgb", texture="test-img.png")
    srgba = o.to_srgba(o.render(40, 24, 6, seed=9))
    want = np.floor(np.clip(np.float32(255.0) * srgba, 0, 255) + np.float32(0.5)).astype(np.uint8)[::-1]   # std::round, src/framebuffer.cpp:141-165
    assert np.array_equal(np.asarray(Image.open(out)), want)
    pfm = str(tmp_path / "o.pfm")
    r = run(cli, "-s=plane-srgb", "-w=16", "-h=16", "-spp=2", "-o=" + pfm, "--texture=data/scenes/test-img.png", "-io")
    assert r.returncode == 0 and "Plane converges much faster" in r.stderr and os.path.getsize(pfm) == len("PF\n16 16\n-1.0\n") + 16 * 16 * 12


def test_cli_meng_without_grid_file_fails_like_a_missing_data_file(cli, tmp_path):
    r = run(cli, "-s=cornell-srgb", "-w=8", "-h=8", "-spp=1", "-o=" + str(tmp_path / "x.png"), "--texture=data/scenes/test-img.png",
            "--uplift=meng", "--meng-grid=" + str(tmp_path / "missing.bin"))
    assert r.returncode == 255 and "Could not open Meng grid" in r.stderr
    r = run(cli, "-s=cornell-srgb", "-w=8", "-h=8", "-spp=1", "-o=" + str(tmp_path / "x.png"), "--uplift=nope")
    assert r.returncode == 255 and "Invalid value for --uplift" in r.stderr


@pytest.mark.gpu
def test_cli_meng_uplift_matches_the_oracle(cli, tmp_path):
    import ref_lib
    from simple_spectral_amd import meng
    from PIL import Image
    if ref_lib.meng() is None:
        pytest.skip("oracle/_ref/libref_meng.so not built")
    table = ref_lib.meng_table()
    grid = str(tmp_path / "grid.bin")
    meng.save_table(grid, table)
    out = str(tmp_path / "m.png")
    r = run(cli, "-s=cornell-srgb", "-w=32", "-h=24", "-spp=4", "-o=" + out, "--texture=data/scenes/test-img.png", "--seed=3",
            "--uplift=meng", "--meng-grid=" + grid)
    assert r.returncode == 0, r.stderr
    o = ol.Oracle("cornell-srgb", texture="test-img.png", meng=table)
    srgba = o.to_srgba(o.render(32, 24, 4, seed=3))
    want = np.floor(np.clip(np.float32(255.0) * srgba, 0, 255) + np.float32(0.5)).astype(np.uint8)[::-1]
    assert np.array_equal(np.asarray(Image.open(out)), want)


@pytest.mark.gpu
def test_cli_abort_saves_the_partial_render(cli, tmp_path):
    """The reference's abort path (window close -> render_stop -> the last worker saves what exists,
    src/main.cpp:318-327, src/renderer.cpp:388-394) hangs off Ctrl-C here."""
    import signal
    import time
    out = str(tmp_path / "partial.pfm")
    p = subprocess.Popen([cli, "-s=cornell", "-w=2048", "-h=2048", "-spp=4096", "-o=" + out], cwd=ROOT,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    time.sleep(4.0)                       # context + scene upload + a few of the ~50 ms launches (the whole render takes >10 s)
    p.send_signal(signal.SIGINT)
    so, se = p.communicate(timeout=120)
    assert p.returncode == 0, se
    assert "Aborting: saving the partial render" in se and "Render completed in" in so
    data = np.fromfile(out, dtype="<f4", offset=len("PF\n2048 2048\n-1.0\n")).reshape(2048, 2048, 3)
    assert np.isfinite(data).all() and data.max() > 0.0


@pytest.mark.gpu
def test_cli_abort_of_a_tile_major_render_keeps_the_checkerboard(cli, tmp_path):
    """`--tile-major`: the CLI walks through the tiles like the reference, so that the aborted image is what the reference's last
    worker saves (src/renderer.cpp:388-394): finished tiles next to the framebuffer's untouched 8x8 checkerboard
    (src/framebuffer.cpp:15-32: sRGB 0.7 / 0.3; the PFM holds their linear values); the tiles are finished from the bottom of the image up."""
    import signal
    import time
    out = str(tmp_path / "partial.pfm")
    p = subprocess.Popen([cli, "-s=cornell", "-w=2048", "-h=2048", "-spp=4096", "--tile-major", "-o=" + out], cwd=ROOT,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    time.sleep(4.0)
    p.send_signal(signal.SIGINT)
    so, se = p.communicate(timeout=120)
    assert p.returncode == 0, se
    assert "Aborting: saving the partial render" in se and "keep the checkerboard" in se
    data = np.fromfile(out, dtype="<f4", offset=len("PF\n2048 2048\n-1.0\n")).reshape(2048, 2048, 3)
    assert np.isfinite(data).all()
    lin = lambda v: ((v + 0.055) / 1.055) ** 2.4
    top = data[:8, :16, 0]                                # the file starts with the image's TOP row (src/framebuffer.cpp:112-140): the last tile row,
    #                                                       never reached in 4 s of a 6 s render -> checkerboard
    assert np.allclose(top[:, :8], lin(0.3), rtol=1e-3) or np.allclose(top[:, :8], lin(0.7), rtol=1e-3)
    assert not np.allclose(top[:, :8], top[:, 8:])
    floor = data[-256:]                                   # the image's bottom rows were rendered first: a floor, not a two-level pattern
    assert len(np.unique(floor[..., 0])) > 1000
    pattern = np.isclose(data[..., 0], lin(0.3), rtol=1e-3) | np.isclose(data[..., 0], lin(0.7), rtol=1e-3)
    rows = pattern.all(axis=1)                            # rows that are checkerboard from end to end: a block at the top of the image, in whole tiles
    n = int(rows.sum())
    assert 0 < n < 2048 and rows[:n - 8].all() and not rows[n + 8:].any(), n


def _png_of(srgba):
    return np.floor(np.clip(np.float32(255.0) * srgba, 0, 255) + np.float32(0.5)).astype(np.uint8)[::-1]   # std::round, src/framebuffer.cpp:141-165


@pytest.mark.gpu
def test_cli_multi_device_combine_on_the_device(cli, tmp_path):
    """`--gpus=N`: every device renders its tiles into its own HBM, device 0 pulls the peers' framebuffers
    device-to-device and adds them with a kernel (ssx_accumulate_peer; the reference's worker threads
    share one framebuffer instead, src/renderer.cpp:340-379).  SSX_TEST_ONE_GPU=1 puts all contexts on
    device 0, so the whole path -- tile split, peer copy, add, read-back -- runs on a 1-GPU box; the image
    must be the oracle's (= the single-device image)."""
    from PIL import Image
    out = str(tmp_path / "m.png")
    env = dict(os.environ, SSX_TEST_ONE_GPU="1")
    r = subprocess.run([cli, "-s=cornell-srgb", "-w=72", "-h=40", "-spp=5", "-o=" + out, "--texture=data/scenes/test-img.png", "--seed=4", "--gpus=3"],
                       cwd=ROOT, capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    o = ol.Oracle("cornell-srgb", texture="test-img.png")
    assert np.array_equal(np.asarray(Image.open(out)), _png_of(o.to_srgba(o.render(72, 40, 5, seed=4))))


@pytest.mark.gpu
def test_cli_procedural_texture(cli, tmp_path):
    from PIL import Image
    from simple_spectral_amd import textures
    out = str(tmp_path / "p.png")
    r = run(cli, "-s=plane-srgb", "-w=48", "-h=48", "-spp=3", "-o=" + out, "--texture=procedural:1024:3")
    assert r.returncode == 0, r.stderr
    o = ol.Oracle("plane-srgb", texture=textures.procedural_texture(1024, 3))
    assert np.array_equal(np.asarray(Image.open(out)), _png_of(o.to_srgba(o.render(48, 48, 3))))


@pytest.mark.gpu
def test_cli_combine_through_rccl(cli, tmp_path):
    """`--reduce=rccl`: the C++ host's combine of the per-device framebuffers as ONE RCCL reduce (ssx_reduce_rccl: ncclCommInitAll
    over the contexts' devices, grouped ncclReduce(sum) into device 0's buffer) instead of peer copies + adds.  On this one-GPU
    box the communicator has one rank; the file equals the default path's byte for byte.  Several contexts on ONE device (the
    test mode of the multi-device path) are refused: RCCL wants one rank per device."""
    a, b = str(tmp_path / "a.pfm"), str(tmp_path / "b.pfm")
    common = ["-s=cornell-srgb", "-w=48", "-h=32", "-spp=5", "--texture=data/scenes/test-img.png", "--seed=4"]
    r = run(cli, *common, "-o=" + a)
    assert r.returncode == 0, r.stderr
    r = run(cli, *common, "-o=" + b, "--reduce=rccl")
    assert r.returncode == 0, r.stderr
    assert open(a, "rb").read() == open(b, "rb").read()
    env = dict(os.environ, SSX_TEST_ONE_GPU="1")
    r = subprocess.run([cli] + common + ["-o=" + b, "--reduce=rccl", "--gpus=2"], cwd=ROOT, capture_output=True, text=True, env=env)
    assert r.returncode != 0 and "one rank per device" in (r.stderr + r.stdout)
    r = run(cli, *common, "-o=" + b, "--reduce=tree")
    assert r.returncode == 255
