#!/usr/bin/env python3
"""bench.py -- Msamples/s of the spectral integrator on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A step = one full render of BASELINE.json configs[1] (cornell-srgb 512x512, hero-wavelength,
CIE 1931, spp=256 per GPU): the 8x8 tile list is dealt round-robin over the N ranks, every rank
renders its tiles at spp = 256*N (per-GPU work fixed -> weak scaling) into a zero-initialised
full-size float4 XYZA buffer on its GPU, and one RCCL reduce(sum) to rank 0 combines them (x+0
is exact, so the sum is the image).  Inputs (scene tables, texture) are resident in HBM before the
timed region; the timed region is K x (render [+ reduce]).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Algorithmic FP32 work per sample, SURVEY.md section 8(d): 33*T + 28*P + 550*S + 150 with the
# measured per-sample counts (T tri tests, P edge-test passes, S surface interactions).
FLOP_PER_SAMPLE = {"cornell-srgb": 1.28e4, "cornell": 1.28e4, "plane-srgb": 2.8e3}
# Algorithmic HBM bytes per sample of the path megakernel: the 32-byte
# record is read and rewritten, and one 48-byte frame is written per continued bounce
# (frames/sample = interactions that continue: 3.29 Cornell, 1 plane [oracle statistics]).
FRAMES_PER_SAMPLE = {"cornell-srgb": 3.29, "cornell": 3.29, "plane-srgb": 1.0}
SHADOW_RMW_BYTES = {"cornell-srgb": 80, "cornell": 80, "plane-srgb": 30}
# gfx950 FP32 vector peak is 157.3 TFLOP/s counting FMA as 2; the parity contract forbids
# contraction, so the ceiling that applies is the non-fused issue rate, half of it.
PEAK_VALU_TFLOPS = 78.6
PEAK_HBM_GBS = 8000.0


def cpu_baseline(scene, W, H, texture, target_seconds=12.0):
    """The CPU oracle (oracle/, a port of the reference's threaded tile renderer) on the host
    cores of this box, on a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol

    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    o = ol.Oracle(scene, texture=texture)
    t = time.time()
    o.render(W, H, 1, nthreads=cores)
    t1 = max(time.time() - t, 1e-3)
    spp = int(max(1, min(128, target_seconds / t1)))
    t = time.time()
    o.render(W, H, spp, nthreads=cores)
    dt = time.time() - t
    return {"value": round(W * H * spp / dt / 1e6, 4), "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": "%s %dx%d spp=%d (%.1f s, oracle/libssx_oracle.so, %d threads, 8x8 tile queue)" % (scene, W, H, spp, dt, cores)}


def measured_traffic(args, world):
    """HBM bytes per launch of the megakernel from the PMC passes (FETCH_SIZE, WRITE_SIZE collected
    in separate rocprofv3 --pmc runs, FETCH_SIZE doubled per MI355X_MICROARCH.md for wide coalesced
    reads), as recorded by tools/collect_traffic.py for this exact workload; None otherwise."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        t = json.load(open(path))
        key = "%s %d spp%d obs%d gpus%d" % (args.scene, args.res, args.spp, args.observer, world)
        return t.get(key)
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scene", default="cornell-srgb")
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--spp", type=int, default=256, help="samples per pixel PER GPU")
    ap.add_argument("--observer", type=int, default=1931)
    ap.add_argument("--uplift", default="ours", choices=["ours", "jh"])
    ap.add_argument("--batch", type=int, default=0, help="spp per pipelined batch (0 = library default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from simple_spectral_amd import Options, Renderer

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d" % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: simple_spectral_amd has no CPU path")
    # SSX_BENCH_TEST_ONE_GPU=1: plumbing test of the N>1 path on a 1-GPU box (all ranks share
    # device 0 and the reduce goes through gloo on host copies); never set by the driver.
    test_one_gpu = os.environ.get("SSX_BENCH_TEST_ONE_GPU") == "1"
    if test_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if test_one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    W = H = args.res
    spp_total = args.spp * world
    texture = "crystal-lizard-512.png"
    r = Renderer(Options(scene_name=args.scene, res=(W, H), spp=spp_total, texture=texture, device=local_rank,
                         tile_first=rank, tile_stride=world, seed=0, observer=args.observer, uplift=args.uplift,
                         spp_per_launch=args.batch))
    out = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream()

    def reduce_to_rank0():
        # the one exchange step of the path: sum of the per-rank framebuffers (RCCL over xGMI)
        if test_one_gpu:
            h = out.cpu()
            dist.reduce(h, dst=0, op=dist.ReduceOp.SUM)
            out.copy_(h)
        else:
            dist.reduce(out, dst=0, op=dist.ReduceOp.SUM)

    def step():
        r.render_device(out.data_ptr(), stream.cuda_stream)
        if world > 1:
            reduce_to_rank0()

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    r.set_timing(True)  # HIP events on the launch stream around each kernel of the pipeline
    # kernel duration: HIP events on the launch stream around each render (memset + megakernel +
    # finalize; the two small kernels are microseconds next to the megakernel)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record(stream)
        r.render_device(out.data_ptr(), stream.cuda_stream)
        ev[k][1].record(stream)
        if world > 1:
            reduce_to_rank0()
    fence()
    elapsed = time.perf_counter() - t0
    pipeline_ms = sum(a.elapsed_time(b) for a, b in ev) / max(args.steps, 1)
    stage_ms = {k: v / max(args.steps, 1) for k, v in r.get_timing().items()}
    kernel_ms = stage_ms["path"]  # the dominant kernel (ssx_render_kernel), mean per launch
    if world > 1:
        tt = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device="cpu" if test_one_gpu else "cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed, kernel_ms = float(tt[0]), float(tt[1])

    if rank == 0:
        samples_per_step = W * H * spp_total
        value = samples_per_step * args.steps / elapsed / 1e6
        per_gpu_samples = W * H * args.spp
        flop = FLOP_PER_SAMPLE.get(args.scene, 1.28e4)
        achieved_tflops = per_gpu_samples * flop / (kernel_ms * 1e-3) / 1e12
        # algorithmic HBM traffic of the path kernel per launch and sample: record read (32 B) + rewritten at the end
        # of the path (32) + read again and overwritten with XYZA by the fold (32 + 16); 48-B frames written once and
        # read once by the fold
        # read once by the fold; ~3 parked shadow rays per Cornell sample read 16 B of their target and write it back
        # when the light is visible
        hbm_bytes = per_gpu_samples * (112 + 96 * FRAMES_PER_SAMPLE.get(args.scene, 3.29) + SHADOW_RMW_BYTES.get(args.scene, 80))
        info = r.kernel_info()
        line = {
            "metric": "Msamples/s (w*h*spp/s) %s %dx%d" % (args.scene, W, H),
            "value": round(value, 2), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s %dx%d spp=%d/GPU (total spp %d) CIE%d uplift=%s hero-wavelength megakernel" % (args.scene, W, H, args.spp, spp_total, args.observer, args.uplift),
                       "parallelism": "tile-split x%d + RCCL reduce" % world if world > 1 else "single GPU",
                       "texture": texture, "seed": 0},
            "roofline": {"bound": "valu", "achieved": round(achieved_tflops, 3), "peak": PEAK_VALU_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved_tflops / PEAK_VALU_TFLOPS, 4), "traffic": measured_traffic(args, world),
                         "kernel": "ssx_render_kernel", "kernel_ms": round(kernel_ms, 3),
                         "pipeline_ms": round(pipeline_ms, 3), "stage_ms": {k: round(v, 3) for k, v in stage_ms.items()},
                         "flop_per_sample": flop, "note": "FP32 VALU-issue bound, no MFMA, HBM idle by design; peak = 157.3/2 (no FMA contraction under the parity contract)",
                         "hbm": {"achieved": round(hbm_bytes / (kernel_ms * 1e-3) / 1e9, 3), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                 "frac": round(hbm_bytes / (kernel_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 7)},
                         "vgprs": info["vgprs"], "scratch_bytes": info["scratch_bytes"], "lds_bytes": info["lds_bytes"],
                         "plan": r.plan_info()},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.scene, W, H, texture)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
