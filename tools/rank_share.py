#!/usr/bin/env python3
"""Every rank's share of a multi-GPU render, timed on ONE GPU.

bench.py --gpus N deals the 8x8 tile list round-robin over N ranks and renders spp_per_gpu * N
samples per pixel on each (weak scaling: per-GPU work fixed).  A rank's share touches no other
rank until the final reduce, so it can be run alone: render_device(tile_first=r, tile_stride=N,
spp=spp_per_gpu*N).  What changes with N is the SHAPE of the share -- N times fewer tiles, N times
more samples per pixel -- and with it how many units of one tile are in flight at once (the
ordered binary64 pixel sums, ssx_kernels.hip: unit_fold) -- and WHICH tiles a rank owns: with the
plain round-robin a 64-tile row gives 8 ranks vertical stripes, the outer ones 7 % cheaper than the
inner ones; bench.py rotates the tile rows (tile_skew 1).  This prints ms per step of the shares
(--all-ranks: every r; else r = 0 and N-1) for N = 1, 2, 4, 8 and the predicted weak-scaling
efficiency t(1) / max_r t(N, r).

    python tools/rank_share.py [--configs headline,plane,cie2006] [--steps 10] [--tag NAME]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (scene, res, spp per GPU, observer, the N the BASELINE config is quoted on)
    "headline": ("cornell-srgb", 512, 256, 1931, (1, 2, 4, 8)),     # BASELINE configs[1], the bench.py workload
    "plane": ("plane-srgb", 1024, 256, 1931, (1, 4)),                # configs[3]: 4 GPUs, spp 4096 total -> reduced to 256/GPU
    "cie2006": ("cornell-srgb", 2048, 16, 2006, (1, 8)),             # configs[4]: 8 GPUs, spp 16384 total -> reduced to 16/GPU
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="headline,plane,cie2006")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--tag", default="")
    ap.add_argument("--all-ranks", action="store_true", help="every r, not only 0 and N-1")
    ap.add_argument("--skew", type=int, default=1, help="tile_skew of the shares (bench.py: 1); 0 = vertical stripes")
    ap.add_argument("--only-n", type=int, default=0, help="one N only, rank 0 (for counter passes: tools/pmc_share.sh)")
    args = ap.parse_args()

    import torch
    from simple_spectral_amd import Options, Renderer

    torch.cuda.set_device(0)
    stream = torch.cuda.current_stream()
    out_lines = []
    for name in args.configs.split(","):
        scene, res, spp, observer, ns = CONFIGS[name]
        out = torch.zeros((res, res, 4), dtype=torch.float32, device="cuda")
        t1 = None
        for n in ns:
            if args.only_n and n != args.only_n:
                continue
            ranks = [0] if args.only_n else (range(n) if args.all_ranks else sorted({0, n - 1}))
            worst = 0.0
            for r in ranks:
                rd = Renderer(Options(scene_name=scene, res=(res, res), spp=spp * n, texture="crystal-lizard-512.png", device=0,
                                      tile_first=r, tile_stride=n, tile_skew=args.skew if n > 1 else 0, seed=0, observer=observer))
                for _ in range(args.warmup):
                    rd.render_device(out.data_ptr(), stream.cuda_stream)
                torch.cuda.synchronize()
                first = out.clone()  # every later step must reproduce it bit for bit: the hand-over of the pixel sums under real contention
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    rd.render_device(out.data_ptr(), stream.cuda_stream)
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t0) / args.steps * 1e3
                worst = max(worst, ms)
                if not torch.equal(out.view(torch.int32), first.view(torch.int32)):
                    raise SystemExit("rank share N=%d r=%d: the image changed from one step to the next" % (n, r))
                sums = rd.sums_info()
                rd.set_timing(True)  # three more steps with HIP events around the kernels: which one carries a difference
                for _ in range(3):
                    rd.render_device(out.data_ptr(), stream.cuda_stream)
                torch.cuda.synchronize()
                stages = {k: round(v / 3, 3) for k, v in rd.get_timing().items() if k in ("generate", "path")}
                rd.set_timing(False)
                n_units = (args.warmup + args.steps) * (res // 8) ** 2 // n * ((spp * n + 3) // 4)  # (units of 4 samples per pixel: Cornell; the plane scene's have 8)
                rd.close()
                line = {"tag": args.tag, "config": name, "scene": scene, "res": res, "spp_per_gpu": spp, "observer": observer,
                        "N": n, "rank": r, "tiles": (res // 8) ** 2 // n, "spp": spp * n, "ms_per_step": round(ms, 3),
                        "Msamples_per_s": round(res * res * spp / ms / 1e3, 1),
                        "units_parked_frac": round(sums["units_parked"] / n_units, 4), "units_chained_frac": round(sums["units_chained"] / n_units, 4), "stage_ms": stages}
                print(json.dumps(line), flush=True)
                out_lines.append(line)
            if n == 1 or t1 is None:
                t1 = worst
            print("# %s %s N=%d: worst share %.3f ms, predicted weak-scaling efficiency t(1)/max_r t(N,r) = %.3f"
                  % (args.tag, name, n, worst, t1 / worst), flush=True)
        del out


if __name__ == "__main__":
    main()
