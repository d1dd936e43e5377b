#!/usr/bin/env python3
"""bench.py -- Msamples/s of the spectral integrator on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

N > 1 runs one process per GPU over RCCL: either launched by torch.distributed.run (the driver's way:
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment) or, when WORLD_SIZE is not set, this
script spawns the N ranks itself (127.0.0.1 rendezvous) and relays rank 0's JSON line.

A step = one full render of BASELINE.json configs[1] (cornell-srgb 512x512, hero-wavelength,
CIE 1931, spp=256 per GPU): the 8x8 tile list is dealt round-robin over the N ranks -- each tile row rotated by its row number
(tile_skew 1), so that a rank owns diagonals, not vertical stripes of unequal cost --, every rank
renders its tiles at spp = 256*N (per-GPU work fixed -> weak scaling) into a zero-initialised
full-size float4 XYZA buffer on its GPU, and one RCCL reduce(sum) to rank 0 combines them (x+0
is exact, so the sum is the image).  Inputs (scene tables, texture) are resident in HBM before the
timed region; the timed region is K x (render [+ reduce] + copy of the combined XYZA image into pinned host memory on
rank 0): the metric as SURVEY.md section 8(d) defines it ("framebuffer reduce + D2H of XYZA included").  The copies run on their
own stream between two device and two host images, so step k's image travels while step k+1 renders; all K images are in host
memory when the timed region ends.  The same K steps without the copy are timed afterwards and reported as value_device_resident.

Prints ONE JSON line on rank 0 -- and only for an image that passed: the last timed step's image (from pinned host memory) is compared
with the CPU oracle on six 8x8 tiles at full spp ("check"), every rank's record is gathered ("ranks": device uuid / bus id, hostname, ms per
step, path kernel ms, share of parked units), two ranks on one device or overlapping rank images end the run without a line ("distributed"),
and at N = 1 the HBM traffic of the path and generate kernels is measured in the run by two rocprofv3 --pmc passes over a child run of the
same workload (roofline.traffic; replayed from profiles/traffic.json with a staleness stamp when counters are unavailable).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Algorithmic FP32 work per sample, SURVEY.md section 8(d): 33*T + 28*P + 550*S + 150 with the
# measured per-sample counts (T tri tests, P edge-test passes, S surface interactions).
FLOP_PER_SAMPLE = {"cornell-srgb": 1.28e4, "cornell": 1.28e4, "plane-srgb": 2.8e3}
# Algorithmic HBM bytes per sample of the whole pipeline, with L = continued levels
# per sample (the calibration render of the scene at upload measures it: plan_info()["frames_per_sample"], 4.07
# for the Cornell box, 1.0 for the plane), R = parked shadow rays per sample (3.08 Cornell, 1.36 plane: lane
# statistics of the profiling build, profiles/*/lanestat.log) and E = levels with an emission term (~0.01:
# camera rays that hit the light):
#   generate  : write camera ray 16 + stream 16 + camera hit 16
#   path      : read the three 48; per continued level append fs 16 + np 8 + chain word 4; emission term 16 E;
#               at the end write {lambda, tail word, final stream state} 16
#   shadow    : R x write nee 16 (write-only: contribution or zeros)
#   fold      : read tail 16 + (fs 16 + np 8 + chain word 4) L + nee 16 R + emission 16 E; the pixel sums (4 x binary64 per pixel)
#               are read and written once per work unit of U samples per pixel: 64 / U
LEVELS = {"cornell-srgb": 4.07, "cornell": 4.07, "plane-srgb": 1.0}
SHADOW = {"cornell-srgb": 3.08, "cornell": 3.08, "plane-srgb": 0.5}   # (plane-srgb: 0.5 parked -- the other 0.86 of SURVEY's 1.36 shadow rays start on black walls and carry nothing)
# SURVEY 8(d)'s formula prices every surface interaction S at 550 flop (next-event estimation ~450 incl. 16 transcendentals, BSDF sample ~100).  In
# plane-srgb the second of a path's two interactions lies on a BLACK wall: everything it computes is multiplied by f_s = 0, and since round 6 the
# kernel reduces it, exactly, to its random draws (csrc/ssx_kernels.hip path_step).  The contract's figure (2.8e3) stays the one `roofline.achieved`
# uses; the line also states the fraction with those 500 flop per sample taken out of the numerator (roofline.frac_without_eliminated_work).
FLOP_ELIMINATED_PER_SAMPLE = {"plane-srgb": 500.0}
# gfx950 FP32 vector peak is 157.3 TFLOP/s counting FMA as 2; the parity contract forbids
# contraction, so the ceiling that applies is the non-fused issue rate, half of it.
PEAK_VALU_TFLOPS = 78.6
# one-thread rate of oracle/libssx_oracle_refshape.so in the build container (8-core Xeon 2.1 GHz; profiles/r06/NOTES.md), Msamples/s
REFSHAPE_BUILD_BOX = {"cornell-srgb": 0.097, "plane-srgb": 0.415}
PEAK_HBM_GBS = 8000.0


# what a megakernel with ALL state on chip would move -- one float4 store per pixel and the texel
# fetches (3.8 B per sample for the textured Cornell box, 4.5 for the plane, none for "cornell") -- against which the
# design's own bytes (below: the recursion's levels are logged to HBM, the price of the post-order fold that reproduces
# the reference's float rounding) and the counters are reported.
TEXEL_BYTES_PER_SAMPLE = {"cornell-srgb": 3.8, "cornell": 0.0, "plane-srgb": 4.5}


def algorithmic_bytes_per_sample_8d(scene, spp):
    return 16.0 / spp + TEXEL_BYTES_PER_SAMPLE.get(scene, 3.8)


def design_bytes_per_sample(scene, path_kernel_only=False, levels=None, plan=None):
    L = levels if levels else LEVELS.get(scene, 4.07)
    R = SHADOW.get(scene, 3.08)
    E = 0.01
    U = 4.0 if L >= 2.0 else 8.0                                                          # samples per pixel of a work unit (make_batch)
    # the records the generate kernel writes and the refill reads: ray 16 + stream 16 (+ camera hit 16 where camera rays are traced ahead); none
    # where the path kernel makes its samples itself (plan_info: samples_made_in, round 6: plane-srgb)
    rec = 48.0
    if plan and plan.get("camera_rays") == "path loop":
        rec = 32.0
    if plan and str(plan.get("samples_made_in", "")).startswith("path kernel"):
        rec = 0.0
    path = (rec + 28 * L + 16 * E + 16) + 16 * R + (16 + 28 * L + 16 * R + 16 * E + 64.0 / U)   # path loop + shadow flush + fold
    return path if path_kernel_only else rec + path


def host_cpu_info():
    """What the CPU baseline may use: affinity mask, cgroup CPU quota, physical cores."""
    aff = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = float(q) / float(p)
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / p
        except Exception:
            pass
    phys = set()
    model = ""
    try:
        pid = cid = None
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("physical id"):
                pid = ln.split(":")[1].strip()
            elif ln.startswith("core id"):
                cid = ln.split(":")[1].strip()
                phys.add((pid, cid))
            elif ln.startswith("model name") and not model:
                model = ln.split(":")[1].strip()
    except Exception:
        pass
    return {"affinity_threads": aff, "cgroup_quota_cpus": quota, "physical_cores": len(phys) or None, "model": model}


def cpu_baseline(scene, W, H, texture, target_seconds=10.0):
    """The CPU oracle (oracle/, a port of the reference's threaded tile renderer: 8x8 tile queue under a
    mutex, src/renderer.cpp:340-409) on the host cores of this box, on a bounded sample of the same
    workload: first one thread (a tile rectangle), then all usable threads (the whole image).
    Usable = affinity mask, capped by the cgroup CPU quota when there is one."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol

    info = host_cpu_info()
    cores = info["affinity_threads"]
    if info["cgroup_quota_cpus"]:
        cores = max(1, min(cores, int(info["cgroup_quota_cpus"] + 0.5)))
    cores = min(cores, 256)
    o = ol.Oracle(scene, texture=texture)
    # one thread: 64x64 pixels in the middle of the image
    rect = (W // 2 - 32, H // 2 - 32, W // 2 + 32, H // 2 + 32) if W >= 64 and H >= 64 else (0, 0, W, H)
    npx = (rect[2] - rect[0]) * (rect[3] - rect[1])
    t = time.time(); o.render(W, H, 2, rect=rect, nthreads=1); t1 = max(time.time() - t, 1e-3)
    spp1 = int(max(2, min(64, 2 * 2.5 / t1)))
    t = time.time(); o.render(W, H, spp1, rect=rect, nthreads=1); dt1 = time.time() - t
    rate1 = npx * spp1 / dt1 / 1e6
    # all threads: the whole image
    t = time.time(); o.render(W, H, 1, nthreads=cores); tn = max(time.time() - t, 1e-3)
    spp = int(max(1, min(256, 0.6 * target_seconds / tn)))
    t = time.time(); o.render(W, H, spp, nthreads=cores); dt = time.time() - t
    rate = W * H * spp / dt / 1e6
    phys = info["physical_cores"] or cores
    # How the port compares with the reference BINARY (BASELINE.md 4(i) asked for +-10 %; it is ~2x as fast: same algorithm without
    # std::function recursion, virtual calls and GLM temporaries): measured in the build container by tools/port_vs_reference_probe.py
    probe = None
    try:
        pr = json.load(open(os.path.join(ROOT, "profiles", "r04", "port_vs_reference_probe.json")))
        sc = pr["scenes"].get(scene)
        if sc:
            probe = {"one_thread_ref": sc["one_thread_ref"], "one_thread_port_same_box": sc["one_thread_port_same_box"],
                     "port_over_ref_one_thread": sc["port_over_ref_one_thread"], "port_over_ref_eight_threads": sc["port_over_ref_eight_threads"],
                     "box": pr["host"], "source": "tools/port_vs_reference_probe.py in the build container; reference side: survey probe of the reference binary there (BASELINE.md section 2)"}
    except Exception:
        pass
    # The number BASELINE.md section 4 wants next to the GPU figure is the REFERENCE's rate on these cores.  MEASURED since round 6: the
    # oracle built with the reference binary's call structure (oracle/libssx_oracle_refshape.so, -DORACLE_REFERENCE_SHAPED: virtual intersect
    # per primitive, the shear constants of geometry.cpp:17-37 per triangle, run-time indexed vec3 temporaries, virtual material calls, the
    # recursion through a function pointer like the std::function of renderer.cpp:148; same bits as the oracle), timed on these cores like the
    # port above.  The figure derived from the probe's port/reference ratio (rounds 4-5) stays next to it, marked as derived.
    ref_equiv = None
    try:
        oref = ol.Oracle(scene, texture=texture, variant="refshape")
        t = time.time(); oref.render(W, H, 2, rect=rect, nthreads=1); t1r = max(time.time() - t, 1e-3)
        spp1r = int(max(2, min(64, 2 * 2.0 / t1r)))
        t = time.time(); oref.render(W, H, spp1r, rect=rect, nthreads=1); dt1r = time.time() - t
        rate1r = npx * spp1r / dt1r / 1e6
        t = time.time(); oref.render(W, H, 1, nthreads=cores); tnr = max(time.time() - t, 1e-3)
        sppr = int(max(1, min(256, 0.6 * target_seconds / tnr)))
        t = time.time(); oref.render(W, H, sppr, nthreads=cores); dtr = time.time() - t
        rater = W * H * sppr / dtr / 1e6
        ref_equiv = {"value": round(rater, 4), "unit": "Msamples/s", "cores": cores, "one_thread": round(rate1r, 4), "measured": True,
                     "kind": "port, reference-shaped: oracle/libssx_oracle_refshape.so (the oracle's arithmetic in the reference binary's call structure)",
                     "port_over_reference_shaped": round(rate / rater, 3),
                     "sample": "%s %dx%d spp=%d (%.1f s, %d threads); one thread: %d px x spp=%d (%.1f s)" % (scene, W, H, sppr, dtr, cores, npx, spp1r, dt1r),
                     "how": "timed in this run on this box's cores (the reference binary itself cannot run here: it needs GLM and /root/reference)",
                     "build_box_check": {"reference_shaped_one_thread": REFSHAPE_BUILD_BOX.get(scene), "survey_reference_one_thread": probe["one_thread_ref"] if probe else None,
                                         "box": "8-core Xeon 2.1 GHz build container (profiles/r06/NOTES.md: the shaped port explains part of the port/reference gap; the survey's reference figure was taken with a GLM stand-in and is 'indicative', SURVEY section 6)"}}
        if probe:
            ref_equiv["derived_from_probe_ratio"] = {"value": round(rate / probe["port_over_ref_eight_threads"], 4), "port_over_ref": probe["port_over_ref_eight_threads"],
                                                     "how": "cpu_baseline.value / port_over_ref of tools/port_vs_reference_probe.py (rounds 4-5's figure; derived, not measured)"}
    except Exception as e:  # noqa: BLE001 -- (a tree without the refshape library: the derived figure alone, as in round 5)
        if probe:
            ref_equiv = {"value": round(rate / probe["port_over_ref_eight_threads"], 4), "unit": "Msamples/s", "cores": cores, "measured": False,
                         "port_over_ref": probe["port_over_ref_eight_threads"], "why_not_measured": "%s: %s" % (type(e).__name__, str(e)[:200]),
                         "how": "cpu_baseline.value / port_over_ref (derived)"}
    return {"value": round(rate, 4), "unit": "Msamples/s", "cores": cores, "kind": "port", "reference_equivalent": ref_equiv, "vs_reference_probe": probe,
            "one_thread": round(rate1, 4), "scaling_efficiency": round(rate / (rate1 * cores), 3),
            "efficiency_vs_physical_cores": round(rate / (rate1 * min(cores, phys)), 3), "host": info,
            "sample": "%s %dx%d spp=%d (%.1f s, oracle/libssx_oracle.so, %d threads, 8x8 tile queue); one thread: %d px x spp=%d (%.1f s); port, ~%sx the reference binary's rate (vs_reference_probe)"
                      % (scene, W, H, spp, dt, cores, npx, spp1, dt1, ("%.1f" % probe["port_over_ref_one_thread"]) if probe else "2")}


def oracle_check(img, scene, W, H, spp_total, observer, texture, n_tiles=6):
    """The checker of the cpu_baseline leg, applied to the image the timed region produced: `n_tiles` 8x8 tiles of the LAST timed step's
    combined image (as it arrived in pinned host memory) against the CPU oracle at full spp, bit for bit.  The oracle is only ever the
    checker here -- never the thing measured.  A bench line is printed only for an image that passes."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_lib as ol

    o = ol.Oracle(scene, observer=observer, texture=texture)
    tx, ty = (W + 7) // 8, (H + 7) // 8
    # corners, centre and two tiles at fixed fractions of the image (the red textured wall, a block face): spread over the tile list
    picks = [(0, 0), (tx // 2, ty // 2), (tx - 1, ty - 1), ((15 * tx) // 64, (50 * ty) // 64), ((55 * tx) // 64, (20 * ty) // 64), ((5 * tx) // 64, (30 * ty) // 64)][:n_tiles]
    t = time.time()
    differing = 0
    checked = []
    for (a, b) in picks:
        i0, j0 = a * 8, b * 8
        i1, j1 = min(i0 + 8, W), min(j0 + 8, H)
        ref = o.render(W, H, spp_total, rect=(i0, j0, i1, j1), nthreads=1)
        got = np.ascontiguousarray(img[j0:j1, i0:i1], dtype=np.float32).view(np.uint32)
        differing += int((got != np.ascontiguousarray(ref[j0:j1, i0:i1], dtype=np.float32).view(np.uint32)).sum())
        checked.append([i0, j0])
    alpha = float(img[..., 3].mean())
    return {"tiles": len(checked), "tile_origins": checked, "spp": spp_total, "differing_floats": differing, "image": "last timed step, from pinned host memory",
            "against": "oracle/libssx_oracle.so (CPU restatement), uint32 compare of float4 XYZA", "mean_alpha": round(alpha, 5), "seconds": round(time.time() - t, 2)}


def n1_reference(src_id=None):
    """The N = 1 figure the efficiency field is quoted against -- only one taken on THIS build's kernel sources counts (VERDICT r05 item 5:
    a figure of another round's kernel must be refused, not divided by): the driver's N = 1 records in the tree (BENCH_rNN.json, whose
    `config` carries kernel_source_id since round 6) and the builder's own closing lines (profiles/rNN/bench.json), newest match first;
    or SSX_BENCH_N1_VALUE=<Msamples/s>, named as what it is.  No match: value None and the reason."""
    import glob
    src_id = src_id or kernel_source_id()
    env = os.environ.get("SSX_BENCH_N1_VALUE")
    if env:
        try:
            return {"value": float(env), "source": "SSX_BENCH_N1_VALUE (given by the caller: not checked against the kernel sources)", "kernel_source_id": None}
        except ValueError:
            pass
    seen = []
    for f in sorted(glob.glob(os.path.join(ROOT, "BENCH_r*.json"))) + sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]", "bench.json"))):
        try:
            d = json.load(open(f))
            d = d.get("parsed") or d
            if d.get("n_gpus") == 1 and d.get("value") and "cornell-srgb 512x512 spp=256/GPU" in str((d.get("config") or {}).get("workload", "")):
                seen.append((os.path.relpath(f, ROOT), float(d["value"]), (d.get("config") or {}).get("kernel_source_id")))
        except Exception:
            pass
    match = [x for x in seen if x[2] == src_id]
    if match:
        f, v, _ = match[-1]
        return {"value": v, "source": "%s (N = 1 run on these kernel sources)" % f, "kernel_source_id": src_id}
    newest = seen[-1] if seen else None
    return {"value": None, "source": None, "kernel_source_id": src_id,
            "refused": "no N = 1 figure taken on this build's kernel sources (id %s) in the tree%s" % (
                src_id, "; newest other: %s = %.2f Msamples/s on %s" % (newest[0], newest[1], newest[2] or "sources without an id (before round 6)") if newest else "")}


def sha256_of(*paths):
    import hashlib
    h = hashlib.sha256()
    for p in paths:
        try:
            h.update(open(p, "rb").read())
        except OSError:
            h.update(b"<missing>")
    return h.hexdigest()[:16]


def kernel_source_id():
    """Identifies the device code a measurement belongs to: hash of every source the HIP library is compiled from (simple_spectral_amd/build.py
    HIP_DEPS minus the generated file -- the .hip files, their headers incl. ssx_jit.h / ssx_lanestat.h, include/ssx.h and ssx_fmath.h), not of the
    .so, whose bytes differ per build host."""
    from simple_spectral_amd import build as b
    return sha256_of(*sorted(d for d in b.HIP_DEPS if not d.endswith("ssx_sources_gen.h")))


def newest_pmc_summary():
    """profiles/rNN/pmc_summary.csv of the latest round that has one (what a replayed traffic figure is said to agree with)."""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]", "pmc_summary.csv")))
    return found[-1] if found else None


PMC_KERNELS = (("path", "ssx_render_kernel"), ("generate", "ssx_generate_kernel"))


def pmc_passes(args):
    """HBM traffic of THIS run's workload, measured now: two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: no trace flags next to
    them, as /opt/skills/guides/MI355X_MICROARCH.md prescribes) over a short child run of this script (--pmc-child: same scene, same launches,
    prints nothing).  Corrections of that guide: both counters are KiB; on gfx950 FETCH_SIZE reports half the bytes of wide coalesced read
    streams (what this pipeline's reads are: calibrated on known byte counts, tools/ubench/traffic_calib) -> doubled; WRITE_SIZE as is.
    Returns ({kernel: bytes per render}, detail) or (None, reason)."""
    import csv
    import glob
    import shutil
    import tempfile
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None, "rocprofv3 not found"
    out_root = os.path.join(ROOT, "gpurun_out") if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else tempfile.gettempdir()
    child = [sys.executable, os.path.abspath(__file__), "--pmc-child", "--steps", "2", "--warmup", "1", "--scene", args.scene, "--res", str(args.res), "--spp", str(args.spp),
             "--observer", str(args.observer), "--uplift", args.uplift, "--texture", args.texture, "--batch", str(args.batch), "--scratch-cap-gb", str(args.scratch_cap_gb)]
    env = dict(os.environ, TMPDIR="/tmp")
    got = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="ssx_pmc_%s_" % counter.lower(), dir=out_root)
        try:
            subprocess.run([prof, "--pmc", counter, "--output-format", "csv", "-d", d, "--"] + child, cwd="/tmp", env=env, timeout=240,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if not files:
                return None, "rocprofv3 wrote no counter file for %s" % counter
            per = {k: [] for k, _ in PMC_KERNELS}
            renders = 0
            for row in csv.DictReader(open(files[0])):
                if row.get("Counter_Name") != counter:
                    continue
                if "ssx_finalize_kernel" in row.get("Kernel_Name", ""):
                    renders += 1                                                 # one per render: a render may be several launches (batches)
                for k, pat in PMC_KERNELS:
                    if pat in row.get("Kernel_Name", ""):
                        per[k].append(float(row["Counter_Value"]))
            # per RENDER (= per step of the bench), not per launch: a render split into batches is several launches of each kernel (until
            # round 6 the mean over launches was divided by the whole step's samples -- right for the one-launch headline, wrong for plane-srgb
            # 1024^2 spp 1024, which runs in batches).  The child's first renders are its warm-up: same size, so every render counts.
            got[counter] = {k: (sum(v) / renders if v and renders else (0.0 if renders and k == "generate" else None)) for k, v in per.items()}
            got[counter + "_launches"] = {k: len(v) for k, v in per.items()}
            got[counter + "_renders"] = renders
        except Exception as e:  # a pool without counter access, a timeout: say so, the line then replays the stamped file
            return None, "%s pass failed: %s" % (counter, str(e)[:200])
        finally:
            shutil.rmtree(d, ignore_errors=True)
    by_kernel = {}
    for k, _ in PMC_KERNELS:
        f, w = got["FETCH_SIZE"][k], got["WRITE_SIZE"][k]
        if f is None or w is None:
            return None, "no %s launches in the counter files" % k
        by_kernel[k] = int(2.0 * f * 1024 + w * 1024)
    detail = {"FETCH_SIZE_KiB": got["FETCH_SIZE"], "WRITE_SIZE_KiB": got["WRITE_SIZE"], "fetch_correction": 2.0, "launches": got["FETCH_SIZE_launches"], "renders": got["FETCH_SIZE_renders"],
              "bytes_per_launch": by_kernel, "per": "render (= step): every launch of the kernel within one render summed"}
    return by_kernel, detail


def measured_traffic(args, world):
    """HBM bytes per launch of the path kernel (roofline.traffic) and of the generate kernel.  N = 1: measured in this run by two
    rocprofv3 --pmc passes (pmc_passes) unless --no-pmc / SSX_BENCH_NO_PMC=1 or the passes fail; the result goes to
    gpurun_out/traffic_measured.json (and refreshes profiles/traffic.json only with --update-traffic).  Otherwise replayed from that file -- stamped with the hash of the kernel sources it was taken on and of
    the counter summary it agrees with, and marked stale when the sources have changed since, so that it cannot go stale silently."""
    path = os.environ.get("SSX_BENCH_TRAFFIC_JSON") or os.path.join(ROOT, "profiles", "traffic.json")   # (tests point it elsewhere)
    key = "%s %d spp%d obs%d gpus%d" % (args.scene, args.res, args.spp, args.observer, world)
    src_id = kernel_source_id()
    if world == 1 and not args.no_pmc and os.environ.get("SSX_BENCH_NO_PMC") != "1":
        by_kernel, detail = pmc_passes(args)
        if by_kernel:
            detail["kernel_source_id"] = src_id
            # A run does not touch the tracked replay file (every driver run dirtied the tree, and the stored evidence followed
            # whichever box ran last): the fresh figures go to gpurun_out/traffic_measured.json (scratch), and into profiles/traffic.json
            # only with --update-traffic (the builder's closing run of a round).
            targets = [os.path.join(ROOT, "gpurun_out", "traffic_measured.json")] + ([path] if args.update_traffic else [])
            for tp in targets:
                try:
                    if os.path.isdir(os.path.dirname(tp)):
                        t = json.load(open(tp)) if os.path.exists(tp) else {}
                        t[key] = by_kernel["path"]
                        t[key + " detail"] = detail
                        json.dump(t, open(tp, "w"), indent=1, sort_keys=True)
                except Exception:
                    pass
            return by_kernel["path"], detail, "measured in this run: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over a child run of the same workload; FETCH_SIZE x 2 (gfx950 wide-read correction), KiB -> bytes; path kernel (traffic) and generate kernel (traffic_detail.bytes_per_launch)"
        why = detail
    else:
        why = "--no-pmc" if world == 1 else "N > 1: counters are collected at N = 1"
    try:
        t = json.load(open(path))
        detail = dict(t.get(key + " detail") or {})
        taken_on = detail.get("kernel_source_id")
        summary = newest_pmc_summary()
        detail["replayed"] = {"file": "profiles/traffic.json", "file_sha256_16": sha256_of(path), "why_not_measured": why,
                              "agrees_with": os.path.relpath(summary, ROOT) if summary else None, "pmc_summary_sha256_16": sha256_of(summary) if summary else None,
                              "taken_on_kernel_source_id": taken_on, "this_build_kernel_source_id": src_id,
                              "stale": (taken_on != src_id) if taken_on else "unknown (taken before round 5 stamped the sources)"}
        return t.get(key), detail, "REPLAYED from profiles/traffic.json (not measured in this run: %s); see traffic_detail.replayed for the stamp" % why
    except Exception:
        return None, None, None




def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks (one per GPU) ourselves."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(args.gpus):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    # any nonzero exit -- also a negative one (a rank killed by a signal after a GPU fault) -- is a failure; when one
    # rank fails the others would sit in a collective until the RCCL timeout, so they are terminated at once
    rc = 0
    live = list(procs)
    while live and rc == 0:
        time.sleep(0.05)
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0:
                rc = code if code > 0 else 128 - code
    for p in live:
        if rc != 0:
            p.terminate()
    for p in live:
        try:
            p.wait(timeout=30)
        except subprocess.TimeoutExpired:
            p.kill()
    sys.exit(rc)


def dist_dry_run(args, r, rank, world, local_rank, use_dist, test_one_gpu, W, H, spp_total, batch, per_spp_bytes):
    """bench.py --dist-dry-run [--gpus N]: the first real multi-GPU run should be boring.  No benchmark: every rank
    reports its device, its free memory, the scratch its share of the workload will take, peer access to the devices it can see; the ranks
    run the collectives the timed loop uses (an all_reduce that must equal the world size, a full-size framebuffer reduce to rank 0,
    timed) and one render of their share at 1/16 of the samples; rank 0 runs the C++ host's combine as a probe (ssx_rccl_probe:
    ncclCommInitAll over every visible device).  Whatever fails is written into the line, which is printed in any case."""
    import ctypes as C
    import torch
    import torch.distributed as dist
    from simple_spectral_amd import _capi
    rep = {"rank": rank, "local_rank": local_rank, "hostname": socket.gethostname(), "visible_devices": torch.cuda.device_count()}
    try:
        props = torch.cuda.get_device_properties(local_rank)
        free_b, total_b = torch.cuda.mem_get_info(local_rank)
        rep.update({"device": props.name, "device_uuid": str(getattr(props, "uuid", "")), "free_bytes": int(free_b), "total_bytes": int(total_b),
                    "peer_access": {str(k): bool(torch.cuda.can_device_access_peer(local_rank, k)) for k in range(torch.cuda.device_count()) if k != local_rank},
                    "scratch_after_upload": r.scratch_info(),
                    "sample_bytes_needed": int(per_spp_bytes * (batch or spp_total)), "spp_per_launch": batch or spp_total, "spp_total": spp_total,
                    "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")})
    except Exception as e:  # noqa: BLE001
        rep["device_error"] = "%s: %s" % (type(e).__name__, str(e)[:300])
    try:
        small = Renderer_share(r, args, rank, world, local_rank, W, H, max(1, spp_total // 16))
        rep.update(small)
    except Exception as e:  # noqa: BLE001
        rep["render_error"] = "%s: %s" % (type(e).__name__, str(e)[:300])
    coll = {}
    if use_dist:
        dev = "cpu" if test_one_gpu else "cuda"
        try:
            one = torch.ones(1, dtype=torch.float32, device=dev)
            dist.all_reduce(one)
            coll["all_reduce_ones"] = float(one[0])
            img = torch.full((H, W, 4), float(rank + 1), dtype=torch.float32, device=dev)
            times = []
            for _ in range(3):
                if dev == "cuda":
                    torch.cuda.synchronize()
                dist.barrier()
                t = time.perf_counter()
                dist.reduce(img, dst=0, op=dist.ReduceOp.SUM)
                if dev == "cuda":
                    torch.cuda.synchronize()
                times.append(round((time.perf_counter() - t) * 1e3, 3))
                img.fill_(float(rank + 1))
            coll["framebuffer_reduce_ms"] = times
            coll["framebuffer_bytes"] = H * W * 16
        except Exception as e:  # noqa: BLE001
            coll["error"] = "%s: %s" % (type(e).__name__, str(e)[:300])
        gathered = [None] * world
        try:
            dist.all_gather_object(gathered, rep)
        except Exception as e:  # noqa: BLE001
            gathered = [rep]
            coll["gather_error"] = "%s: %s" % (type(e).__name__, str(e)[:300])
    else:
        gathered = [rep]
    if rank == 0:
        probe = None
        try:
            lib = _capi.hip_lib()
            buf = C.create_string_buffer(16384)
            n_ok = lib.ssx_rccl_probe(buf, len(buf))
            probe = json.loads(buf.value.decode() or "{}")
            probe["returned"] = n_ok
        except Exception as e:  # noqa: BLE001
            probe = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
        devs = [(x or {}).get("device_uuid") for x in gathered]
        print(json.dumps({"dry_run": True, "n_gpus": world, "backend": dist.get_backend() if use_dist else None, "ranks": gathered, "collectives": coll,
                          "devices_distinct": len(set(devs)) == len(devs), "rccl_probe_single_process": probe,
                          "expect": {"all_reduce_ones": float(world)}}), flush=True)
    if use_dist:
        try:
            dist.barrier(); dist.destroy_process_group()
        except Exception:
            pass


def Renderer_share(r, args, rank, world, local_rank, W, H, spp_small):
    """one render of this rank's share at `spp_small` samples per pixel, timed: the kernels launch, the tile split is what the bench uses"""
    import torch
    from simple_spectral_amd import Options, Renderer
    rs = Renderer(Options(scene_name=args.scene, res=(W, H), spp=spp_small, texture=args.texture, device=local_rank, tile_first=rank, tile_stride=world,
                          tile_skew=1 if world > 1 else 0, seed=0, observer=args.observer, uplift=args.uplift))
    out = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream()
    rs.render_device(out.data_ptr(), st.cuda_stream); torch.cuda.synchronize()
    t = time.perf_counter()
    rs.render_device(out.data_ptr(), st.cuda_stream); torch.cuda.synchronize()
    ms = (time.perf_counter() - t) * 1e3
    own = int((out[..., 3] != 0).sum())
    return {"share_render_ms": round(ms, 3), "share_spp": spp_small, "share_pixels_nonzero": own, "share_scratch": rs.scratch_info(), "kernel": rs.plan_info().get("kernel")}


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
    ap.add_argument("--texture", default="crystal-lizard-512.png", help="PNG under data/scenes, or procedural:N[:SEED]")
    ap.add_argument("--batch", type=int, default=0, help="spp per pipelined batch (0 = library default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true", help="skip the oracle check of the timed image (A/B scripts; the driver's command never passes it)")
    ap.add_argument("--no-pmc", action="store_true", help="do not measure HBM traffic with rocprofv3 --pmc passes in this run (replay profiles/traffic.json, stamped)")
    ap.add_argument("--scratch-cap-gb", type=float, default=16.0, help="per-rank cap on the library's device scratch (sample arrays + level logs): a render that would need more is split into batches of fewer samples per pixel (the default workload needs 4.4 GB and is not split)")
    ap.add_argument("--dist-dry-run", action="store_true", help="no benchmark: go through the multi-GPU plumbing on whatever devices exist (process group, a timed framebuffer reduce, ssx_rccl_probe, per-rank scratch) and REPORT what was found in one JSON line")
    ap.add_argument("--update-traffic", action="store_true", help="also write the measured HBM traffic into the tracked replay file profiles/traffic.json (a round's closing run)")
    ap.add_argument("--quick", action="store_true", help="= --no-cpu-baseline --no-check --no-pmc (A/B and profiling scripts)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)  # the run under rocprofv3 --pmc: renders, prints nothing
    args = ap.parse_args()
    if args.pmc_child or args.quick:
        args.no_check = args.no_cpu_baseline = args.no_pmc = True

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)

    import torch
    import torch.distributed as dist

    from simple_spectral_amd import Options, Renderer

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: simple_spectral_amd has no CPU path")
    # SSX_BENCH_TEST_ONE_GPU=1: plumbing test of the N>1 path on a 1-GPU box (all ranks share
    # device 0 and the reduce goes through gloo on host copies); never set by the driver.
    test_one_gpu = os.environ.get("SSX_BENCH_TEST_ONE_GPU") == "1"
    if test_one_gpu:
        local_rank = 0
    # A launcher that gives every rank its own visible device (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES per process) leaves each
    # rank with ONE device, number 0, whatever its LOCAL_RANK: take the device that exists.  Fewer devices than ranks otherwise: say so.
    n_dev = torch.cuda.device_count()
    if local_rank >= n_dev:
        if n_dev == 1 and world > 1:
            local_rank = 0
        else:
            raise SystemExit("bench.py: LOCAL_RANK %d but %d visible device(s)" % (local_rank, n_dev))
    torch.cuda.set_device(local_rank)
    # SSX_BENCH_FORCE_DIST=1: with --gpus 1 still initialise RCCL (world size 1) and run the framebuffer reduce on
    # the device buffer inside the timed loop -- the N>1 code path (process group on this device, reduce on torch's
    # stream and buffer next to libssx_hip.so) executed on a 1-GPU box.
    force_dist = os.environ.get("SSX_BENCH_FORCE_DIST") == "1" and world == 1
    use_dist = world > 1 or force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if force_dist and "MASTER_PORT" not in os.environ:
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        if test_one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    W = H = args.res
    spp_total = args.spp * world
    texture = args.texture
    # what this rank's render keeps on the device: 48 B per sample of a launch (+ 5 words per 64 samples) and the persistent waves' level
    # logs (1.2 / 2.4 GB whatever the launch renders).  Above --scratch-cap-gb the render is split into batches of fewer samples per pixel
    # (the pixel sums continue from batch to batch: same image); the line's `ranks` carry what each rank really allocated.
    n_tiles_all = ((W + 7) // 8) * ((H + 7) // 8)
    tiles_mine = (n_tiles_all - rank + world - 1) // world
    log_bytes_est = 2.5e9 if args.scene == "plane-srgb" else 1.25e9
    per_spp_bytes = max(tiles_mine, 1) * 64 * (48 + 5 * 4 / 64.0)
    batch = args.batch
    fit_spp = int(max(1, (args.scratch_cap_gb * 2**30 - log_bytes_est) // per_spp_bytes))
    if fit_spp < spp_total and (batch == 0 or batch > fit_spp):
        batch = fit_spp
    r = Renderer(Options(scene_name=args.scene, res=(W, H), spp=spp_total, texture=texture, device=local_rank,
                         tile_first=rank, tile_stride=world, tile_skew=1 if world > 1 else 0, seed=0, observer=args.observer, uplift=args.uplift,
                         spp_per_launch=batch))
    if args.dist_dry_run:
        return dist_dry_run(args, r, rank, world, local_rank, use_dist, test_one_gpu, W, H, spp_total, batch, per_spp_bytes)
    out = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream()

    def reduce_to_rank0(img):
        # the one exchange step of the path: sum of the per-rank framebuffers (RCCL over xGMI)
        if test_one_gpu:
            h = img.cpu()
            dist.reduce(h, dst=0, op=dist.ReduceOp.SUM)
            img.copy_(h)
        else:
            dist.reduce(img, dst=0, op=dist.ReduceOp.SUM)

    def step():
        r.render_device(out.data_ptr(), stream.cuda_stream)
        if use_dist:
            reduce_to_rank0(out)

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # rank 0's copy of the combined image in pinned host memory: part of every timed step.  Two device images and two
    # host images, the copies on a stream of their own: step k's image travels while step k+1 renders (the render stream waits for
    # the copy that last read the image it is about to overwrite); the closing fence waits for the last copy.
    outs = [out, torch.zeros_like(out)]
    host_imgs = [torch.empty((H, W, 4), dtype=torch.float32, pin_memory=True) for _ in range(2)] if rank == 0 else None
    copy_stream = torch.cuda.Stream() if rank == 0 else None
    img_ready = [torch.cuda.Event() for _ in range(2)]
    img_copied = [None, None]

    def host_step(k):
        # one step of the metric: render [+ reduce] into image k & 1, then its copy to the host behind it on the copy stream
        img = outs[k & 1]
        if img_copied[k & 1] is not None:
            stream.wait_event(img_copied[k & 1])
        r.render_device(img.data_ptr(), stream.cuda_stream)
        if use_dist:
            reduce_to_rank0(img)
        if rank == 0:
            img_ready[k & 1].record(stream)
            copy_stream.wait_event(img_ready[k & 1])
            with torch.cuda.stream(copy_stream):
                host_imgs[k & 1].copy_(img, non_blocking=True)
                img_copied[k & 1] = torch.cuda.Event()
                img_copied[k & 1].record(copy_stream)

    for k in range(args.warmup):  # the same step as the timed ones (the first copy on a new stream sets up its queue)
        host_step(k)
    fence()
    r.set_timing(True)  # HIP events on the launch stream around each kernel of the pipeline
    sums0 = r.sums_info()
    t0 = time.perf_counter()
    for k in range(args.steps):
        host_step(k)
    fence()
    elapsed = time.perf_counter() - t0
    elapsed_local = elapsed
    sums1 = r.sums_info()
    last_host_img = host_imgs[(args.steps - 1) & 1].numpy().copy() if rank == 0 and args.steps > 0 else None  # what the last timed step delivered
    stage_ms = {k: v / max(args.steps, 1) for k, v in r.get_timing().items()}
    pipeline_ms = sum(stage_ms.values())  # the kernels of a step, first to last (events without a system fence: csrc/ssx_api.hip timing_events)
    r.set_timing(False)
    # The same K steps without the copy to the host (rounds 1-3 reported this figure as `value`): value_device_resident.
    fence()
    t1 = time.perf_counter()
    for k in range(args.steps):
        step()
    fence()
    elapsed_resident = time.perf_counter() - t1
    # The integrator is two kernels since round 3: ssx_generate_kernel* (camera rays AND their closest hits, traced
    # coherently) and the path megakernel (everything behind the first hit).  The algorithmic flop figure of SURVEY 8(d)
    # covers both, so the roofline is quoted over both durations (rocprofv3: the two rows of kernel_stats.csv).
    path_ms = stage_ms["path"]
    kernel_ms = stage_ms["path"] + stage_ms["generate"]
    if use_dist:
        tt = torch.tensor([elapsed, kernel_ms, elapsed_resident, path_ms], dtype=torch.float64, device="cpu" if test_one_gpu else "cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed, kernel_ms, elapsed_resident, path_ms = float(tt[0]), float(tt[1]), float(tt[2]), float(tt[3])
    # ---- who took part: one record per rank, gathered on rank 0; a run in which two ranks share a device, or in which
    # two ranks' images overlap (a pixel nonzero on both: the partition is wrong), does not print a bench line.
    my_ms, my_path_ms = elapsed_local / max(args.steps, 1) * 1e3, stage_ms["path"]
    n_tiles_img = ((W + 7) // 8) * ((H + 7) // 8)
    my_tiles = (n_tiles_img - rank + world - 1) // world
    sums = {k: sums1[k] - sums0[k] for k in sums0}
    n_units = max(1, sums.get("units", 0))                                                # work units of the timed steps: the library's own count (ssx_units_info)
    props = torch.cuda.get_device_properties(local_rank)
    scratch = r.scratch_info()
    me = {"rank": rank, "local_rank": local_rank, "hostname": socket.gethostname(), "pid": os.getpid(),
          "device": props.name, "device_uuid": str(getattr(props, "uuid", "")), "pci_bus_id": "%04x:%02x:%02x" % (getattr(props, "pci_domain_id", 0), getattr(props, "pci_bus_id", 0), getattr(props, "pci_device_id", 0)),
          "ms_per_step": round(my_ms, 3), "path_ms": round(my_path_ms, 3), "generate_ms": round(stage_ms["generate"], 3),
          "tiles_owned": my_tiles, "units_parked_frac": round(sums["units_parked"] / float(n_units), 4),
          "device_scratch_bytes": scratch, "device_scratch_total": int(sum(scratch.values())) if isinstance(scratch, dict) else None}
    ranks = [me]
    overlap = None
    evidence_error = None
    if use_dist:
        # Every rank takes the same path through the collectives: the work that can fail on ONE rank -- an extra render, its
        # buffers -- comes first, without a collective in it; the ranks then agree on an ok flag (MIN), and only if every rank is fine do they
        # gather the records and sum the row masks.  A rank that failed reports why; nobody is left waiting in a collective the other skipped.
        local_error, nz = None, None
        rows = sorted({0, H // 2, H - 1})
        try:
            # the un-reduced image of this rank on three sampled rows: the nonzero sets must be pairwise disjoint across ranks
            own = torch.zeros_like(out)
            r.render_device(own.data_ptr(), stream.cuda_stream)
            torch.cuda.synchronize()
            nz = (own[rows] != 0).any(dim=-1).to(torch.int32)                        # [rows, W]
            del own
        except Exception as e:  # noqa: BLE001
            local_error = "%s: %s" % (type(e).__name__, str(e)[:300])
        okf = torch.tensor([0 if local_error else 1], dtype=torch.int32, device="cpu" if test_one_gpu else "cuda")
        try:
            dist.all_reduce(okf, op=dist.ReduceOp.MIN)
            all_ok = int(okf[0]) == 1
            gathered = [None] * world
            dist.all_gather_object(gathered, dict(me, evidence_error=local_error))
            ranks = gathered
            if all_ok:
                tot = nz.clone() if not test_one_gpu else nz.cpu()
                dist.all_reduce(tot, op=dist.ReduceOp.SUM)
                overlap = {"rows": rows, "pixels_nonzero_on_more_than_one_rank": int((tot > 1).sum()), "pixels_nonzero_on_some_rank": int((tot > 0).sum()), "row_pixels": len(rows) * W}
            else:
                evidence_error = "; ".join("rank %d: %s" % (x["rank"], x["evidence_error"]) for x in gathered if x and x.get("evidence_error")) or "a rank failed"
        except Exception as e:  # noqa: BLE001 -- (a backend without all_gather_object, say: every rank raises at the same call)
            evidence_error = "%s: %s" % (type(e).__name__, str(e)[:300])
    if os.environ.get("SSX_BENCH_DUMP"):  # tests: rank 0's combined image
        if rank == 0:
            import numpy as np
            np.save(os.environ["SSX_BENCH_DUMP"], out.cpu().numpy())

    if args.pmc_child:
        if use_dist:
            dist.barrier(); dist.destroy_process_group()
        return
    refusal = None
    check = None
    devs = []
    if rank == 0:
        # ---- the timed image is a checked image: no line for an image that differs from the oracle
        if not args.no_check:
            from simple_spectral_amd import textures as _tx
            check = oracle_check(last_host_img, args.scene, W, H, spp_total, args.observer, _tx.resolve(texture)) if args.uplift == "ours" else {"skipped": "uplift %s: the check needs the model the renderer fitted" % args.uplift}
            if check.get("differing_floats"):
                refusal = "bench.py: the timed image differs from the CPU oracle (%d floats on %d tiles): no bench line" % (check["differing_floats"], check["tiles"])
        devs = [(x["hostname"], x["device_uuid"] or x["pci_bus_id"]) for x in ranks]
        if not refusal and len(ranks) == world and len(set(devs)) != len(devs) and not test_one_gpu:
            refusal = "bench.py: two ranks report the same device: %s" % devs
        if not refusal and overlap and overlap["pixels_nonzero_on_more_than_one_rank"]:
            refusal = "bench.py: rank images overlap on %d sampled pixels: the tile partition is wrong" % overlap["pixels_nonzero_on_more_than_one_rank"]
    # rank 0's verdict reaches every rank BEFORE anybody leaves (a rank 0 that raised left the others in the closing barrier until
    # the launcher's timeout): one broadcast word, then every rank exits with the same code
    if use_dist:
        verdict = torch.tensor([1 if refusal else 0], dtype=torch.int32, device="cpu" if test_one_gpu else "cuda")
        dist.broadcast(verdict, src=0)
        if int(verdict[0]):
            dist.destroy_process_group()
            raise SystemExit(refusal or "bench.py: rank 0 refused to print a bench line (its message is on rank 0's stderr)")
    elif refusal:
        raise SystemExit(refusal)
    if rank == 0:
        samples_per_step = W * H * spp_total
        value = samples_per_step * args.steps / elapsed / 1e6
        per_gpu_samples = W * H * args.spp
        flop = FLOP_PER_SAMPLE.get(args.scene, 1.28e4)
        achieved_tflops = per_gpu_samples * flop / (kernel_ms * 1e-3) / 1e12
        plan = r.plan_info()
        L = plan["frames_per_sample"]  # continued levels per sample, measured on this scene at upload
        hbm_bytes = per_gpu_samples * design_bytes_per_sample(args.scene, levels=L, plan=plan)
        traffic, traffic_detail, traffic_source = measured_traffic(args, world)
        info = r.kernel_info()
        n1 = n1_reference()
        line = {
            "metric": "Msamples/s (w*h*spp/s) %s %dx%d" % (args.scene, W, H),
            "value": round(value, 2), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / max(args.steps, 1) * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            # `value`: every step ends with the combined image in (pinned) host memory on rank 0 -- the metric as SURVEY 8(d) defines it
            # ("framebuffer reduce + D2H of XYZA included"); value_device_resident: the same steps without that copy (what rounds 1-3
            # reported as `value`; value_host_inclusive is kept as an alias of `value` for readers of those rounds)
            "value_device_resident": round(samples_per_step * args.steps / elapsed_resident / 1e6, 2),
            "ms_per_step_device_resident": round(elapsed_resident / max(args.steps, 1) * 1e3, 3),
            "value_host_inclusive": round(value, 2),
            "config": {"workload": "%s %dx%d spp=%d/GPU (total spp %d) CIE%d uplift=%s hero-wavelength megakernel" % (args.scene, W, H, args.spp, spp_total, args.observer, args.uplift),
                       "parallelism": "tile-split x%d (round-robin over the tile list, rows rotated: tile_skew 1) + RCCL reduce" % world if world > 1 else ("single GPU + RCCL reduce (world size 1, SSX_BENCH_FORCE_DIST)" if force_dist else "single GPU"),
                       "texture": texture, "seed": 0, "kernel_source_id": kernel_source_id()},
            "roofline": {"bound": "valu", "achieved": round(achieved_tflops, 3), "peak": PEAK_VALU_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved_tflops / PEAK_VALU_TFLOPS, 4),
                         "frac_without_eliminated_work": round(achieved_tflops * (1.0 - FLOP_ELIMINATED_PER_SAMPLE.get(args.scene, 0.0) / flop) / PEAK_VALU_TFLOPS, 4),
                         "traffic": traffic,
                         "traffic_source": traffic_source,
                         "traffic_detail": traffic_detail,
                         "kernel": ("%s (makes its samples in its refill: no generate kernel)" if str(plan.get("samples_made_in", "")).startswith("path kernel") else "ssx_generate_kernel + %s") % (plan.get("kernel") or "ssx_render_kernel"),
                         "kernel_ms": round(kernel_ms, 3), "path_kernel_ms": round(path_ms, 3),
                         "pipeline_ms": round(pipeline_ms, 3), "stage_ms": {k: round(v, 3) for k, v in stage_ms.items()},
                         "flop_per_sample": flop,
                         "note": "FP32 VALU-issue bound (no MFMA: traversal/sampling); peak = 157.3/2 TFLOP/s because the parity contract forbids FMA contraction. HBM is busy but not the limiter: see hbm.  Instruction-level account of the path kernel (regions x measured trip counts x measured opcode rates; pair-aware issue time 80 % of the kernel's time): profiles/r05/isa_census.txt, profiles/r05/NOTES.md.",
                         "hbm": {  # two yardsticks for the counter traffic: SURVEY 8(d)'s all-state-on-chip megakernel, and this design's own bytes
                                 "algorithmic_bytes_per_sample_8d": round(algorithmic_bytes_per_sample_8d(args.scene, spp_total), 3),
                                 "traffic_over_algorithmic_8d": round(traffic / (per_gpu_samples * algorithmic_bytes_per_sample_8d(args.scene, spp_total)), 1) if traffic else None,
                                 "why": "the recursion's levels and next-event terms are logged to HBM and folded post-order at the end of each work unit: the price of reproducing the reference's float rounding (radiance = direct + ((L_next * n.l) * f_s) / pdf, innermost first) with samples, not pixels, as the parallel unit; HBM stays at ~20 % of peak and is not the limiter (VALU issue is)",
                                 "design_bytes_per_sample": round(design_bytes_per_sample(args.scene, levels=L, plan=plan), 1),
                                 "path_kernel_design_bytes_per_sample": round(design_bytes_per_sample(args.scene, True, L, plan), 1),
                                 "traffic_over_path_kernel_design": round(traffic / (per_gpu_samples * design_bytes_per_sample(args.scene, True, L, plan)), 3) if traffic else None,
                                 "traffic_bytes_per_sample": round(traffic / per_gpu_samples, 1) if traffic else None,
                                 "design_GBps": round(hbm_bytes / (pipeline_ms * 1e-3) / 1e9, 1),
                                 "measured_GBps": round(traffic / (path_ms * 1e-3) / 1e9, 1) if traffic else None,
                                 "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                 "frac_design": round(hbm_bytes / (pipeline_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                                 "frac_measured": round(traffic / (path_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4) if traffic else None},
                         "device_scratch_bytes": r.scratch_info(),
                         "vgprs": info["vgprs"], "scratch_bytes": info["scratch_bytes"], "lds_bytes": info["lds_bytes"],
                         "plan": plan},
        }
        line["check"] = check
        line["ranks"] = ranks
        line["distributed"] = {"backend": dist.get_backend() if use_dist else None, "world_size": dist.get_world_size() if use_dist else 1,
                               "devices_distinct": (len(set(devs)) == len(devs)) if len(ranks) == world else None, "overlap": overlap, "evidence_error": evidence_error,
                               "slowest_rank_ms": max(x["ms_per_step"] for x in ranks), "fastest_rank_ms": min(x["ms_per_step"] for x in ranks)}
        # weak-scaling efficiency against a stated N = 1 figure (the driver computes its own from its per-N runs; this one names what it used)
        line["efficiency_vs_n1_reference"] = {"value": round(value / (world * n1["value"]), 4) if n1["value"] else None, "n1_value": n1["value"], "n1_source": n1["source"],
                                              "kernel_source_id": n1.get("kernel_source_id"), "refused": n1.get("refused")}
        if world == 1 and not args.no_cpu_baseline:
            from simple_spectral_amd import textures
            line["cpu_baseline"] = cpu_baseline(args.scene, W, H, textures.resolve(texture))  # "procedural:N[:SEED]" -> the same texels the GPU run used
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
