#!/usr/bin/env python3
"""How fast is bench.py's CPU baseline (kind "port": the oracle's C renderer) against the reference BINARY?

BASELINE.md section 4(i) wants the port within +-10 % of the reference's own rate.  It is not: the port is the same
algorithm without the reference's std::function recursion, virtual dispatch and GLM temporaries.  The reference cannot be
built in this image (GLM absent, CMakeLists.txt:19), so its side of the comparison is the survey's measurement in this
same kind of container (8 x Xeon 2.10 GHz; BASELINE.md section 2: unmodified sources, g++ 11.4 -O2, one thread);
this script measures the port's side here -- same container type, same compiler, one thread and eight -- and writes
both to profiles/r04/port_vs_reference_probe.json, which bench.py's cpu_baseline quotes ("vs_reference_probe").

    python tools/port_vs_reference_probe.py          (CPU only, ~1 min)
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol  # noqa: E402  (TEST INFRASTRUCTURE: the CPU baseline leg)

# BASELINE.md section 2, survey probe of the reference binary: Msamples/s
REFERENCE = {"cornell-srgb": {"one_thread": 0.060, "eight_threads_512_spp16": 0.513},
             "cornell": {"one_thread": 0.071, "eight_threads_512_spp16": 0.582},
             "plane-srgb": {"one_thread": 0.329, "eight_threads_512_spp64": 2.397}}


def rate(o, W, H, spp, rect, nthreads):
    o.render(W, H, 1, rect=rect, nthreads=nthreads)
    t = time.time()
    o.render(W, H, spp, rect=rect, nthreads=nthreads)
    dt = time.time() - t
    npx = (rect[2] - rect[0]) * (rect[3] - rect[1])
    return npx * spp / dt / 1e6, dt


def main():
    model = [ln.split(":")[1].strip() for ln in open("/proc/cpuinfo") if ln.startswith("model name")]
    out = {"host": {"model": model[0] if model else "", "cpus": os.cpu_count()},
           "note": "reference = survey probe of the reference binary (BASELINE.md section 2); port = oracle/libssx_oracle.so timed by this script in the build container",
           "scenes": {}}
    for scene, spp8 in (("cornell-srgb", 16), ("cornell", 16), ("plane-srgb", 64)):
        o = ol.Oracle(scene, texture=os.path.join(ROOT, "data", "scenes", "crystal-lizard-512.png") if scene != "cornell" else None)
        r1, dt1 = rate(o, 512, 512, 96, (224, 224, 288, 288), 1)      # the survey's one-thread driver also walked pixels of the image centre
        r8, dt8 = rate(o, 512, 512, spp8, (0, 0, 512, 512), 8)
        ref = REFERENCE[scene]
        k8 = [k for k in ref if k.startswith("eight")][0]
        out["scenes"][scene] = {"one_thread_ref": ref["one_thread"], "one_thread_port_same_box": round(r1, 4), "port_over_ref_one_thread": round(r1 / ref["one_thread"], 2),
                                "eight_threads_ref": ref[k8], "eight_threads_port_same_box": round(r8, 4), "port_over_ref_eight_threads": round(r8 / ref[k8], 2),
                                "seconds": [round(dt1, 1), round(dt8, 1)]}
        print(scene, out["scenes"][scene], flush=True)
    os.makedirs(os.path.join(ROOT, "profiles", "r04"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "profiles", "r04", "port_vs_reference_probe.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
