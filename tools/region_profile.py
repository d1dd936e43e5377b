"""Run the region-timing build (tools/_abl/abl_regions.so, -DSSX_PROFILE_REGIONS) once and print
the share of wave cycles per region of the megakernel iteration.  Profiling aid only."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys
sys.path.insert(0, %r)
import torch
from simple_spectral_amd import Options, Renderer
r = Renderer(Options(scene_name=%r, res=(512, 512), spp=64, texture="crystal-lizard-512.png"))
out = torch.zeros((512, 512, 4), device="cuda")
r.render_device(out.data_ptr(), torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
'''
for scene in sys.argv[1:] or ["cornell-srgb"]:
    env = dict(os.environ, SSX_HIP_LIB_OVERRIDE=os.path.join(ROOT, "tools", "_abl", "abl_regions.so"))
    p = subprocess.run([sys.executable, "-c", code % (ROOT, scene)], env=env, capture_output=True, text=True)
    m = re.search(r"\[region profile\] (.*)", p.stderr)
    if not m:
        print(p.stderr[-2000:]); continue
    kv = dict(x.split("=") for x in m.group(1).split())
    kv = {k: int(v) for k, v in kv.items()}
    tot = kv["total_wave_cycles"]
    print("scene", scene, " iterations/wave-unit total:", kv["wave_iterations"], " mean active lanes at loop top: %.1f" % (kv["active_lanes"] / kv["wave_iterations"]))
    print("  cycles per wave-iteration: %.0f" % (tot / kv["wave_iterations"]))
    for k in ("refill", "trace_primary", "hit+albedo", "sample_light", "trace_shadow", "nee_contrib", "bsdf_sample", "frame_push", "finish/fold"):
        print("  %-14s %5.1f %%  (%6.0f cycles/iter)" % (k, 100.0 * kv[k] / tot, kv[k] / kv["wave_iterations"]))
