"""Timing-only ablations of the megakernel (run on the GPU box).  Builds variants with parts of
the path replaced by stubs and reports Msamples/s; the images are wrong by construction -- this
is a profiling aid, nothing here is used by the package, the tests or bench.py."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VARIANTS = {"base": [], "no_transcendentals": ["-DSSX_ABL_TRANS"], "no_shadow_trace": ["-DSSX_ABL_NOSHADOW"],
            "no_nee": ["-DSSX_ABL_NONEE"], "no_nee_no_trans": ["-DSSX_ABL_NONEE", "-DSSX_ABL_TRANS"]}

def build():
    from simple_spectral_amd import build as b
    for name, flags in VARIANTS.items():
        out = os.path.join(ROOT, "tools", "_abl", "abl_%s.so" % name)
        os.makedirs(os.path.dirname(out), exist_ok=True)
        subprocess.check_call([b.hipcc()] + b.HIP_FLAGS + flags + b.HIP_SRC + ["-o", out, "-lpthread"])

def run(name, scene):
    code = r'''
import sys, time
sys.path.insert(0, %r)
import torch
from simple_spectral_amd import Options, Renderer
r = Renderer(Options(scene_name=%r, res=(512, 512), spp=256, texture="crystal-lizard-512.png"))
out = torch.zeros((512, 512, 4), device="cuda")
s = torch.cuda.current_stream()
for _ in range(2): r.render_device(out.data_ptr(), s.cuda_stream)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(3): r.render_device(out.data_ptr(), s.cuda_stream)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 3
print("%%-20s %%-13s %%7.2f ms  %%8.1f Msamples/s" %% (%r, %r, dt * 1e3, 512 * 512 * 256 / dt / 1e6))
''' % (ROOT, scene, name, scene)
    env = dict(os.environ, SSX_HIP_LIB_OVERRIDE=os.path.join(ROOT, "tools", "_abl", "abl_%s.so" % name))
    subprocess.check_call([sys.executable, "-c", code], env=env)

if __name__ == "__main__":
    if "--build" in sys.argv:
        build()
    else:
        names = [a for a in sys.argv[1:] if not a.startswith("-")] or list(VARIANTS)
        for name in names:
            for scene in ("cornell-srgb",):
                run(name, scene)
