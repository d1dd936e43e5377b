import sys, os, ctypes as C
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
os.environ["SSX_HIP_LIB_OVERRIDE"] = os.path.join(R, "tools", "_abl", "abl_%s.so" % (sys.argv[1] if len(sys.argv) > 1 else "cands"))
import torch
from simple_spectral_amd import Options, Renderer, _capi
r = Renderer(Options(scene_name="cornell-srgb", res=(512, 512), spp=32, texture="crystal-lizard-512.png"))
lib = _capi.hip_lib()
lib.ssx_cand_stats.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
buf = (C.c_ulonglong * 4)()
lib.ssx_cand_stats(buf, 1)
out = torch.zeros((512, 512, 4), device="cuda")
r.render_device(out.data_ptr(), torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize()
lib.ssx_cand_stats(buf, 0)
t, lanes, cands, trips = [int(x) for x in buf]
print("traces (wave-level) %d; lanes with a ray per trace %.1f; candidates per ray %.2f; pass-2 loop trips per trace (max over lanes) %.2f" % (t, lanes / t, cands / lanes, trips / t))
