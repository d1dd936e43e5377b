import sys, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import oracle_lib as ol
from simple_spectral_amd import Options, Renderer
def bits(a): return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
for scene, res, jr, W, H, spp in (("cornell-srgb", "test-img.png", 16, 48, 40, 5), ("cornell-srgb", "test-img.png", 64, 48, 40, 5), ("plane-srgb", "test-img.png", 16, 48, 40, 5), ("cornell-srgb", "test-img.png", 16, 64, 64, 4)):
    r = Renderer(Options(scene_name=scene, res=(W, H), spp=spp, seed=4, texture=res, uplift="jh", jh_res=jr))
    r.render_start(); r.render_wait()
    ref = ol.Oracle(scene, texture=res, jh=r.scene.jh_model()).render(W, H, spp, seed=4)
    d = bits(r.xyza) != bits(ref)
    print(scene, jr, W, H, spp, "diff floats", d.sum(), "pixels", d.any(axis=2).sum(), "gpu alpha mean %.3f ref alpha mean %.3f" % (r.xyza[..., 3].mean(), ref[..., 3].mean()), "nan gpu", np.isnan(r.xyza).sum(), "nan ref", np.isnan(ref).sum())
    if d.any():
        jj, ii = np.argwhere(d.any(axis=2))[0]
        print("  first bad", ii, jj, r.xyza[jj, ii], ref[jj, ii])
