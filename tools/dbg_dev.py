import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np, torch
import oracle_lib as ol
from simple_spectral_amd import Options, Renderer
def bits(a): return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
for scene, W, H, spp in (("plane-srgb", 40, 24, 6), ("cornell-srgb", 40, 24, 6), ("cornell-srgb", 64, 64, 40)):
    ref = ol.Oracle(scene, texture="test-img.png").render(W, H, spp, seed=2)
    r = Renderer(Options(scene_name=scene, res=(W, H), spp=spp, seed=2, texture="test-img.png"))
    out = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    r.render_device(out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    d = out.cpu().numpy()
    r.render_start(); r.render_wait()
    print(scene, W, H, spp, "device-vs-oracle diff:", (bits(d) != bits(ref)).sum(), " async-vs-oracle diff:", (bits(r.xyza) != bits(ref)).sum())
    bad = np.argwhere((bits(d) != bits(ref)).any(axis=2))
    print("   first bad pixels (j,i):", bad[:8].tolist())
    out2 = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    r.render_device(out2.data_ptr(), torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize()
    print("   second device render diff:", (bits(out2.cpu().numpy()) != bits(ref)).sum())
