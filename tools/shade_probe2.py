import os, sys
sys.path.insert(0, "/root/repo")
import torch, vermilion_amd as va
from vermilion_amd import scenes
pos, nrm, uv = scenes.sponza260k(); c = scenes.sponza_camera()
W, H, spp = 1920, 1080, 256
cam = va.make_camera(c["position"], c["rotation_deg"], W, H, spp, back_size=(3.6, 3.6 * H / W))
d = va.default_spheres()
out = torch.empty((H, W, 5), dtype=torch.float32, device="cuda")
for n in (8, 5, 2, 1):
    sub = (va._lib.Sphere * n)(*[d[i] for i in range(n)])
    sc = va.Scene(pos, nrm, uv, spheres=sub)
    for form, name in ((0x100, "headline"), (0, "default")):
        for r in range(2):
            st = sc.render_device(cam, va.make_opts(seed=1, early_stop=False, pipeline=form), out.data_ptr())
        t = sc.timings()
        print(f"spheres {n} {name}: " + " ".join(f"{k} {v['ms']:.2f}" for k, v in t.items() if v["launches"]), f"| rays2 {st['rays_secondary']/1e6:.1f}M")
    sc.close()
