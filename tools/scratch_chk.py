import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch, vermilion_amd as va
from vermilion_amd import scenes
pos, nrm, uv = scenes.sponza260k(); c = scenes.sponza_camera()
W, H, spp = 1920, 1080, 256
cam = va.make_camera(c["position"], c["rotation_deg"], W, H, spp, back_size=(3.6, 3.6 * H / W))
sc = va.Scene(pos, nrm, uv)
out = torch.empty((H, W, 5), device="cuda")
o = va.make_opts(seed=1, early_stop=False, sampling=va.VMX_SAMPLING_ELIDE_DEAD)
imgs = []
for i in range(4):
    out.zero_()
    st = sc.render_device(cam, o, out.data_ptr()); torch.cuda.synchronize()
    imgs.append(out.cpu().numpy().view(np.uint32).copy())
    print(i, st["ms_device"], st["rays_primary"], st["rays_secondary"], {n: round(v["ms"], 2) for n, v in sc.timings().items() if v["launches"]}, flush=True)
o0 = va.make_opts(seed=1, early_stop=False)
st = sc.render_device(cam, o0, out.data_ptr()); torch.cuda.synchronize()
ref = out.cpu().numpy().view(np.uint32)
print([bool(np.array_equal(ref, im)) for im in imgs])
