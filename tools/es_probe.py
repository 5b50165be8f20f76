import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vermilion_amd as va
from vermilion_amd import scenes
pos, nrm, uv = scenes.sponza260k(); c = scenes.sponza_camera()
W, H, spp = 1920, 1080, 256
cam = va.make_camera(c["position"], c["rotation_deg"], W, H, spp, back_size=(3.6, 3.6 * H / W))
sc = va.Scene(pos, nrm, uv)
out = torch.empty((H, W, 5), device="cuda")
o = va.make_opts(seed=1, early_stop=True)
for r in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); st = sc.render_device(cam, o, out.data_ptr()); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
    print(f"wall {dt:.2f} ms dev {st['ms_device']:.2f} passes {st['passes']} launches {st['kernel_launches']} rays {st['rays_primary']/1e6:.1f}+{st['rays_secondary']/1e6:.1f}M")
    print({k: (round(v["ms"], 2), v["launches"]) for k, v in sc.timings().items() if v["launches"]})
