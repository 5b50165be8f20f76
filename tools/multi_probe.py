import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, vermilion_amd as va
from vermilion_amd import scenes
pos, nrm, uv = scenes.sponza260k(); c = scenes.sponza_camera()
W, H, spp = 1920, 1080, 256
cam = va.make_camera(c["position"], c["rotation_deg"], W, H, spp, back_size=(3.6, 3.6 * H / W))
for world in (1, 2, 8):
    with va.MultiScene(pos, nrm, uv, devices=[0] * world) as m:
        o = va.make_opts(seed=1, early_stop=False, stripe_rows=4 if world > 4 else 16)
        m.render(cam, o)
        t0 = time.perf_counter(); img, st = m.render(cam, o); dt = (time.perf_counter() - t0) * 1e3
        print(f"vmx_multi world {world} (all replicas on device 0): wall {dt:.1f} ms incl. the 41 MB copy of the frame to the host; device {st['ms_device']:.1f} ms")
