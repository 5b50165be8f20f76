"""BruteForceTracer frame time on the bench scene (k_bruteforce, the engine's default integrator, integrators.cpp:9-186)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np, vermilion_amd as va
from vermilion_amd import scenes
pos, nrm, uv = scenes.sponza260k(); c = scenes.sponza_camera()
W, H = 1920, 1080
sc = va.Scene(pos, nrm, uv)
for spp in (16, 256):
    cam = va.make_camera(c["position"], c["rotation_deg"], W, H, spp, back_size=(3.6, 3.6 * H / W))
    o = va.make_opts(seed=1)
    for r in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        img, st = sc.render_bruteforce(cam, o)
        dt = (time.perf_counter() - t0) * 1e3
    print(f"spp {spp}: wall {dt:.1f} ms (with the copy to the host), device {st['ms_device']:.2f} ms, samples {st['samples'] / 1e6:.1f} M, "
          f"rays {(st['rays_primary'] + st['rays_secondary']) / 1e6:.1f} M -> {(st['rays_primary'] + st['rays_secondary']) / st['ms_device'] / 1e3:.0f} Mrays/s")
