"""Shading-kernel probe: the bench frame with the reference sphere table, with only the two light
spheres, and with the first N spheres — how much of k_shade is the sphere table."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import vermilion_amd as va
from vermilion_amd import scenes

def main():
    spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    pos, nrm, uv = scenes.sponza260k()
    c = scenes.sponza_camera()
    W, H = 1920, 1080
    cam = va.make_camera(c["position"], c["rotation_deg"], W, H, spp, back_size=(3.6, 3.6 * H / W))
    d = va.default_spheres()
    for n in (8, 2, 1):
        sub = (va._lib.Sphere * n)(*[d[i] for i in range(n)])
        sc = va.Scene(pos, nrm, uv, spheres=sub)
        out = torch.empty((H, W, 5), dtype=torch.float32, device="cuda")
        for r in range(2):
            st = sc.render_device(cam, va.make_opts(seed=1, early_stop=False), out.data_ptr())
        print(f"spheres {n}: dev {st['ms_device']:.1f} ms primary {st['primary']['ms']:.1f} bounce {st['bounce']['ms']:.1f} "
              f"({st['rays_secondary']/1e6:.1f}M rays) shade {st['shade']['ms']:.1f}")
        sc.close()

main()
