import sys
sys.path.insert(0, "/root/repo")
import vermilion_amd as va
from vermilion_amd import scenes
pos, nrm, uv = scenes.sponza260k(); c = scenes.sponza_camera()
W, H = 1920, 1080
sc = va.Scene(pos, nrm, uv)
prev = None
for spp in (3, 4, 5, 6, 8, 12, 16, 32, 64, 256):
    cam = va.make_camera(c["position"], c["rotation_deg"], W, H, spp, back_size=(3.6, 3.6 * H / W))
    img, st = sc.render_bruteforce(cam, va.make_opts(seed=1))
    print(spp, st["samples"], "avg", st["samples"] / (W * H), "ms", round(st["ms_device"], 2))
