"""Frame time of the bench frame against the tail threshold (live paths below which the remaining bounce generations run in
the fused kernel), default and headline form."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vermilion_amd as va
from vermilion_amd import scenes
pos, nrm, uv = scenes.sponza260k(); c = scenes.sponza_camera()
W, H, spp = 1920, 1080, 256
cam = va.make_camera(c["position"], c["rotation_deg"], W, H, spp, back_size=(3.6, 3.6 * H / W))
sc = va.Scene(pos, nrm, uv)
out = torch.empty((H, W, 5), device="cuda")
for form, name in ((0, "default"), (0x100, "headline")):
    for thr in (0, 8 << 20, 4 << 20, 1 << 20, 256 << 10):
        o = va.make_opts(seed=1, early_stop=False, pipeline=form, tail_threshold=thr)
        sc.render_device(cam, o, out.data_ptr())
        ms = 0.0
        for _ in range(3):
            ms += sc.render_device(cam, o, out.data_ptr())["ms_device"] / 3
        t = sc.timings()
        print(f"{name:9s} tail_threshold {thr >> 20:3d} M: {ms:7.2f} ms  " + " ".join(f"{k} {v['ms']:.2f}x{v['launches']}" for k, v in t.items() if v["launches"] and k in ("trace_bounce", "shade_bounce", "tail")))
sc.close()
