"""Per-kernel time of the bench frame in the library's default form (the traversal kernels sort their finished rays) and in
the headline form (every step shaded in full), for A/B libraries: VMX_LIB=build/<lib>.so python tools/sorted_frame.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vermilion_amd as va
from vermilion_amd import scenes
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
pos, nrm, uv = scenes.sponza260k(); c = scenes.sponza_camera()
W, H, spp = 1920, 1080, 256
cam = va.make_camera(c["position"], c["rotation_deg"], W, H, spp, back_size=(3.6, 3.6 * H / W))
sc = va.Scene(pos, nrm, uv)
out = torch.empty((H, W, 5), device="cuda")
ref = None
for name, kw in (("default", {}), ("headline", dict(pipeline=0x100)), ("elided", dict(sampling=va.VMX_SAMPLING_ELIDE_DEAD)),
                 ("early-stop", dict(early_stop=True))):
    kw = dict(dict(early_stop=False), **kw)
    o = va.make_opts(seed=1, **kw)
    sc.render_device(cam, o, out.data_ptr())
    acc, ms = {}, 0.0
    for _ in range(reps):
        st = sc.render_device(cam, o, out.data_ptr())
        ms += st["ms_device"] / reps
        for k, v in sc.timings().items():
            if v["launches"]:
                acc[k] = acc.get(k, 0.0) + v["ms"] / reps
    chk = int(out.view(torch.int32).to(torch.int64).sum().item())
    print(f"{name:10s} {ms:8.3f} ms  checksum {chk}  " + " ".join(f"{k} {v:.3f}" for k, v in acc.items()))
sc.close()
