"""VMX_SAMPLING_ELIDE_DEAD: per-kernel time of the bench frame with and without the flag, fixed-count and early-stop,
and a bit-compare of the frames.   python tools/elide_probe.py [spp]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, vermilion_amd as va
from vermilion_amd import scenes
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 256
pos, nrm, uv = scenes.sponza260k(); c = scenes.sponza_camera()
W, H = 1920, 1080
cam = va.make_camera(c["position"], c["rotation_deg"], W, H, spp, back_size=(3.6, 3.6 * H / W))
sc = va.Scene(pos, nrm, uv)
out = torch.empty((H, W, 5), device="cuda")
for es in (False, True):
    ref = None
    for flag, tail, pipe in [(f, int(t), int(p, 0)) for p in os.environ.get("PIPE", "0").split(",") for f in (0, va.VMX_SAMPLING_ELIDE_DEAD)
                             for t in os.environ.get("TAIL", "0").split(",")]:
        o = va.make_opts(seed=1, early_stop=es, sampling=flag, tail_threshold=tail, pipeline=pipe)
        sc.render_device(cam, o, out.data_ptr())
        st = sc.render_device(cam, o, out.data_ptr()); torch.cuda.synchronize()
        t = sc.timings()
        img = out.cpu().numpy().view(np.uint32)
        if ref is None: ref = img.copy()
        same = bool(np.array_equal(ref, img))
        k = {n: round(v["ms"], 2) for n, v in t.items() if v["launches"]}
        print(f"early_stop={es} sampling={flag:#x} tail={tail} pipeline={pipe:#x}: frame {st['ms_device']:.2f} ms  rays {st['rays_primary']}+{st['rays_secondary']}  {k}  identical={same}", flush=True)
