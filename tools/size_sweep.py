"""Launch time against launch size: the bench frame at 256 ... 8 spp (split pipeline forced), per-kernel device time.
Fixed cost per launch = the intercept; it is what strong scaling and early-stop frames pay."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vermilion_amd as va
from vermilion_amd import scenes
pos, nrm, uv = scenes.sponza260k(); c = scenes.sponza_camera()
W, H = 1920, 1080
sc = va.Scene(pos, nrm, uv)
out = torch.empty((H, W, 5), device="cuda")
for spp in (256, 128, 64, 32, 16, 8):
    cam = va.make_camera(c["position"], c["rotation_deg"], W, H, spp, back_size=(3.6, 3.6 * H / W))
    o = va.make_opts(seed=1, early_stop=False, pipeline=4, tail_threshold=int(os.environ.get("TAIL", 1 << 20)))
    sc.render_device(cam, o, out.data_ptr())
    st = sc.render_device(cam, o, out.data_ptr()); torch.cuda.synchronize()
    t = sc.timings()
    print(f"spp {spp:3d}: camera rays {st['rays_primary'] / 1e6:6.1f} M, bounce rays {st['rays_secondary'] / 1e6:5.1f} M, frame {st['ms_device']:6.2f} ms | " +
          "  ".join(f"{k} {v['ms']:.2f}/{v['launches']}" for k, v in t.items() if v["launches"]))
