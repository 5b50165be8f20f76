"""Bounce reordering (path_sort.hip): per-kernel time of the bench frame for a list of sort keys, and a bit-compare of
every frame with the unsorted one.   VMX_LIB=build/libvermilion_hip_ab.so python tools/sort_probe.py [spp] [mode,mode,...]
(the experiment lives in the A/B library: make -C vermilion_amd/csrc ab)
mode = obits | dbits << 4 | dir_major << 8 | chunk_log2 << 12 | shade_sorted << 20   (hex accepted)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, vermilion_amd as va
from vermilion_amd import scenes
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 256
modes = [int(m, 0) for m in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 0x35, 0x45, 0x135, 0x30, 0x05]
sampling = int(os.environ.get("SAMPLING", "0"))
pos, nrm, uv = scenes.sponza260k(); c = scenes.sponza_camera()
W, H = 1920, 1080
cam = va.make_camera(c["position"], c["rotation_deg"], W, H, spp, back_size=(3.6, 3.6 * H / W))
sc = va.Scene(pos, nrm, uv, builder=int(os.environ.get("BUILDER", "0")))
out = torch.empty((H, W, 5), device="cuda")
ref = None
for m in modes:
    o = va.make_opts(seed=1, early_stop=False, sampling=sampling, reorder=m)
    sc.render_device(cam, o, out.data_ptr())
    st = sc.render_device(cam, o, out.data_ptr()); torch.cuda.synchronize()
    t = sc.timings()
    img = out.cpu().numpy().view(np.uint32)
    if ref is None: ref = img.copy()
    same = bool(np.array_equal(ref, img))
    k = {n: round(v["ms"], 2) for n, v in t.items() if v["launches"]}
    print(f"mode {m:#08x}: frame {st['ms_device']:.2f} ms  bounce+tail {k.get('trace_bounce', 0) + k.get('tail', 0):.2f}  {k}  identical={same}", flush=True)
