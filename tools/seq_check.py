"""Sequences of frames with different options on one scene: every frame must equal the first frame rendered with the same
early-stop setting, and report the same ray counts as the first frame with the same options.   python tools/seq_check.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, vermilion_amd as va
from vermilion_amd import scenes
pos, nrm, uv = scenes.sponza260k(); c = scenes.sponza_camera()
W, H, spp = 1920, 1080, int(sys.argv[1]) if len(sys.argv) > 1 else 256
cam = va.make_camera(c["position"], c["rotation_deg"], W, H, spp, back_size=(3.6, 3.6 * H / W))
sc = va.Scene(pos, nrm, uv)
out = torch.empty((H, W, 5), device="cuda")
stream = torch.cuda.current_stream().cuda_stream if len(sys.argv) > 2 else None
E = va.VMX_SAMPLING_ELIDE_DEAD
seq = [(False, "count"), (False, 0), (True, 0), (False, 0), (False, E), (True, 0), (True, E), (False, E), (False, E), (False, E), (True, E), (True, E), (False, 0), (False, E)]
ref, cnt, bad = {}, {}, 0
for i, (es, samp) in enumerate(seq):
    if samp == "count":
        sc.render_device(cam, va.make_opts(seed=1, early_stop=es, collect_counters=True), out.data_ptr())
        continue
    o = va.make_opts(seed=1, early_stop=es, sampling=samp)
    st = sc.render_device(cam, o, out.data_ptr(), stream) if stream is not None else sc.render_device(cam, o, out.data_ptr())
    torch.cuda.synchronize()
    img = out.cpu().numpy().view(np.uint32)
    same = True
    if es in ref: same = bool(np.array_equal(ref[es], img))
    else: ref[es] = img.copy()
    rays = (st["rays_primary"], st["rays_secondary"])
    okc = cnt.setdefault((es, samp), rays) == rays
    bad += (not same) or (not okc)
    print(f"{i:2d} es={int(es)} sampling={samp:#x}: {st['ms_device']:.2f} ms rays {rays} same_frame={same} same_counts={okc}", flush=True)
print("bad:", bad)
