"""Frames of the bench scene for rocprofv3 --kernel-trace.   python tools/frame_once.py [spp] [sampling] [pipeline] [es]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, vermilion_amd as va
from vermilion_amd import scenes
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 256
sampling = int(sys.argv[2], 0) if len(sys.argv) > 2 else 0
pipe = int(sys.argv[3], 0) if len(sys.argv) > 3 else 0
es = len(sys.argv) > 4 and sys.argv[4] == "es"
pos, nrm, uv = scenes.sponza260k(); c = scenes.sponza_camera()
W, H = 1920, 1080
cam = va.make_camera(c["position"], c["rotation_deg"], W, H, spp, back_size=(3.6, 3.6 * H / W))
sc = va.Scene(pos, nrm, uv)
out = torch.empty((H, W, 5), device="cuda")
o = va.make_opts(seed=1, early_stop=es, sampling=sampling, pipeline=pipe)
for _ in range(3):
    st = sc.render_device(cam, o, out.data_ptr())
torch.cuda.synchronize()
print(st["ms_device"])
