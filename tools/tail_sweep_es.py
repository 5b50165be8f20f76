"""The plugin's default frame (early stop on) against vmx_opts.tail_threshold, plain and VMX_SAMPLING_ELIDE_DEAD."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vermilion_amd as va
from vermilion_amd import scenes
pos, nrm, uv = scenes.sponza260k(); c = scenes.sponza_camera()
W, H, spp = 1920, 1080, 256
cam = va.make_camera(c["position"], c["rotation_deg"], W, H, spp, back_size=(3.6, 3.6 * H / W))
sc = va.Scene(pos, nrm, uv)
out = torch.empty((H, W, 5), device="cuda")
for samp, name in ((0, "plain"), (va.VMX_SAMPLING_ELIDE_DEAD, "elided")):
    for thr in (0, 4 << 20, 2 << 20, 1 << 20, 512 << 10, 128 << 10, 1):
        o = va.make_opts(seed=1, early_stop=True, sampling=samp, tail_threshold=thr)
        sc.render_device(cam, o, out.data_ptr())
        ms = 0.0; acc = {}
        for _ in range(5):
            st = sc.render_device(cam, o, out.data_ptr()); ms += st["ms_device"] / 5
            for k, v in sc.timings().items():
                if v["launches"]:
                    a = acc.setdefault(k, [0.0, 0]); a[0] += v["ms"] / 5; a[1] = v["launches"]
        print(f"{name:7s} tail_threshold {thr >> 10:6d} K: {ms:6.2f} ms passes {st['passes']} launches {st['kernel_launches']} | " + " ".join(f"{k} {v[0]:.2f}x{v[1]}" for k, v in acc.items()), flush=True)
sc.close()
