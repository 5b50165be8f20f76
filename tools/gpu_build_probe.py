"""Scene build time and frame time per BVH builder (0 reference topology, 1 binned SAH on the host, 2 LBVH on the GPU, 3 PLOC on the GPU)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import vermilion_amd as va
from vermilion_amd import scenes

name = sys.argv[1] if len(sys.argv) > 1 else "sponza260k"
gen, camf = scenes.SCENES[name]
pos, nrm, uv = gen()
c = camf()
W, H, spp = 1920, 1080, 64
cam = va.make_camera(c["position"], c["rotation_deg"], W, H, spp, back_size=(3.6, 3.6 * H / W))
out = torch.empty((H, W, 5), dtype=torch.float32, device="cuda")
va.Scene(pos, nrm, uv).close()  # warm the context
for b, label in ((0, "reference"), (1, "SAH host"), (2, "LBVH gpu"), (3, "PLOC gpu")):
    ts = []
    for r in range(3):
        t0 = time.time()
        sc = va.Scene(pos, nrm, uv, builder=b)
        ts.append(time.time() - t0)
        if r < 2: sc.close()
    d = sc.describe()
    for r in range(2):
        st = sc.render_device(cam, va.make_opts(seed=1, early_stop=False, collect_counters=(r == 0)), out.data_ptr())
        if r == 0: iv = (st["primary"]["inner_visits"] + st["bounce"]["inner_visits"]) / (st["rays_primary"] + st["rays_secondary"])
    print(f"{label:10s} build {min(ts)*1e3:7.1f} ms  nodes {d['n_nodes']} depth {d['max_depth']}  frame@64spp {st['ms_device']:.1f} ms "
          f"({(st['rays_primary']+st['rays_secondary'])/st['ms_device']/1e3:.0f} Mrays/s)  inner visits/ray {iv:.1f}")
    sc.close()
