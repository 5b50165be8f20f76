"""Where a rank's share of the bench frame loses against full frame / world: per-kernel device time of rank 0 of `world`
(headline form, the one bench.py --gpus N times) beside the full frame's kernels divided by world, and the wall clock of
the call beside its device time (host gaps).  One GPU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vermilion_amd as va
from vermilion_amd import scenes
pos, nrm, uv = scenes.sponza260k(); c = scenes.sponza_camera()
W, H, spp = 1920, 1080, 256
cam = va.make_camera(c["position"], c["rotation_deg"], W, H, spp, back_size=(3.6, 3.6 * H / W))
sc = va.Scene(pos, nrm, uv)
out = torch.empty((H, W, 5), device="cuda")
FORM = int(os.environ.get("FORM", "0x100"), 0)
def run(rank, world, reps=4):
    o = va.make_opts(seed=1, early_stop=False, rank=rank, world=world, stripe_rows=16 if world <= 4 else 4, pipeline=FORM)
    sc.render_device(cam, o, out.data_ptr())
    wall = dev = 0.0; acc = {}
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        st = sc.render_device(cam, o, out.data_ptr()); torch.cuda.synchronize()
        wall += (time.perf_counter() - t0) * 1e3 / reps; dev += st["ms_device"] / reps
        for k, v in sc.timings().items():
            if v["launches"]:
                acc[k] = acc.get(k, 0.0) + v["ms"] / reps
    return wall, dev, acc, st
w1, d1, k1, st1 = run(0, 1)
print(f"world 1: wall {w1:.2f} device {d1:.2f} launches {st1['kernel_launches']} | " + " ".join(f"{k} {v:.2f}" for k, v in k1.items()))
for world in (2, 4, 8):
    w, d, k, st = run(0, world)
    print(f"world {world} rank 0: wall {w:.2f} device {d:.2f} (ideal {d1 / world:.2f}, efficiency {d1 / world / w:.3f}) launches {st['kernel_launches']} | " +
          " ".join(f"{n} {v:.2f} (+{v - k1.get(n, 0) / world:.2f})" for n, v in k.items()) + f" | sum of kernels {sum(k.values()):.2f}")
sc.close()
