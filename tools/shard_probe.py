"""Per-rank cost of the sharded bench frame on ONE GPU: rank r of `world` renders its stripes; strong-scaling
efficiency a multi-GPU run could reach at best = (full-frame time / world) / (slowest rank's time)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vermilion_amd as va
from vermilion_amd import scenes
pos, nrm, uv = scenes.sponza260k(); c = scenes.sponza_camera()
W, H, spp = 1920, 1080, 256
cam = va.make_camera(c["position"], c["rotation_deg"], W, H, spp, back_size=(3.6, 3.6 * H / W))
sc = va.Scene(pos, nrm, uv)
out = torch.empty((H, W, 5), device="cuda")
STRIPE = int(os.environ.get("STRIPE", "16"))
def run(rank, world, **kw):
    o = va.make_opts(seed=1, early_stop=False, rank=rank, world=world, stripe_rows=STRIPE, **kw)
    sc.render_device(cam, o, out.data_ptr())
    t0 = time.perf_counter(); st = sc.render_device(cam, o, out.data_ptr()); torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3, st
full, st = run(0, 1)
print(f"world 1: {full:.1f} ms")
for world in (2, 4, 8):
    ts = [run(r, world)[0] for r in range(world)]
    st = run(0, world)[1]
    print(f"world {world}: ranks {min(ts):.1f}..{max(ts):.1f} ms; ideal {full / world:.1f}; best-case efficiency {full / world / max(ts):.2f} | "
          f"rank 0: primary {st['primary']['ms']:.1f} bounce {st['bounce']['ms']:.1f} x{st['bounce']['launches']} shade {st['shade']['ms']:.1f} launches {st['kernel_launches']}")
if len(sys.argv) > 1:
    for world in (4, 8):
        for tail in (16 << 20, 8 << 20, 4 << 20, 2 << 20, 1 << 20, 512 << 10):
            ms, st = run(0, world, tail_threshold=tail)
            print(f"world {world} rank 0 tail_threshold {tail >> 10}K: {ms:.1f} ms | primary {st['primary']['ms']:.1f} bounce {st['bounce']['ms']:.1f} x{st['bounce']['launches']} shade {st['shade']['ms']:.1f} launches {st['kernel_launches']}")
    for tail in (16 << 20, 2 << 20):
        ms, st = run(0, 8, tail_threshold=tail)
        print("world 8 rank 0 tail", tail >> 10, "K:", {k: (round(v["ms"], 2), v["launches"]) for k, v in sc.timings().items() if v["launches"]})
