"""overlap_probe.py with the two sub-shards on streams of different priority: does the high-priority half's drain fill with the
other half's blocks (and vice versa), instead of both pipelines running in lockstep?"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vermilion_amd as va
from vermilion_amd import scenes
pos, nrm, uv = scenes.sponza260k(); c = scenes.sponza_camera()
W, H, spp = 1920, 1080, 256
cam = va.make_camera(c["position"], c["rotation_deg"], W, H, spp, back_size=(3.6, 3.6 * H / W))
scs = [va.Scene(pos, nrm, uv) for _ in range(2)]
outs = [torch.empty((H, W, 5), device="cuda") for _ in range(2)]
lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
print("priority range", lo, hi)
FORM = int(os.environ.get("FORM", "0x100"), 0)
def timed(world, prios, es=False, reps=5, split=(1, 1)):
    stripe = 16 if world <= 4 else 4
    K = len(prios)
    streams = [torch.cuda.Stream(priority=p) for p in prios]
    opts = [va.make_opts(seed=1, early_stop=es, rank=i, world=K * world, stripe_rows=max(1, stripe // K), pipeline=FORM) for i in range(K)]
    def job(i):
        scs[i].render_device(cam, opts[i], outs[i].data_ptr(), streams[i].cuda_stream)
    best = 1e9
    for r in range(reps + 1):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        th = [threading.Thread(target=job, args=(i,)) for i in range(1, K)]
        for t in th: t.start()
        job(0)
        for t in th: t.join()
        torch.cuda.synchronize()
        if r: best = min(best, (time.perf_counter() - t0) * 1e3)
    return best
for es in (False, True):
    for world in (1, 4, 8):
        print(f"early_stop {int(es)} world {world} rank 0: one call {timed(world, [0], es):.2f} ms | two halves, equal priority {timed(world, [0, 0], es):.2f} | "
              f"high / low {timed(world, [-1, 0], es):.2f} | low / high {timed(world, [0, -1], es):.2f}", flush=True)
for s in scs: s.close()
