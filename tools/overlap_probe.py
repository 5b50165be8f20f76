"""Does a rank's share of the frame finish sooner as K concurrent sub-shards (K scenes = K workspaces and streams on the one
GPU, K host threads), i.e. do the drain tails of one sub-shard's persistent kernels fill with the other's work?
Rank 0 of `world` is split into sub-ranks (K r + i, K world) with stripe_rows / K."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vermilion_amd as va
from vermilion_amd import scenes
pos, nrm, uv = scenes.sponza260k(); c = scenes.sponza_camera()
W, H, spp = 1920, 1080, 256
cam = va.make_camera(c["position"], c["rotation_deg"], W, H, spp, back_size=(3.6, 3.6 * H / W))
KMAX = 4
scs = [va.Scene(pos, nrm, uv) for _ in range(KMAX)]
outs = [torch.empty((H, W, 5), device="cuda") for _ in range(KMAX)]
FORM = int(os.environ.get("FORM", "0x100"), 0)
def timed(world, K, es=False, reps=5):
    stripe = 16 if world <= 4 else 4
    if world == 1: stripe = 16
    opts = [va.make_opts(seed=1, early_stop=es, rank=K * 0 + i, world=K * world, stripe_rows=max(1, stripe // K), pipeline=FORM) for i in range(K)]
    def job(i):
        scs[i].render_device(cam, opts[i], outs[i].data_ptr())
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
    for world in (1, 2, 4, 8):
        base = timed(world, 1, es)
        print(f"early_stop {int(es)} world {world} rank 0: one call {base:.2f} ms | " + " | ".join(f"{K} concurrent sub-shards {timed(world, K, es):.2f}" for K in (2, 3, 4) if (16 if world <= 4 else 4) % K == 0 or K == 2), flush=True)
for s in scs: s.close()
