"""Perf probe on the GPU box: sponza260k at a given resolution/spp, per-stage timings."""
import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vermilion_amd as va
from vermilion_amd import scenes

def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "sponza260k"
    W = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
    H = int(sys.argv[3]) if len(sys.argv) > 3 else 1080
    spp = int(sys.argv[4]) if len(sys.argv) > 4 else 64
    modes = sys.argv[5] if len(sys.argv) > 5 else "es0s0,es1s0"
    reps = int(sys.argv[6]) if len(sys.argv) > 6 else 2
    gen, camf = scenes.SCENES[name]
    pos, nrm, uv = gen()
    sc = va.Scene(pos, nrm, uv, builder=int(os.environ.get("BUILDER", "0")))  # 0 reference, 1 SAH, 2 GPU LBVH
    print(name, sc.describe())
    c = camf()
    cam = va.make_camera(c["position"], c["rotation_deg"], W, H, spp, back_size=(3.6, 3.6 * H / W))
    for mode in modes.split(","):
        es = int(mode[2]); sampling = int(mode[4]); extra = mode[5:]
        kw = {}
        if "m" in extra: kw["pipeline"] = 1
        if "o" in extra: kw["pipeline"] = 2
        if "O" in extra: kw["pipeline"] = 3
        if "S" in extra: kw["pipeline"] = 4
        if "c" in extra: kw["collect_counters"] = True
        if os.environ.get("REFILL"): kw["refill_min"] = int(os.environ["REFILL"])
        if os.environ.get("SHADE"): kw["shade_min"] = int(os.environ["SHADE"])
        if os.environ.get("TAIL"): kw["tail_threshold"] = int(os.environ["TAIL"])
        if os.environ.get("MAXP"): kw["max_paths"] = int(os.environ["MAXP"])
        if os.environ.get("REORDER"): kw["reorder"] = int(os.environ["REORDER"], 0)
        if os.environ.get("LDSE"): kw["lds_entries"] = int(os.environ["LDSE"])
        for r in range(reps):
            opts = va.make_opts(seed=1, early_stop=bool(es), sampling=sampling, **kw)
            t0 = time.time()
            import torch
            st = None
            out = torch.empty((H, W, 5), dtype=torch.float32, device="cuda")
            st = sc.render_device(cam, opts, out.data_ptr())
            dt = time.time() - t0
            rays = st["rays_primary"] + st["rays_secondary"]
            p, b = st["primary"], st["bounce"]
            sh = st["shade"]
            print(f"{mode} rep{r}: {rays/1e6:.1f} Mrays  dev {st['ms_device']:.1f} ms  wall {dt*1e3:.1f} ms -> {rays/st['ms_device']/1e3:.1f} Mrays/s | "
                  f"primary {p['rays']/1e6:.1f}M rays {p['ms']:.1f} ms x{p['launches']} ({p['rays']/max(p['ms'],1e-9)/1e3:.0f} Mr/s) | bounce {b['rays']/1e6:.1f}M rays {b['ms']:.1f} ms x{b['launches']} | "
                  f"shade {sh['ms']:.1f} ms x{sh['launches']} | passes {st['passes']} launches {st['kernel_launches']} samples {st['samples']/1e6:.1f}M disc {st['samples_discarded']}"
                  + (f" inner/ray {(p['inner_visits']+b['inner_visits'])/rays:.1f} tri/ray {(p['tri_tests']+b['tri_tests'])/rays:.1f}" if "c" in extra else ""))
    sc.close()

if __name__ == "__main__":
    main()
