"""Diagnostic run on the GPU box: product vs oracle on every level, with mismatch counts."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O
import vermilion_amd as va
from vermilion_amd import scenes

def bits(a): return np.ascontiguousarray(a).view(np.uint32)

def rand_rays(n, seed, lo=(-1500, 5, -900), hi=(1500, 950, 900)):
    r = np.random.RandomState(seed)
    o = r.uniform(lo, hi, size=(n, 3)).astype(np.float32)
    d = r.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    return o, d.astype(np.float32)

def check_scene(name, W, H, spp, nrays=200000):
    gen, camf = scenes.SCENES[name]
    pos, nrm, uv = gen()
    t0 = time.time(); sc = va.Scene(pos, nrm, uv); t_gpu_build = time.time() - t0
    osc = O.OracleScene(pos, nrm, uv)
    print(f"== {name}: {pos.shape[0]} tris  desc={sc.describe()} build {t_gpu_build:.2f}s")
    gb, ob = sc.bvh(), osc.bvh()
    for k in gb: print("  bvh", k, "equal" if np.array_equal(gb[k], ob[k]) else "DIFFERENT")
    o, d = rand_rays(nrays, 3)
    tri, t = sc.trace(o, d); rtri, rt = osc.trace(o, d)
    print("  trace: id mismatches", int((tri != rtri).sum()), " t-bit mismatches", int((bits(t) != bits(rt)).sum()), " hits", int((tri >= 0).sum()))
    h = sc.raycast(o, d); rh = osc.raycast(o, d)
    for f in h.dtype.names:
        if f == "pad": continue
        a, b = h[f], rh[f]
        neq = (bits(a) != bits(b)) if a.dtype == np.float32 else (a != b)
        if neq.ndim > 1: neq = neq.any(axis=1)
        print(f"  raycast.{f}: mismatches {int(neq.sum())}")
    c = camf()
    cam = va.make_camera(c["position"], c["rotation_deg"], W, H, spp, back_size=(3.6, 3.6 * H / W))
    for sampling in (0, 1):
        opts = va.make_opts(seed=5, sampling=sampling, collect_counters=True)
        po, pd = O.primary_rays(cam, opts, 0)
        rad, st = sc.radiance(po[:nrays], pd[:nrays], opts); rrad, rst = osc.radiance(po[:nrays], pd[:nrays], opts)
        print(f"  radiance sampling={sampling}: mismatching paths", int((bits(rad) != bits(rrad)).any(axis=1).sum()),
              " rays gpu/orc", st["rays_primary"] + st["rays_secondary"], rst["rays_primary"] + rst["rays_secondary"],
              " inner", st["primary"]["inner_visits"] + st["bounce"]["inner_visits"], rst["primary"]["inner_visits"])
    for es in (1, 0):
        for sampling in (0, 1):
            opts = va.make_opts(seed=9, early_stop=bool(es), sampling=sampling)
            t0 = time.time(); img, st = sc.render(cam, opts); tg = time.time() - t0
            t0 = time.time(); ref, rst = osc.render(cam, opts); tc = time.time() - t0
            neq = (bits(img) != bits(ref)).reshape(H, W, 5).any(axis=2)
            rays = st["rays_primary"] + st["rays_secondary"]
            print(f"  render es={es} sampling={sampling}: pixel mismatches {int(neq.sum())}/{W*H}  rays gpu {rays} orc {rst['rays_primary'] + rst['rays_secondary']}"
                  f"  gpu {st['ms_device']:.1f} ms ({rays / max(st['ms_device'], 1e-9) / 1e3:.1f} Mrays/s, wall {tg*1e3:.0f} ms, passes {st['passes']}, launches {st['kernel_launches']})  cpu {tc*1e3:.0f} ms")
            if neq.sum():
                ys, xs = np.nonzero(neq)
                for y, x in list(zip(ys, xs))[:5]:
                    print("     px", x, y, img[y, x], ref[y, x])
    tri, t = sc.primary_ids(cam, va.make_opts(seed=9), 0)
    po, pd = O.primary_rays(cam, va.make_opts(seed=9), 0)
    rtri, rt = osc.trace(po, pd)
    print("  primary_ids: id mismatches", int((tri != rtri).sum()), " t-bit mismatches", int((bits(t) != bits(rt)).sum()))
    # megakernel pipeline must give the same frame
    img1, st1 = sc.render(cam, va.make_opts(seed=9, early_stop=True))
    for pl in (1, 4):
        img2, st2 = sc.render(cam, va.make_opts(seed=9, early_stop=True, pipeline=pl))
        print(f"  pipeline={pl} vs 0 frame equal:", np.array_equal(bits(img1), bits(img2)), f" {st2['ms_device']:.1f} ms vs {st1['ms_device']:.1f} ms",
              "rays equal:", st1["rays_secondary"] == st2["rays_secondary"])
    sc.close()

if __name__ == "__main__":
    print("devices:", va._lib.lib().vmx_device_count())
    check_scene("cornell8", 256, 256, 16)
    check_scene("bunny70k", 256, 256, 16)
    check_scene("sponza260k", 480, 270, 16)
