"""Differential fuzz of the production kernels against the CPU oracle (run on the GPU box; not part of the test suite):
random triangle soups (the generators of tests/test_gpu_parity.py), random cameras inside and outside the geometry,
64 ... 256 samples per pixel (camera-ray waves of one pixel: the assembly loop with its uniform pops and leaves), leaf
sizes 1 ... 16, all four tree builders (device-built trees: the oracle traverses the exported tree), both samplings, several
LDS stack depths, random sphere tables, and every case again under VMX_SAMPLING_ELIDE_DEAD.  Frames and sample counts must be
bit-identical.   python tools/fuzz_parity.py [cases] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import oracle_lib as O
import vermilion_amd as va
from test_gpu_parity import _random_soup, bits

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)
bad = 0
t0 = time.time()
for case in range(cases):
    kind = ["sheets", "duplicates", "slivers", "scales"][rng.integers(0, 4)]
    n = int(rng.choice([3, 40, 300, 2500, 12000]))
    leaf = int(rng.choice([1, 2, 4, 7, 16]))
    builder = int(rng.choice([0, 0, 1, 2, 3]))
    spp = int(rng.choice([64, 128, 256]))
    sampling = int(rng.choice([0, 0, 1, 0x100]))
    es = bool(rng.integers(0, 2))
    W, H = (48, 32) if spp >= 128 else (64, 40)
    pos, nrm, uv = _random_soup(rng, n, kind)
    cpos = rng.uniform(-900, 900, 3) if rng.random() < 0.5 else rng.uniform(-1900, 1900, 3) * np.array([1, 0.25, 1]) + np.array([0, 500, 0])
    rot = rng.uniform(-180, 180, 3) * np.array([0.3, 1.0, 0.1])
    cam = va.make_camera(tuple(float(v) for v in cpos), tuple(float(v) for v in rot), W, H, spp, back_size=(3.6, 3.6 * H / W))
    # a random sphere table in one case of three (weak and strong emitters, big and small, some around the camera)
    spheres = None
    if rng.random() < 0.34:
        tab = []
        for _ in range(int(rng.integers(1, 9))):
            ctr = cpos + rng.normal(0, 1, 3) * float(rng.choice([50, 400, 2000]))
            tab.append(dict(centre=tuple(float(v) for v in ctr), radius=float(rng.choice([20, 150, 900, 5000])),
                            colour=tuple(float(v) for v in rng.uniform(0, 1, 3) * float(rng.choice([0.3, 1.0, 3.0]))),
                            emit=bool(rng.random() < 0.6), normal_sign=float(rng.choice([1, -1]))))
        spheres = va.spheres_array(tab)
    kw = [{"pipeline": 4}, {"pipeline": 4, "tail_threshold": 1}, {"pipeline": 4, "lds_entries": int(rng.choice([1, 3, 6, 40]))}, {},
          {"pipeline": 4 | 0x200, "tail_threshold": 1}, {"pipeline": 4 | 0x100, "tail_threshold": 1}][rng.integers(0, 6)]
    try:
        # a random texture in one case of four: 1-4 channels, values below 0 and above 1, now and then an infinity or a
        # zero (throughput 0 or inf: the steps whose colour product would be NaN must be shaded in full)
        tex = None
        if rng.random() < 0.25:
            ch = int(rng.integers(1, 5))
            tex = rng.uniform(-0.5, 2.0, (int(rng.integers(1, 9)), int(rng.integers(1, 9)), ch)).astype(np.float32)
            tex[rng.random(tex.shape) < 0.05] = np.float32(rng.choice([0.0, np.inf, -np.inf]))
            if ch == 1: tex = tex[:, :, 0]
        with va.Scene(pos, nrm, uv, spheres=spheres, leaf_size=leaf, builder=builder) as g:
            osc = O.OracleScene(pos, nrm, uv, spheres=spheres, leaf_size=leaf, tree=g.bvh() if builder else None)
            if tex is not None:
                g.bind_texture(tex)
                osc.bind_texture(tex)
            seed = int(rng.integers(1, 1 << 30))
            opts = va.make_opts(seed=seed, early_stop=es, sampling=sampling, **kw)
            img, st = g.render(cam, opts)
            ref, rst = osc.render(cam, opts)
            same = np.array_equal(bits(img), bits(ref)) and st["samples"] == rst["samples"]
            why = "" if same else f" [default form: {int((bits(img) != bits(ref)).any(axis=2).sum())} pixels differ]"
            # VMX_SAMPLING_ELIDE_DEAD: the same frame from fewer rays
            img2, st2 = g.render(cam, va.make_opts(seed=seed, early_stop=es, sampling=sampling | va.VMX_SAMPLING_ELIDE_DEAD, **kw))
            if not np.array_equal(bits(img2), bits(ref)): why += f" [with elision: {int((bits(img2) != bits(ref)).any(axis=2).sum())} pixels differ]"
            same = same and np.array_equal(bits(img2), bits(ref)) and st2["samples"] == rst["samples"]
            same = same and st2["rays_primary"] + st2["rays_secondary"] <= st["rays_primary"] + st["rays_secondary"]
            osc.close()
    except va.VmxError as e:  # e.g. LBVH deeper than the reference's 64-entry stack on many coincident centroids
        print(f"case {case}: {kind} n={n} leaf={leaf} builder={builder}: {e}")
        continue
    bad += not same
    print(f"case {case:3d}: {kind:10s} n={n:5d} leaf={leaf:2d} builder={builder} spp={spp:3d} sampling={sampling:#x} es={int(es)} {kw} -> "
          f"{'ok' if same else 'MISMATCH'} ({st['rays_primary'] + st['rays_secondary']} rays, {st2['rays_primary'] + st2['rays_secondary']} with elision{', spheres' if spheres is not None else ''}{', textured' if tex is not None else ''}){why}", flush=True)
print(f"{cases} cases, {bad} mismatches, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
