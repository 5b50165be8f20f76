"""Second differential fuzz (run on the GPU box; not part of the test suite): the parts of the ABI tools/fuzz_parity.py
does not vary.  Per case, on a random triangle soup with a random sphere table and 0-2 random textures:
  frames      odd image sizes (1 ... 97 x 1 ... 61), spp that are not multiples of 4, random film, random paths-per-pass
              and samples-per-batch (many passes), a random rank of a random world with random stripes against the rows
              of the oracle's whole frame, every pipeline form, with and without VMX_SAMPLING_ELIDE_DEAD
  radiance    explicit rays incl. axis-parallel, zero, infinite and NaN directions and origins
  raycast     the same rays through vmx_raycast, vmx_trace; vmx_primary_ids of a random sample index
  multi       the frame through vmx_multi_* with 2-3 replicas (one case in three)
  bruteforce  BruteForceTracer frames with both readings of abs()
Everything must be bit-identical to the oracle.   python tools/fuzz_wide.py [cases] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import oracle_lib as O
import vermilion_amd as va
from test_gpu_parity import _random_soup, bits, same_f32

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
bad, t0 = 0, time.time()


def rows_of(H, stripe, rank, world):
    return [r for r in range(H) if (r // stripe) % world == rank]


def special_rays(n):
    o = rng.uniform(-1500, 1500, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d = d.astype(np.float32)
    k = n // 8
    d[:k, rng.integers(0, 3)] = 0.0                                  # axis-parallel
    d[k:2 * k] = np.eye(3, dtype=np.float32)[rng.integers(0, 3, k)] * rng.choice([-1, 1], (k, 1)).astype(np.float32)
    d[2 * k] = 0.0                                                   # zero direction
    d[2 * k + 1, 0] = np.nan
    d[2 * k + 2] = np.nan
    d[2 * k + 3, 1] = np.inf
    o[2 * k + 4, 2] = np.inf
    o[2 * k + 5, 0] = np.nan
    d[2 * k + 6] *= np.float32(1e-30)                                # not unit length
    d[2 * k + 7] *= np.float32(1e20)
    return o, d


for case in range(cases):
    kind = ["sheets", "duplicates", "slivers", "scales"][rng.integers(0, 4)]
    n = int(rng.choice([1, 3, 40, 300, 2500]))
    leaf = int(rng.choice([1, 2, 4, 7, 16]))
    pos, nrm, uv = _random_soup(rng, n, kind)
    cpos = rng.uniform(-900, 900, 3)
    tab = []
    for _ in range(int(rng.integers(0, 9))):
        ctr = cpos + rng.normal(0, 1, 3) * float(rng.choice([50, 400, 2000]))
        tab.append(dict(centre=tuple(float(v) for v in ctr), radius=float(rng.choice([20, 150, 900, 5000])),
                        colour=tuple(float(v) for v in rng.uniform(0, 1, 3) * float(rng.choice([0.3, 1.0, 3.0]))),
                        emit=bool(rng.random() < 0.5), normal_sign=float(rng.choice([1, -1]))))
    spheres = va.spheres_array(tab) if tab and rng.random() < 0.7 else None
    texs = []
    for _ in range(int(rng.choice([0, 0, 1, 2]))):
        ch = int(rng.integers(1, 5))
        t = rng.uniform(-0.5, 2.0, (int(rng.integers(1, 9)), int(rng.integers(1, 9)), ch)).astype(np.float32)
        t[rng.random(t.shape) < 0.03] = np.float32(rng.choice([0.0, np.inf]))
        texs.append(t[:, :, 0] if ch == 1 else t)
    W, H = int(rng.integers(1, 98)), int(rng.integers(1, 62))
    spp = int(rng.choice([4, 5, 7, 12, 16, 20, 36, 64, 100, 128, 256]))
    if W * H * spp > 600_000:
        spp = 16
    rot = rng.uniform(-180, 180, 3) * np.array([0.3, 1.0, 0.1])
    cam = va.make_camera(tuple(float(v) for v in cpos), tuple(float(v) for v in rot), W, H, spp,
                         back_size=(float(rng.uniform(0.5, 8)), float(rng.uniform(0.5, 8))), back_distance=float(rng.uniform(0.5, 20)))
    world = int(rng.choice([1, 1, 2, 3, 5])); rank = int(rng.integers(0, world)); stripe = int(rng.choice([1, 3, 16]))
    seed = int(rng.integers(1, 1 << 30))
    sampling = int(rng.choice([0, 0, 1, 0x100]))
    es = bool(rng.integers(0, 2))
    msgs = []
    try:
        with va.Scene(pos, nrm, uv, spheres=spheres, leaf_size=leaf) as g:
            osc = O.OracleScene(pos, nrm, uv, spheres=spheres, leaf_size=leaf)
            for t in texs:
                g.bind_texture(t); osc.bind_texture(t)
            ref, rst = osc.render(cam, va.make_opts(seed=seed, early_stop=es, sampling=sampling))
            rows = rows_of(H, stripe, rank, world)
            for kw in ({}, {"pipeline": 1}, {"pipeline": 4}, {"pipeline": 4 | 0x100, "tail_threshold": 1}, {"pipeline": 4 | 0x200, "tail_threshold": 1},
                       {"pipeline": 4, "max_paths": int(rng.integers(1000, 200000)), "samples_per_batch": int(rng.integers(0, spp + 1))}):
                for flag in (0, va.VMX_SAMPLING_ELIDE_DEAD):
                    o = va.make_opts(seed=seed, early_stop=es, sampling=sampling | flag, rank=rank, world=world, stripe_rows=stripe, **kw)
                    img, st = g.render(cam, o)
                    if not np.array_equal(bits(img), bits(ref[rows])):
                        msgs.append(f"frame {kw} flag {flag:#x}: {int((bits(img) != bits(ref[rows])).any(axis=2).sum())} pixels")
            # explicit rays
            ro, rd = special_rays(4096)
            ropts = va.make_opts(seed=seed, sampling=sampling)
            for kw in ({}, {"pipeline": 4, "tail_threshold": 1}):
                rad, _ = g.radiance(ro, rd, va.make_opts(seed=seed, sampling=sampling, **kw))
                rrad, _ = osc.radiance(ro, rd, ropts)
                if not np.all(same_f32(rad, rrad)):  # (any NaN equals any NaN: x86 and gfx950 give inf * 0 different sign bits)
                    badp = np.argwhere((~same_f32(rad, rrad)).any(axis=1))[:, 0]
                    i0 = int(badp[0])
                    hit = osc.raycast(ro[i0:i0 + 1], rd[i0:i0 + 1])
                    msgs.append(f"radiance {kw}: {len(badp)} paths, e.g. {badp[:6].tolist()}: o={ro[i0].tolist()} d={rd[i0].tolist()} gpu={rad[i0].tolist()} "
                                f"oracle={rrad[i0].tolist()} first hit uv={hit['uv'][0].tolist()} tri={int(hit['tri_id'][0])}")
            tri, tt = g.trace(ro, rd)
            rtri, rtt = osc.trace(ro, rd)
            if not (np.array_equal(tri, rtri) and np.all(same_f32(tt, rtt))):
                msgs.append(f"trace: {int((tri != rtri).sum())} ids, {int((~same_f32(tt, rtt)).sum())} distances")
            kk = int(rng.integers(0, 4 * (spp // 4)))
            ptri, pt = g.primary_ids(cam, va.make_opts(seed=seed), kk)
            po, pd = O.primary_rays(cam, va.make_opts(seed=seed), kk)
            qtri, qt = osc.trace(po, pd)
            if not (np.array_equal(ptri, qtri) and np.all(same_f32(pt, qt))):
                msgs.append("primary_ids")
            a, b = g.raycast(ro, rd), osc.raycast(ro, rd)
            for f in a.dtype.names:
                if f == "pad":
                    continue
                same = same_f32(a[f], b[f]) if a[f].dtype == np.float32 else (a[f] == b[f])
                if not np.all(same):
                    msgs.append(f"raycast.{f}: {int((~same).sum())} (first at ray {int(np.argwhere(~same)[0][0])})")
            # the one-process multi-device path (replicas on this GPU), whole frame
            if rng.random() < 0.3:
                mw = int(rng.choice([2, 3]))
                with va.MultiScene(pos, nrm, uv, devices=[0] * mw, spheres=spheres, leaf_size=leaf) as m:
                    for t in texs:
                        m.bind_texture(t)
                    for flag in (0, va.VMX_SAMPLING_ELIDE_DEAD):
                        mi, _ = m.render(cam, va.make_opts(seed=seed, early_stop=es, sampling=sampling | flag, stripe_rows=stripe))
                        if not np.array_equal(bits(mi), bits(ref)):
                            msgs.append(f"multi x{mw} flag {flag:#x}: {int((bits(mi) != bits(ref)).any(axis=2).sum())} pixels")
            # BruteForceTracer
            bcam = va.make_camera(tuple(float(v) for v in cpos), tuple(float(v) for v in rot), min(W, 48), min(H, 32), min(spp, 64))
            for flags in (0, va._lib.VMX_BF_ABS_INT):
                bi, _ = g.render_bruteforce(bcam, va.make_opts(seed=seed), flags)
                br, _ = osc.render_bruteforce(bcam, va.make_opts(seed=seed), flags)
                if not np.array_equal(bits(bi), bits(br)):
                    msgs.append(f"bruteforce flags {flags}: {int((bits(bi) != bits(br)).any(axis=2).sum())} pixels")
            osc.close()
    except va.VmxError as e:
        print(f"case {case}: {e}")
        continue
    bad += bool(msgs)
    print(f"case {case:3d}: {kind:10s} n={n:5d} leaf={leaf:2d} {W}x{H} spp={spp} sampling={sampling:#x} es={int(es)} rank {rank}/{world} stripe {stripe} "
          f"spheres={len(tab) if spheres is not None else 'ref'} tex={len(texs)} -> {'ok' if not msgs else 'MISMATCH ' + '; '.join(msgs)}", flush=True)
print(f"{cases} cases, {bad} mismatches, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
