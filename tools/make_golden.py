"""Generates tests/golden/*.npz from the CPU oracle (parity build, this container).

The reference ships no golden vectors and cannot be built here (DESIGN.md
"Oracle"), so these fixtures pin the *oracle's* outputs: they guard it against
regressions / cross-machine drift and give the GPU tests a second, committed
checker.  Inputs are stored next to the expected outputs; geometry uses only
small integers (exact in float32), no transcendental functions.

    python tools/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
import vermilion_amd as va  # noqa: E402
from vermilion_amd import scenes  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def special_rays():
    """axis-aligned / zero-component / grazing / NaN rays: the slab-NaN and tie cases"""
    o, d = [], []
    # straight down and straight up over the lattice, origins exactly on cell corners
    for x in (-900.0, -720.0, 0.0, 35.0, 150.0):
        for z in (-700.0, -560.0, 10.0):
            o.append((x, 800.0, z)), d.append((0.0, -1.0, 0.0))
            o.append((x, 5.0, z)), d.append((0.0, 1.0, 0.0))
    # along +x / -z with zero components, origin on box planes of the Cornell block
    for y in (1.0, 200.0, 400.0):
        o.append((-1000.0, y, 0.0)), d.append((1.0, 0.0, 0.0))
        o.append((150.0, y, 900.0)), d.append((0.0, 0.0, -1.0))
        o.append((-250.0, y, 900.0)), d.append((0.0, 0.0, -1.0))
    # exactly along the shared diagonal of a quad (tie between its two triangles)
    o.append((0.0, 500.0, 1000.0)), d.append((0.0, 0.0, -1.0))
    o.append((-600.0, 1.0, 600.0)), d.append((0.70710677, 0.0, -0.70710677))
    # NaN / zero / inf directions and origins
    o.append((0.0, 100.0, 0.0)), d.append((np.nan, np.nan, np.nan))
    o.append((0.0, 100.0, 0.0)), d.append((0.0, 0.0, 0.0))
    o.append((0.0, 100.0, 0.0)), d.append((np.inf, 0.0, 0.0))
    o.append((np.nan, 100.0, 0.0)), d.append((0.0, -1.0, 0.0))
    # the light-leak ray of SURVEY A-2 and the ceiling ray of §8a-6
    v = np.array([0.0, 3300.0 - 300.0, 1300.0 - 5000.0])
    o.append((0.0, 300.0, 5000.0)), d.append(tuple(v / np.linalg.norm(v)))
    o.append((0.0, 500.0, 1800.0)), d.append((0.0, 1.0, 0.0))
    return np.asarray(o, np.float32), np.asarray(d, np.float32)


def random_rays(n, seed):
    r = np.random.RandomState(seed)
    o = r.uniform((-1200, 5, -900), (1200, 950, 1500), size=(n, 3)).astype(np.float32)
    d = r.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return o, d.astype(np.float32)


def make(name, gen, camf, W, H, spp):
    pos, nrm, uv = gen()
    sc = O.OracleScene(pos, nrm, uv)
    so, sd = special_rays()
    ro, rd = random_rays(3000, 11)
    o = np.concatenate([so, ro])
    d = np.concatenate([sd, rd])
    out = {"pos": pos, "nrm": nrm, "uv": uv, "ray_o": o, "ray_d": d}
    tri, t = sc.trace(o, d)
    out["trace_id"], out["trace_t"] = tri, t
    out["raycast"] = sc.raycast(o, d).view(np.uint32).reshape(-1, 16)
    b = sc.bvh()
    for k, v in b.items():
        out["bvh_" + k] = v
    c = camf()
    cam = va.make_camera(c["position"], c["rotation_deg"], W, H, spp)
    out["cam"] = np.array(list(c["position"]) + list(c["rotation_deg"]) + [W, H, spp], np.float64)
    for sampling in (0, 1):
        opts = va.make_opts(seed=3, sampling=sampling)
        po, pd = O.primary_rays(cam, opts, 0)
        if sampling == 0:
            out["primary_o"], out["primary_d"] = po, pd
            pt, ptt = sc.trace(po, pd)
            out["primary_id"], out["primary_t"] = pt, ptt
        rad, _ = sc.radiance(po, pd, opts)
        out[f"radiance_s{sampling}"] = rad
        for es in (0, 1):
            img, st = sc.render(cam, va.make_opts(seed=3, early_stop=bool(es), sampling=sampling))
            out[f"render_es{es}_s{sampling}"] = img
            out[f"rays_es{es}_s{sampling}"] = np.array([st["rays_primary"], st["rays_secondary"], st["samples"]],
                                                      np.uint64)
    out["stream"] = np.stack([O.stream(s, p, k, 8) for (s, p, k) in ((0, 0, 0), (1, 2, 3), (2**63 + 5, 2**31, 255))])
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    make("cornell8", scenes.cornell8, scenes.cornell_camera, 48, 32, 16)
    make("lattice", scenes.lattice, scenes.lattice_camera, 40, 24, 32)
