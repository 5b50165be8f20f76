"""The CPU oracle against (a) the reference-run observations recorded in
SURVEY.md, (b) hand-derived known answers, (c) its own committed golden
fixtures.  The reference has no tests or golden vectors of its own
and cannot be built here, so parity with the reference proper stays unpinned
beyond (a)."""
import os

import numpy as np
import pytest

import oracle_lib as O
import vermilion_amd as va
from vermilion_amd import scenes

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.fixture(scope="module")
def cornell():
    return O.OracleScene(*scenes.cornell8())


def one(sc, o, d):
    return sc.raycast(np.asarray([o], np.float32), np.asarray([d], np.float32))[0]


# ---- (a) observations of the real reference recorded by the survey ------------
def test_wall_sphere_cancellation_matches_survey_probe(cornell):
    """a ray that should hit the ceiling sphere at ~499.9 returned 499.287"""
    h = one(cornell, (0, 500, 1800), (0, 1, 0))
    assert abs(float(h["distance"]) - 499.287) < 5e-4
    assert h["tri_id"] == -1 and (h["flags"] & 1) and not (h["flags"] & 2)


def test_light_leaks_through_nearer_wall_sphere(cornell):
    """SURVEY A-2: light 2 (y=3300) colours a hit whose nearest surface is the ceiling"""
    v = np.array([0.0, 3000.0, -3700.0])
    h = one(cornell, (0, 300, 5000), v / np.linalg.norm(v))
    assert np.allclose(h["colour"], 15.2) and float(h["location"][1]) < 1010.0


def test_dark_pixels_take_seven_samples_at_16spp(cornell):
    """SURVEY A-9 / App. B: 16 spp -> dark pixels take 7 samples, mean 7.02 on an 8-tri scene"""
    c = scenes.cornell_camera()
    cam = va.make_camera(c["position"], c["rotation_deg"], 64, 64, 16)
    img, st = cornell.render(cam, va.make_opts(seed=2))
    depth = img[:, :, 4]
    dark = img[:, :, :3].sum(-1) == 0
    assert dark.mean() > 0.9 and np.all(depth[dark] == 7.0)
    assert 7.0 <= depth.mean() < 7.1
    assert set(np.unique(depth)).issubset({7.0, 10.0, 13.0, 16.0})  # one stratum of 4 at a time
    assert np.all(img[:, :, 3] == 1.0)


def test_nan_direction_misses_everything(cornell):
    """SURVEY A-1: a NaN direction misses the BVH and all spheres"""
    h = one(cornell, (0, 100, 0), (np.nan, np.nan, np.nan))
    assert not (h["flags"] & 1) and h["tri_id"] == -1 and np.isinf(h["distance"])


def test_parity_sampling_rarely_returns_light(cornell):
    """SURVEY A-1: with r2 = 10U only ~1 in 10 diffuse bounces continues; most paths are black"""
    n = 20000
    o = np.tile(np.float32([0, 420, 1900]), (n, 1))
    d = np.tile(np.float32([0, -0.2, -1]) / np.linalg.norm([0, -0.2, -1]), (n, 1)).astype(np.float32)
    rad, st = cornell.radiance(o, d, va.make_opts(seed=4))
    frac = (rad[:, :3].sum(1) > 0).mean()
    assert frac < 0.01
    # continuing bounces ~ 0.04 + 0.96*0.1 of the material hits
    assert 0.10 < st["rays_secondary"] / st["rays_primary"] < 0.20


# ---- (b) hand-derived known answers ---------------------------------------------
def test_moller_trumbore_known_answers():
    pos = np.float32([[0, 0, 0, 4, 0, 0, 0, 4, 0]])
    nrm = np.float32([[0, 0, 1] * 3])
    sc = O.OracleScene(pos, nrm, spheres=(va._lib.Sphere * 1)())  # no usable sphere: radius 0
    tri, t = sc.trace([[1, 1, 5], [1, 1, -5], [3, 3, 5], [1, 1, 5], [0, 0, 5]],
                      [[0, 0, -1], [0, 0, 1], [0, 0, -1], [0, 0, 1], [0, 0, -1]])
    assert tri.tolist() == [0, 0, -1, -1, 0]          # two-sided (A-11); u+v>1 misses; t>0 only; vertex hit
    assert t[0] == 5.0 and t[1] == 5.0 and t[4] == 5.0
    assert t[2] == np.float32(999999999.0)             # bvh.cpp:48
    h = sc.raycast([[1, 1, 5]], [[0, 0, -1]])[0]
    assert np.allclose(h["normal"], [0, 0, -1])        # interpolated normal is negated (A-12)
    assert (h["flags"] & 3) == 3


def test_first_tested_triangle_wins_ties():
    """strict < at bvh.cpp:90: two coincident triangles -> the earlier leaf slot wins"""
    pos = np.float32([[0, 0, 0, 4, 0, 0, 0, 4, 0], [0, 0, 0, 4, 0, 0, 0, 4, 0]])
    nrm = np.float32([[0, 0, 1] * 3] * 2)
    sc = O.OracleScene(pos, nrm)
    tri, _ = sc.trace([[1, 1, 5]], [[0, 0, -1]])
    order = sc.bvh()["prim_order"]
    assert tri[0] == order[0]


def test_bvh_topology_small():
    pos, nrm, uv = scenes.cornell8()
    sc = O.OracleScene(pos, nrm, uv)
    b = sc.bvh()
    assert sc.describe() == {"n_nodes": 5, "n_leaves": 3, "max_depth": 2}
    assert b["right_offset"][0] != 0 and sorted(b["prim_order"].tolist()) == list(range(8))
    # leaves partition the primitive range; every leaf has <= 4 prims (bvh.h:29)
    leaves = b["right_offset"] == 0
    assert b["nprims"][leaves].sum() == 8 and b["nprims"][leaves].max() <= 4
    # 4 triangles or fewer: the root itself is a leaf
    sc1 = O.OracleScene(pos[:3], nrm[:3])
    assert sc1.describe()["n_nodes"] == 1
    # node boxes bound their triangles
    for i in np.nonzero(leaves)[0]:
        ids = b["prim_order"][b["start"][i]:b["start"][i] + b["nprims"][i]]
        p = pos[ids].reshape(-1, 3)
        assert np.all(p.min(0) >= b["bbox"][i, :3]) and np.all(p.max(0) <= b["bbox"][i, 3:])


def test_split_axis_quirk_compares_z_with_y_only():
    """bbox.cpp:41-46: extent (10, 1, 5) picks z (5 > 1), not x"""
    tris = []
    for i in range(8):
        x, z = 10.0 * i / 7, 5.0 * (i % 2)
        tris.append([x, 0, z, x + .1, 0, z, x, .1, z])
    pos = np.float32(tris)
    nrm = np.float32([[0, 1, 0] * 3] * 8)
    b = O.OracleScene(pos, nrm).bvh()
    left = b["prim_order"][b["start"][1]:b["start"][1] + b["nprims"][1]]
    assert sorted(left.tolist()) == [0, 2, 4, 6]       # split on z, although x is the widest axis


def test_camera_matrix_and_forward_axis():
    cam = va.make_camera((0, 0, 0), (0, 0, 0), 8, 8, 4)
    assert np.array_equal(O.camera_matrix(cam), np.eye(3, dtype=np.float32))
    o, d = O.primary_rays(va.make_camera((1, 2, 3), (0, 0, 0), 33, 17, 4), va.make_opts(seed=1), 0)
    assert np.all(o == np.float32([1, 2, 3])) and np.all(d[:, 2] < -0.9)
    assert np.allclose(np.linalg.norm(d, axis=1), 1, atol=1e-6)
    # rotation (rx,ry,rz) degrees -> Ry(-ry) Rx(-rx) Rz(rz) (camera.cpp:43-47, pathtracer.cpp:219-221)
    rx, ry, rz = 10.0, 35.0, -20.0
    a, b, c = np.radians([-rx, -ry, rz])
    Rx = np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])
    Ry = np.array([[np.cos(b), 0, np.sin(b)], [0, 1, 0], [-np.sin(b), 0, np.cos(b)]])
    Rz = np.array([[np.cos(c), -np.sin(c), 0], [np.sin(c), np.cos(c), 0], [0, 0, 1]])
    M = O.camera_matrix(va.make_camera((0, 0, 0), (rx, ry, rz), 8, 8, 4)).T  # [row][col]
    assert np.allclose(M, Ry @ Rx @ Rz, atol=2e-6)


def test_camera_rotation_in_radians_is_taken_as_it_stands():
    """VMX_ROTATION_RADIANS: the adapter holds Camera::mRotation (radians, x and y negated, camera.cpp:43-47) and hands it
    over unchanged; pathtracer.cpp:219-221 rotates by exactly those three floats"""
    rng = np.random.default_rng(5)
    for _ in range(200):
        deg = np.float32(rng.uniform(-180, 180, 3)).astype(np.float64)  # cameraSettings.rotation is float (types.h)
        # what the Camera ctor computes from the settings' degrees, in double, narrowed to float
        rad = np.float32([-deg[0] * 3.1415926535 / 180, -deg[1] * 3.1415926535 / 180, deg[2] * 3.1415926535 / 180])
        a = O.camera_matrix(va.make_camera((0, 0, 0), np.float32(deg), 8, 8, 4))
        b = O.camera_matrix(va.make_camera((0, 0, 0), None, 8, 8, 4, rotation_rad=rad))
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    # arbitrary radians (a host that wrote the public field): the matrix of exactly those angles, no degree round trip
    for _ in range(200):
        r = np.float32(rng.uniform(-7, 7, 3))
        a, b, c = (float(x) for x in r)
        Rx = np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])
        Ry = np.array([[np.cos(b), 0, np.sin(b)], [0, 1, 0], [-np.sin(b), 0, np.cos(b)]])
        Rz = np.array([[np.cos(c), -np.sin(c), 0], [np.sin(c), np.cos(c), 0], [0, 0, 1]])
        M = O.camera_matrix(va.make_camera((0, 0, 0), None, 8, 8, 4, rotation_rad=r)).T
        assert np.allclose(M, Ry @ Rx @ Rz, atol=3e-6)
    # a one-ulp change of an angle is a different camera (what the degree round trip used to do to 9 % of them)
    r = np.float32([0.3, 1.1, -0.7])
    r2 = r.copy()
    r2[1] = np.nextafter(r2[1], np.float32(4))
    m1 = O.camera_matrix(va.make_camera((0, 0, 0), None, 8, 8, 4, rotation_rad=r))
    m2 = O.camera_matrix(va.make_camera((0, 0, 0), None, 8, 8, 4, rotation_rad=r2))
    assert not np.array_equal(m1.view(np.uint32), m2.view(np.uint32))


def test_pixel_footprint_and_film_geometry():
    """jitter footprint [-0.75, 0.25) around the integer pixel coordinate"""
    W, H = 16, 8
    cam = va.make_camera((0, 0, 0), (0, 0, 0), W, H, 64, back_distance=6.0, back_size=(3.6, 2.4))
    for k in (0, 17, 40, 63):
        _, d = O.primary_rays(cam, va.make_opts(seed=k), k)
        film = d / -d[:, 2:3] * 6.0
        px = (film[:, 0] / 3.6 + 0.5) * W
        py = (-film[:, 1] / 2.4 + 0.5) * H
        ex, ey = px - np.arange(W * H) % W, py - np.arange(W * H) // W
        s = k // 16
        lox, loy = (s >> 1) * 0.5 - 0.75, (s & 1) * 0.5 - 0.75
        assert np.all(ex >= lox - 1e-4) and np.all(ex < lox + 0.5 + 1e-4)
        assert np.all(ey >= loy - 1e-4) and np.all(ey < loy + 0.5 + 1e-4)


def test_rng_stream():
    import ctypes as C
    # splitmix64 known-answer (Vigna's reference implementation, seed 1234567)
    st = C.c_uint64(1234567)
    got = [O.lib().orc_splitmix64(C.byref(st)) for _ in range(3)]
    assert got == [6457827717110365317, 3203168211198807973, 9817491932198370423]
    # known answers of the keyed stream (seed, pixel, sample), from an independent restatement in Python integers of the
    # definition in oracle/vmx_oracle.cpp: Xoshiro::init / vmx_kernels.hip: rng_pixel_key + rng_init_keyed (round 4: two
    # mix64 per pixel, three 64-bit multiplies per sample) followed by xoshiro256**
    M = (1 << 64) - 1

    def mix64(z):
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
        return z ^ (z >> 31)

    def rotl(x, k):
        return ((x << k) | (x >> (64 - k))) & M

    def keyed(seed, pixel, k, n):
        a = mix64(seed ^ ((pixel << 32) & M))
        b = mix64((a + 0x9E3779B97F4A7C15) & M)
        s0 = mix64((a + k) & M)
        t = ((s0 ^ b) * 0xD6E8FEB86659FD93) & M
        s1 = t ^ (t >> 32)
        s = [s0, s1, rotl(s0, 24) ^ b, rotl(s1, 37) ^ a]
        out = []
        for _ in range(n):
            out.append((rotl((s[1] * 5) & M, 7) * 9) & M)
            t = (s[1] << 17) & M
            s[2] ^= s[0]
            s[3] ^= s[1]
            s[1] ^= s[2]
            s[0] ^= s[3]
            s[2] ^= t
            s[3] = rotl(s[3], 45)
        return out
    assert O.stream(1, 2, 3, 2).tolist() == [10614261844916165048, 15287749866284423084]
    assert O.stream(0, 0, 0, 2).tolist() == [675374820455444685, 3476903706756334224]
    for key in ((1, 2, 3), (12345678901234567, 2073599, 255), (7, 1 << 31, 4000000000), (M, 0xFFFFFFFF, 0xFFFFFFFF)):
        assert O.stream(*key, 8).tolist() == keyed(*key, 8), key
    a = O.stream(1, 2, 3, 64)
    assert np.array_equal(a, O.stream(1, 2, 3, 64)) and len(set(a.tolist())) == 64
    assert not np.array_equal(a, O.stream(1, 2, 4, 64)) and not np.array_equal(a, O.stream(1, 3, 3, 64))
    big = np.concatenate([O.stream(9, p, 0, 256) for p in range(64)])
    u = (big >> np.uint64(11)).astype(np.float64) * 2.0**-53
    assert abs(u.mean() - 0.5) < 0.01 and abs(u.var() - 1 / 12) < 0.005
    # neighbouring streams — consecutive samples of a pixel, the same sample of consecutive pixels, consecutive seeds —
    # are unrelated: every bit of the first outputs flips with probability 1/2 from one stream to the next, and the
    # first draws of 2^16 streams are uniform and serially uncorrelated (the jitters of a frame ARE such first draws)
    for streams in ([O.stream(5, 77, k, 4) for k in range(4096)], [O.stream(5, p, 11, 4) for p in range(4096)],
                    [O.stream(s_, 3, 2, 4) for s_ in range(4096)]):
        x = np.stack(streams)                                   # [stream, draw]
        flips = x[1:] ^ x[:-1]
        for d in range(4):
            f = np.unpackbits(flips[:, d].copy().view(np.uint8).reshape(-1, 8), axis=1).mean()
            assert abs(f - 0.5) < 0.004, (d, f)
        per_bit = np.unpackbits(flips[:, 0].copy().view(np.uint8).reshape(-1, 8), axis=1).mean(axis=0)
        assert np.all(np.abs(per_bit - 0.5) < 0.04)            # 4095 trials per bit: 5 sigma = 0.039
    first = np.concatenate([O.stream(2, p, k, 1) for p in range(256) for k in range(256)])
    u = (first >> np.uint64(11)).astype(np.float64) * 2.0**-53
    assert abs(u.mean() - 0.5) < 0.006 and abs(u.var() - 1 / 12) < 0.003
    assert abs(np.corrcoef(u[1:], u[:-1])[0, 1]) < 0.02 and abs(np.corrcoef(u[256:], u[:-256])[0, 1]) < 0.02
    h, _ = np.histogram(u, bins=64, range=(0, 1))
    assert ((h - 1024.0) ** 2 / 1024.0).sum() < 130             # chi-square, 63 degrees of freedom (p ~ 1e-6 at 130)


def test_early_stop_off_takes_every_sample_and_spp_rounds_down(cornell):
    c = scenes.cornell_camera()
    cam = va.make_camera(c["position"], c["rotation_deg"], 24, 16, 18)  # 18/4 = 4 -> 16 samples (A-17)
    img, st = cornell.render(cam, va.make_opts(seed=1, early_stop=False))
    assert np.all(img[:, :, 4] == 16.0) and st["samples"] == 24 * 16 * 16 == st["rays_primary"]


def test_reference_rng_and_keyed_rng_agree_statistically():
    """mt19937_64 (the reference's generator) vs the keyed xoshiro streams: same estimator"""
    pos, nrm, uv = scenes.cornell8()
    lights = va.spheres_array([
        dict(centre=(0, 700, 300), radius=200, colour=(1.5, 1.2, 0.9), emit=True),
        dict(centre=(0, -5e7, 0), radius=5e7), dict(centre=(0, 5e7 + 1000, 0), radius=5e7),
        dict(centre=(-5e7 + 2000, 0, 0), radius=5e7, normal_sign=-1), dict(centre=(5e7 - 2000, 0, 0), radius=5e7, normal_sign=-1),
        dict(centre=(0, 0, -5e7 + 2000), radius=5e7, normal_sign=-1), dict(centre=(0, 0, 5e7 - 2000), radius=5e7)])
    sc = O.OracleScene(pos, nrm, uv, spheres=lights)
    c = scenes.cornell_camera()
    cam = va.make_camera(c["position"], c["rotation_deg"], 24, 24, 256)
    opts = va.make_opts(seed=5, early_stop=False, sampling=va.VMX_SAMPLING_CORRECTED)
    a, _ = sc.render(cam, opts, rng_mode=O.ORC_RNG_XOSHIRO_KEYED)
    b, _ = sc.render(cam, opts, rng_mode=O.ORC_RNG_MT19937_64)
    ma, mb = a[:, :, :3].mean(), b[:, :, :3].mean()
    assert ma > 0.02 and abs(ma - mb) / ma < 0.05
    l2 = np.sqrt(((a[:, :, :3] - b[:, :, :3]) ** 2).sum(-1)).mean()
    assert l2 < 0.08  # per-pixel L2, Monte-Carlo noise at 256 spp


def test_elision_argument_holds_on_the_oracle():
    """The argument behind the kernels' two-phase shading and VMX_SAMPLING_ELIDE_DEAD (DESIGN_HISTORY.md 5.1), audited on the
    oracle's own Radiance, step by step: from a copy of the stream taken BEFORE a step, predict that the step is the
    path's last one for the material flag its hit turns out to have, and that no light sphere can colour it; then
    check that the path did end there (or went on with an all-NaN direction), that accumColour moved by exactly
    accumRadiance * (hitColour as the sphere table up to its last light decides it), and, where the step was predicted
    last for both flag values with no light in reach, that accumColour did not move at all.  No violations — on the
    reference's room, on tables with lights late and in view, with a texture (throughput != 1), in both samplings."""
    pos, nrm, uv = scenes.cornell8()
    c = scenes.cornell_camera()
    cam = va.make_camera(c["position"], c["rotation_deg"], 96, 64, 16)
    tables = [None,
              va.spheres_array([
                  dict(centre=(0, -5e7, 0), radius=5e7), dict(centre=(0, 5e7 + 1000, 0), radius=5e7),
                  dict(centre=(0, 700, 300), radius=200, colour=(1.5, 1.2, 0.9), emit=True),
                  dict(centre=(-5e7 + 2000, 0, 0), radius=5e7, normal_sign=-1),
                  dict(centre=(300, 500, 400), radius=60, colour=(0.4, 0.1, 0.1), emit=True),  # weak: the path goes on
                  dict(centre=(5e7 - 2000, 0, 0), radius=5e7, normal_sign=-1),
                  dict(centre=(0, 0, -5e7 + 2000), radius=5e7, normal_sign=-1), dict(centre=(0, 0, 5e7 - 2000), radius=5e7)])]
    tex = (np.random.RandomState(3).random_sample((8, 8, 3)) * 1.5).astype(np.float32)
    for table in tables:
        for textured in (False, True):
            sc = O.OracleScene(pos, nrm, uv, spheres=table)
            if textured:
                sc.bind_texture(tex)
            for sampling in (va.VMX_SAMPLING_PARITY, va.VMX_SAMPLING_CORRECTED):
                opts = va.make_opts(seed=12, sampling=sampling)
                o, d = O.primary_rays(cam, opts, 3)
                ref, _ = sc.radiance(o, d, opts)
                rad, au = sc.audit_elision(o, d, opts)
                assert np.array_equal(bits(rad), bits(ref))  # the audit's hooks do not touch the computation
                assert au["not_last"] == 0 and au["dead_changed"] == 0 and au["colour_mismatch"] == 0, (au, sampling, textured)
                assert au["steps"] >= o.shape[0]
                if sampling == va.VMX_SAMPLING_PARITY:  # r2 = 10 U: most steps end their path, most rays are not needed
                    assert au["predicted_last"] > 0.8 * au["steps"] and au["predicted_dead"] > 0.7 * au["steps"], au
                else:                                    # r2 = U: only Russian roulette, past depth 5
                    assert 0 < au["predicted_last"] < 0.2 * au["steps"], au
            sc.close()


# ---- (c) committed golden fixtures ---------------------------------------------------
@pytest.mark.parametrize("name", ["cornell8", "lattice"])
def test_oracle_reproduces_golden(name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    sc = O.OracleScene(g["pos"], g["nrm"], g["uv"])
    tri, t = sc.trace(g["ray_o"], g["ray_d"])
    assert np.array_equal(tri, g["trace_id"]) and np.array_equal(bits(t), bits(g["trace_t"]))
    assert np.array_equal(sc.raycast(g["ray_o"], g["ray_d"]).view(np.uint32).reshape(-1, 16), g["raycast"])
    b = sc.bvh()
    for k, v in b.items():
        assert np.array_equal(v, g["bvh_" + k]), k
    cp = g["cam"]
    cam = va.make_camera(cp[:3], cp[3:6], int(cp[6]), int(cp[7]), int(cp[8]))
    for sampling in (0, 1):
        rad, _ = sc.radiance(g["primary_o"], g["primary_d"], va.make_opts(seed=3, sampling=sampling))
        assert np.array_equal(bits(rad), bits(g[f"radiance_s{sampling}"]))
        for es in (0, 1):
            img, st = sc.render(cam, va.make_opts(seed=3, early_stop=bool(es), sampling=sampling))
            assert np.array_equal(bits(img), bits(g[f"render_es{es}_s{sampling}"]))
            assert [st["rays_primary"], st["rays_secondary"], st["samples"]] == g[f"rays_es{es}_s{sampling}"].tolist()
    po, pd = O.primary_rays(cam, va.make_opts(seed=3), 0)
    assert np.array_equal(bits(pd), bits(g["primary_d"]))
    for i, (s, p, k) in enumerate(((0, 0, 0), (1, 2, 3), (2**63 + 5, 2**31, 255))):
        assert np.array_equal(O.stream(s, p, k, 8), g["stream"][i])


def test_mt_radiance_is_deterministic_per_seed(cornell):
    g = np.load(os.path.join(GOLD, "cornell8.npz"))
    o, d = g["primary_o"][:512], g["primary_d"][:512]
    seeds = np.arange(512, dtype=np.uint64) + 100
    a = cornell.radiance_mt(o, d, seeds)
    assert np.array_equal(bits(a), bits(cornell.radiance_mt(o, d, seeds)))
    assert np.all(a[:, 3] > 0)  # w = primary hit distance (pathtracer.cpp:44-47)


# ---- §8 f-2: VermiTexture::Sample (meshEngine.cpp:21-46) -------------------------------------
def test_texture_sample_known_answers():
    """wrap by x - floor(x), nearest by round(x*(W-1)); the texel modulates the throughput of the
    diffuse-with-material branch only (pathtracer.cpp:153)"""
    pos = np.float32([[-500, 0, -500, 500, 0, -500, 500, 0, 500], [-500, 0, -500, 500, 0, 500, -500, 0, 500]])
    nrm = np.float32([[0, 1, 0] * 3] * 2)
    uv = np.float32([[0, 0, 2, 0, 2, 2], [0, 0, 2, 2, 0, 2]])  # uv spans [0,2]^2: wraps once
    lights = va.spheres_array([dict(centre=(0, 900, 0), radius=300, colour=(2, 2, 2), emit=True)])
    tex = np.zeros((2, 4, 3), np.float32)
    tex[0, :, 0] = [0.1, 0.2, 0.3, 0.4]
    tex[1, :, 0] = [0.5, 0.6, 0.7, 0.8]
    tex[:, :, 1] = 1.0
    tex[:, :, 2] = 0.25
    sc = O.OracleScene(pos, nrm, uv, spheres=lights)
    sc.bind_texture(tex)
    # straight down onto the quad; with r2 = U every diffuse bounce continues upwards and most hit the light
    n = 4000
    o = np.tile(np.float32([-250, 400, -250]), (n, 1))   # uv = (0.5, 0.5): mx = round(.5*3) = 2, my = round(.5*1) = 1... 
    d = np.tile(np.float32([0, -1, 0]), (n, 1))
    rad, _ = sc.radiance(o, d, va.make_opts(seed=1, sampling=va.VMX_SAMPLING_CORRECTED))
    lit = rad[:, 0] > 0
    assert lit.mean() > 0.05
    # k diffuse bounces on the quad before the light: green texel 1 -> 2, blue texel .25 -> 2*.25^k,
    # red = 2 * product of texels; one-bounce paths sample texel (my = 1, mx = round(.5*3) = 2) = 0.7
    assert np.allclose(rad[lit, 1], 2.0)
    k = np.round(np.log(rad[lit, 2] / 2.0) / np.log(0.25)).astype(int)
    assert k.min() == 1 and np.allclose(rad[lit, 2], 2.0 * 0.25 ** k)
    one = rad[lit][k == 1]
    assert len(one) > 50 and np.allclose(one[:, 0], 1.4)  # my = round(0.5 * 1) = 1 (half away from zero): row 1
    # without a texture the same paths return the untinted light
    sc2 = O.OracleScene(pos, nrm, uv, spheres=lights)
    rad2, _ = sc2.radiance(o, d, va.make_opts(seed=1, sampling=va.VMX_SAMPLING_CORRECTED))
    assert np.array_equal(rad2[:, 0] > 0, lit) and np.allclose(rad2[lit, :3], 2.0)
    with pytest.raises(ValueError):
        sc2.bind_texture(np.zeros((2, 2, 5), np.float32))


def test_bvh_statistics_match_the_survey_probe():
    """SURVEY App. C ran the reference's BVH::build on sine heightfields: 8 tris -> 3 nodes / 2 leaves;
    69,938 tris -> 47,871 nodes / 23,936 leaves (0.6845 nodes per triangle); 260,100 -> 130,785 / 65,393
    (0.5028).  The survey's exact generator is not recorded, so only the 8-triangle case can be matched
    exactly; the large grids must land within 0.5 % of the recorded node density."""
    sc = O.OracleScene(*scenes.heightfield(2))
    assert sc.describe()["n_nodes"] == 3 and sc.describe()["n_leaves"] == 2
    d = O.OracleScene(*scenes.heightfield(187)).describe()
    assert abs(d["n_nodes"] / 69938 - 47871 / 69938) < 0.005 and abs(d["n_leaves"] / 69938 - 23936 / 69938) < 0.003
    assert d["n_nodes"] == 2 * d["n_leaves"] - 1
    # traversal statistics of downward rays from one origin: ~14 inner visits, ~4-5 triangle tests per ray
    r = np.random.RandomState(0)
    dd = np.stack([r.uniform(-.5, .5, 20000), -np.ones(20000), r.uniform(-.5, .5, 20000)], 1)
    dd /= np.linalg.norm(dd, axis=1, keepdims=True)
    oo = np.tile(np.float32([0, 900, 0]), (20000, 1))
    tri, t, c = O.OracleScene(*scenes.heightfield(187)).trace(oo, dd.astype(np.float32), counters=True)
    assert (tri >= 0).mean() > 0.99
    assert 10 < c["inner_visits"] / 20000 < 22 and 3 < c["tri_tests"] / 20000 < 7 and c["max_stack"] <= 12


# ---- BruteForceTracer restatement (integrators.cpp:9-186) -----------------------------------
def test_bruteforce_known_answers():
    """hand-derived: a camera looking straight down at the floor sphere from the middle of the room —
    every ray hits the floor (y = 0, normal +y): N.L = L.y with L = normalize((500,1100,2000) - hit), the
    mirror probe leaves through nothing but walls (always a hit), so the colour is the constant albedo
    (0.890196078, 0.258823529, 0.203921569) * N.L (:148-156), alpha 1, and the break (:166-172) fires at
    the third sample because neighbouring sub-pixel samples differ by far less than 0.001"""
    pos, nrm, uv = scenes.cornell8()
    far = (pos + np.float32([5000, 0, 0] * 3)).astype(np.float32)  # move the mesh out of the way (outside the room)
    sc = O.OracleScene(far, nrm, uv)
    cam = va.make_camera((0, 500, 0), (90, 0, 0), 8, 8, 16, back_distance=6.0, back_size=(0.2, 0.2))  # looks straight down
    img, st = sc.render_bruteforce(cam, va.make_opts(seed=1))
    assert st["samples"] == 3 * 64 and np.all(img[:, :, 3] == 1.0)
    # the floor sphere's catastrophic cancellation (r = 5e7) puts the hit at ~499.29 instead of 500
    assert np.all(np.abs(img[:, :, 4] - 499.3) < 0.7)
    L = np.array([500.0, 1100.0, 2000.0]) - np.array([0.0, 0.0, 0.0])
    ndl = L[1] / np.linalg.norm(L)
    want = np.array([0.890196078, 0.258823529, 0.203921569]) * ndl
    assert np.allclose(img[:, :, :3].reshape(-1, 3), want, atol=4e-3)  # (the hits are ~0.6 above y = 0 and up to 8 off the axis)
    # the other reading of `abs` (:170): abs(int) truncates, so every |sum| < 1 counts as converged
    img2, st2 = sc.render_bruteforce(cam, va.make_opts(seed=1), flags=va._lib.VMX_BF_ABS_INT)
    assert st2["samples"] == 3 * 64
    # no hit at all (no spheres, mesh far away): accum = 0, alpha 0, depth INFINITY, 3 samples
    none = (va._lib.Sphere * 0)()
    sc0 = O.OracleScene(far, nrm, uv, spheres=none)
    img0, st0 = sc0.render_bruteforce(cam, va.make_opts(seed=1))
    assert np.all(img0[:, :, :4] == 0) and np.all(np.isinf(img0[:, :, 4])) and st0["samples"] == 3 * 64
    assert st0["rays_secondary"] == 0


def test_bruteforce_break_needs_more_than_two_samples_and_is_seed_keyed():
    pos, nrm, uv = scenes.lattice()
    sc = O.OracleScene(pos, nrm, uv)
    c = scenes.lattice_camera()
    for spp in (1, 2, 3, 7):
        img, st = sc.render_bruteforce(va.make_camera(c["position"], c["rotation_deg"], 24, 16, spp), va.make_opts(seed=5))
        assert st["samples"] == 24 * 16 * spp if spp <= 2 else 24 * 16 * 3 <= st["samples"] <= 24 * 16 * spp
    cam = va.make_camera(c["position"], c["rotation_deg"], 24, 16, 12)
    a, _ = sc.render_bruteforce(cam, va.make_opts(seed=5))
    b, _ = sc.render_bruteforce(cam, va.make_opts(seed=5), threads=1)
    d, _ = sc.render_bruteforce(cam, va.make_opts(seed=6))
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))  # keyed streams: thread-count independent
    assert not np.array_equal(a.view(np.uint32), d.view(np.uint32))


# ---- the two readings of cos(r1) / sin(r1) with float r1 (pathtracer.cpp:155,162) --------------------------
TWO_PI_BITS = int(np.float32(2 * np.pi).view(np.uint32))  # r1 = float(2 pi U) lies in [0, float(2 pi)]


def test_restated_cosf_sinf_equal_the_host_libm_on_every_float_of_the_range():
    """The default reading evaluates cosf / sinf by glibc's algorithm, restated in the oracle (and in the
    kernels).  The pin: all 1,086,918,620 floats r1 can be, against this image's libm."""
    dc, ds = O.trig_compare_libm(0, TWO_PI_BITS)
    assert (dc, ds) == (0, 0)


def test_trig_known_answers():
    x = np.array([0.0, 2.0 ** -13, 0.5, np.pi / 4, 1.0, np.pi / 2, 3.0, np.pi, 4.5, 6.0, 2 * np.pi], np.float32)
    cs, sn = O.trig(x)
    assert cs[0] == 1.0 and sn[0] == 0.0 and cs[1] == 1.0 and sn[1] == x[1]
    # within one float ulp of the correctly rounded values (glibc documents 0.56 ulp)
    assert np.all(np.abs(cs.astype(np.float64) - np.cos(x.astype(np.float64))) <= np.spacing(np.abs(cs)) + 1e-45)
    assert np.all(np.abs(sn.astype(np.float64) - np.sin(x.astype(np.float64))) <= np.spacing(np.maximum(np.abs(sn), 1e-30)))


def test_count_paths_that_differ_between_the_two_trig_readings(cornell, capsys):
    """How much the open question matters: of 10^6 paths (each reading's radiance on the same rays and streams) the
    readings differ on next to none — cosf/sinf and the narrowed double functions disagree in the last bit
    for ~1.3 % of the arguments (uniform in [0, 2 pi)), but a path returns sums of the light spheres' constant
    colours: a direction that moves by one ulp changes the result only if it changes WHAT the next ray hits."""
    g = np.load(os.path.join(GOLD, "cornell8.npz"))
    n = 1_000_000
    reps = (n + len(g["primary_o"]) - 1) // len(g["primary_o"])
    o = np.tile(g["primary_o"], (reps, 1))[:n]
    d = np.tile(g["primary_d"], (reps, 1))[:n]  # same camera rays, a different stream per path index
    rows = []
    for sampling in (va.VMX_SAMPLING_PARITY, va.VMX_SAMPLING_CORRECTED):
        a, _ = cornell.radiance(o, d, va.make_opts(seed=77, sampling=sampling))
        b, _ = cornell.radiance(o, d, va.make_opts(seed=77, sampling=sampling | va.VMX_SAMPLING_LIBM_DOUBLE))
        diff = int(np.any(bits(a) != bits(b), axis=1).sum())
        rows.append((sampling, diff))
    with capsys.disabled():
        print(f"\n[trig readings] paths of {n} whose radiance differs: parity sampling {rows[0][1]}, corrected sampling {rows[1][1]}")
    assert rows[0][1] < n // 1000 and rows[1][1] < n // 50
    # the argument-level rate, for the record: share of r1 values (uniform in [0, 2 pi)) on which the readings differ
    r1 = (np.float32(2 * np.pi) * np.random.RandomState(5).random_sample(2_000_000)).astype(np.float32)
    cs, sn = O.trig(r1)
    dc = float(np.mean(cs != np.cos(r1.astype(np.float64)).astype(np.float32)))
    ds = float(np.mean(sn != np.sin(r1.astype(np.float64)).astype(np.float32)))
    with capsys.disabled():
        print(f"[trig readings] arguments on which cosf / sinf differ from the narrowed double functions: {dc:.2e} / {ds:.2e}")
    assert 1e-3 < dc < 5e-2 and 1e-3 < ds < 5e-2  # ~1.3 %: the flag does change the arithmetic
