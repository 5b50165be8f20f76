"""Exactness arguments behind two kernel shortcuts, restated in numpy (CPU only).

The four single-precision tests that let the kernels skip the double-precision sphere solve
(vermilion_amd/csrc/vmx_kernels.hip: sphere_hit_op), restated in numpy and checked against the exact
evaluation of the reference's formula (meshEngine.cpp:182-194): whenever a test fires, the exact result
must be one RayCast ignores — 0, or not below the nearest distance so far.  Rays start inside the
reference's sphere room, ON its walls (where (B*B - C) + R2 cancels at the magnitude of C), and the
limits are placed right around the exact roots."""
import numpy as np

f32 = np.float32
K_REL = f32(9.5367431640625e-07)  # 2^-20, as in the kernel

SPHERES = [((15, 140, 25), 3.5), ((0, 3300, 1300), 250), ((0, -5e7, 0), 5e7), ((0, 5e7 + 1000, 0), 5e7),
           ((-5e7 + 2000, 0, 0), 5e7), ((5e7 - 2000, 0, 0), 5e7), ((0, 0, -5e7 + 2000), 5e7),
           ((0, 0, 5e7 - 2000), 5e7), ((100, 200, 300), 50.0), ((0, 0, 0), 1e3), ((5, 5, 5), 1e-2)]


def dot3(a, b):  # glm::dot order, one rounding per operation
    return ((a[:, 0] * b[:, 0]).astype(f32) + (a[:, 1] * b[:, 1]).astype(f32)).astype(f32) + (a[:, 2] * b[:, 2]).astype(f32)


def exact_and_shortcuts(o, d, centre, radius, lim):
    c = np.array(centre, dtype=f32)
    r2 = f32(f32(radius) * f32(radius))
    op = (c[None, :] - o).astype(f32)
    B, C = dot3(op, d).astype(f32), dot3(op, op).astype(f32)
    b = B.astype(np.float64)
    det = b * b - C.astype(np.float64) + np.float64(r2)
    with np.errstate(invalid="ignore"):
        s = np.sqrt(det)
    t1, t2 = b - s, b + s
    th = np.where(det < 0, 0, np.where(t1 > 1e-4, t1, np.where(t2 > 1e-4, t2, 0))).astype(f32)
    X, BB = (C - r2).astype(f32), (B * B).astype(f32)
    with np.errstate(invalid="ignore", over="ignore"):
        tol_m = (K_REL * ((C + r2).astype(f32) + BB).astype(f32)).astype(f32)
        tol_f = (K_REL * (((C + r2).astype(f32) + BB).astype(f32)
                          + (lim * ((f32(2) * np.abs(B)).astype(f32) + lim).astype(f32)).astype(f32)).astype(f32)).astype(f32)
        Y = (X - (lim * ((f32(2) * B).astype(f32) - lim).astype(f32)).astype(f32)).astype(f32)
        far = (B > (lim * (f32(1) + K_REL)).astype(f32)) & (Y > tol_f)
        miss = BB < (X - tol_m).astype(f32)
        behind = (B < 0) & (B > -1e11) & (X > tol_m)
        inside_far = (X < -tol_m) & (Y < -tol_f)  # origin inside the sphere, its exit beyond the limit
    return th, miss | behind | far | inside_far


def check(o, d, rng):
    n = o.shape[0]
    fired = 0
    for centre, radius in SPHERES:
        th, _ = exact_and_shortcuts(o, d, centre, radius, np.full(n, np.inf, f32))
        delta = np.exp(rng.uniform(np.log(1e-8), np.log(1e-1), n)) * rng.choice([-1, 1], n)
        for lim in (np.full(n, np.inf, f32), (np.where(th > 0, th, rng.uniform(1, 3000, n)) * (1 + delta)).astype(f32)):
            th, skip = exact_and_shortcuts(o, d, centre, radius, lim)
            matters = (th > 0) & (th < lim)
            assert not np.any(skip & matters), (centre, radius)
            fired += int(skip.sum())
    return fired


def test_shortcuts_never_drop_a_result_that_matters():
    rng = np.random.default_rng(5)
    n = 150000
    # inside the room, aimed anywhere
    o = np.empty((n, 3), f32)
    o[:, 0], o[:, 1], o[:, 2] = rng.uniform(-1990, 1990, n), rng.uniform(1, 999, n), rng.uniform(-1990, 1990, n)
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    fired = check(o, d.astype(f32), rng)
    # on the walls (bounce rays start 0.001 off the surface they left), a quarter of them grazing
    for ax, val in ((0, -2000), (0, 2000), (2, -2000), (2, 2000), (1, 0), (1, 1000)):
        o2 = o.copy()
        o2[:, ax] = val + rng.uniform(-0.7, 0.7, n)
        d2 = rng.normal(size=(n, 3))
        d2[: n // 4, ax] *= 1e-5
        d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
        fired += check(o2, d2.astype(f32), rng)
    assert fired > 1000000  # the shortcuts do fire on this population


def test_exact_comparison_would_be_wrong_on_the_walls():
    """what the margin is for: with `C >= R2` instead of `(C - R2) > tol` the behind test drops real hits"""
    rng = np.random.default_rng(6)
    n = 400000
    o = np.empty((n, 3), f32)
    o[:, 0], o[:, 1], o[:, 2] = -2000 + rng.uniform(-0.7, 0.7, n), rng.uniform(0, 1000, n), rng.uniform(-2000, 2000, n)
    d = rng.normal(size=(n, 3))
    d[:, 0] *= 1e-5
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d = d.astype(f32)
    c = np.array((5e7 - 2000, 0, 0), dtype=f32)
    r2 = f32(f32(5e7) * f32(5e7))
    op = (c[None, :] - o).astype(f32)
    B, C = dot3(op, d).astype(f32), dot3(op, op).astype(f32)
    th, _ = exact_and_shortcuts(o, d, (5e7 - 2000, 0, 0), 5e7, np.full(n, np.inf, f32))
    naive_behind = (B < 0) & (C >= r2)
    assert np.any(naive_behind & (th > 0))


def test_octant_tables_near_plane_product_is_the_slab_minimum():
    """k_camera_tables stores, per direction octant, each axis' near plane first, and the wave-uniform
    step takes max3/min3 of the six products without min/max pairs: for lo <= hi and a finite
    reciprocal, min(lo*inv, hi*inv) is the product with lo when inv >= 0 and with hi otherwise
    (float multiplication by a constant is monotonic); only the sign of a zero may differ"""
    rng = np.random.default_rng(8)
    n = 2000000
    a = (rng.normal(size=n) * np.exp(rng.uniform(-20, 20, n))).astype(f32)
    w = np.abs(rng.normal(size=n) * np.exp(rng.uniform(-20, 20, n))).astype(f32)
    w[::7] = 0  # degenerate boxes
    lo, hi = a, (a + w).astype(f32)
    inv = (rng.normal(size=n) * np.exp(rng.uniform(-30, 30, n))).astype(f32)
    inv[::11] = 0
    inv[5::11] = -0.0
    with np.errstate(over="ignore", invalid="ignore"):
        t0, t1 = (lo * inv).astype(f32), (hi * inv).astype(f32)
        neg = np.signbit(inv)
        near = (np.where(neg, hi, lo) * inv).astype(f32)
        far = (np.where(neg, lo, hi) * inv).astype(f32)
    ok = ~(np.isnan(t0) | np.isnan(t1))  # inf * 0: such rays take the exact path in the kernels
    assert np.all((np.minimum(t0, t1) == near)[ok]) and np.all((np.maximum(t0, t1) == far)[ok])


def test_division_by_the_image_size_with_a_reciprocal_and_one_fma_is_the_ieee_division():
    """vmx_kernels.hip: div_by_count — pathtracer.cpp:251-252 divide the sample's film coordinate by the image size in
    double; the kernels multiply by the host's RN(1 / n) and correct once with an FMA (Markstein), which the comment
    there proves equal to the IEEE quotient for every finite numerator.  Here exhaustively: EVERY float a camera ray
    can produce as its coordinate (|fx| up to just past the image size, both signs, the tiny ones included) for the
    image sizes of BASELINE.json's configs and of the tests, and every float at all for a few awkward divisors."""
    import ctypes as C
    import struct
    import oracle_lib as O
    lib = O.lib(fast=False)

    def bits(x):
        return struct.unpack("<I", struct.pack("<f", x))[0]
    total = 0
    for n in (1920, 1080, 3840, 2160, 1024, 512, 256, 640, 360, 160, 110, 96, 64, 33, 17, 1, 3, 7):
        fd, dd = C.c_uint64(1), C.c_uint64(1)
        hi = bits(float(n) + 2.0)
        lib.orc_check_div_by_count(n, 0, hi, C.byref(fd), C.byref(dd))
        assert fd.value == 0 and dd.value == 0, (n, fd.value, dd.value)
        total += 2 * (hi + 1)
    for n in (0xFFFFFFFF, 4294967291, 16777217, 6700417):  # every finite float
        fd, dd = C.c_uint64(1), C.c_uint64(1)
        lib.orc_check_div_by_count(n, 0, 0x7F7FFFFF, C.byref(fd), C.byref(dd))
        assert fd.value == 0 and dd.value == 0, (n, fd.value, dd.value)
        total += 2 * 0x7F800000
    assert total > 1.7e10


def test_pixel_light_cull_is_conservative():
    """vmx_kernels.hip: pixel_may_reach_a_light — k_raygen's per-pixel answer to "can any camera ray of this pixel pass
    sphereIntersect > 0 (meshEngine.cpp:182-194) for this emitting sphere?".  Where it says NO, no ray of the pixel may
    reach the sphere: restated here in float32 as the kernel computes it and checked against the reference's own
    arithmetic (the oracle's camera rays of many samples per pixel, plus the corners and edge midpoints of the sample
    footprint) for random cameras and spheres placed all around the view, most of them right at the edge of it."""
    import oracle_lib as O
    import vermilion_amd as va
    rng = np.random.default_rng(11)

    def may_reach(cam_m, pos, film_dist, sensor, W, H, p, centre, rad2):
        f = np.float32
        hx = (f(p % W) - f(0.25)) / f(W) - f(0.5)
        hy = (f(p // W) - f(0.25)) / f(H) - f(0.5)
        gx, gy, gz = hx * f(sensor[0]), -(hy * f(sensor[1])), -f(film_dist)
        a = (cam_m[:, 0] * gx + cam_m[:, 1] * gy + cam_m[:, 2] * gz).astype(f)     # columns of the 3x3 matrix
        alen = np.sqrt(f(a @ a))
        px_, py_ = f(sensor[0]) / f(W), f(sensor[1]) / f(H)
        sphi = min(f(0.505) * np.sqrt(px_ * px_ + py_ * py_) / alen, f(1))
        cphi = np.sqrt(max(f(1) - sphi * sphi, f(0)))
        op = (np.float32(centre) - np.float32(pos)).astype(f)
        C = f(op @ op)
        ct = f(op @ a) / (np.sqrt(C) * alen)
        st = np.sqrt(max(f(1) - ct * ct, f(0)))
        m = f(1) if ct >= cphi else max(ct * cphi + st * sphi, f(0))
        clear = f(1) - f(rad2) / C
        return not (m * m * f(1.0001) + f(1e-3) < clear)

    def reaches(o, d, centre, rad):  # sphereIntersect > 0, the reference's mixed float / double arithmetic
        op = (np.float32(centre)[None, :] - o).astype(np.float32)
        B, C = dot3(op, d).astype(np.float64), dot3(op, op).astype(np.float64)
        det = B * B - C + np.float64(f32(f32(rad) * f32(rad)))
        with np.errstate(invalid="ignore"):
            s = np.sqrt(det)
        return (det >= 0) & ((B - s > 1e-4) | (B + s > 1e-4))

    culled = violations = tested = 0
    for trial in range(60):
        W, H = int(rng.choice([33, 64, 160])), int(rng.choice([17, 40, 90]))
        pos = rng.uniform(-1500, 1500, 3)
        rot = rng.uniform(-180, 180, 3) * np.array([0.4, 1.0, 0.2])
        cam = va.make_camera(pos, rot, W, H, 64, back_size=(3.6, 3.6 * H / W))
        M = O.camera_matrix(cam).T  # [row][col] -> columns M[:, c]
        rays = [O.primary_rays(cam, va.make_opts(seed=trial), k) for k in (0, 9, 17, 25, 33, 41, 50, 63)]
        # spheres aimed at random pixels' directions, at an angular offset around the edge of what the pixel sees
        o0, d0 = rays[0]
        for _ in range(12):
            p = int(rng.integers(0, W * H))
            dist = np.exp(rng.uniform(np.log(5), np.log(5000)))
            rad = dist * np.exp(rng.uniform(np.log(1e-3), np.log(0.5)))
            axis = d0[p].astype(np.float64)
            t = np.cross(axis, rng.normal(size=3))
            t /= np.linalg.norm(t)
            ang = np.arcsin(min(rad / dist, 1.0)) + rng.uniform(-0.05, 0.08)   # tangent direction +- a few degrees
            centre = pos + dist * (np.cos(ang) * axis + np.sin(ang) * t)
            rad2 = f32(f32(rad) * f32(rad))
            for q in {p, max(p - 1, 0), min(p + 1, W * H - 1), max(p - W, 0), min(p + W, W * H - 1)}:
                tested += 1
                if may_reach(np.float32(M), np.float32(pos), cam.back_distance, cam.back_size, W, H, q, centre, rad2):
                    continue
                culled += 1
                for o, d in rays:
                    if reaches(o[q:q + 1], d[q:q + 1], centre, rad)[0]:
                        violations += 1
                # the footprint's corners and edge midpoints (offsets [-0.75, 0.25] around the pixel coordinate, incl. the ends)
                for ex in (-0.75, -0.25, 0.25):
                    for ey in (-0.75, -0.25, 0.25):
                        hx = (q % W + ex) / W - 0.5
                        hy = (q // W + ey) / H - 0.5
                        g = np.array([hx * cam.back_size[0], -hy * cam.back_size[1], -cam.back_distance])
                        dd = M.astype(np.float64) @ g
                        dd /= np.linalg.norm(dd)
                        if reaches(np.float32(pos)[None, :], np.float32(dd)[None, :], centre, rad)[0]:
                            violations += 1
    assert violations == 0
    assert culled > 0.2 * tested and culled < 0.95 * tested  # the test exercises both answers
