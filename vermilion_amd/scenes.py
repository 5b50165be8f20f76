"""Procedural stand-in scenes: the reference ships no assets and
there is no network, so the Cornell / Bunny / Sponza configurations of
BASELINE.json are generated here, deterministically, as plain triangle arrays
in createBVH order (mesh-major, face-minor; core/engines/meshEngine.cpp:660-718).

Every scene sits inside the reference's hard-coded sphere "room"
(x,z in [-2000,2000], y in [0,1000]; meshEngine.cpp:444-499) and is lit by the
reference's two light spheres (meshEngine.cpp:377,410) unless a sphere table is
passed to the Scene explicitly.

Returned arrays: pos[n,9], nrm[n,9], uv[n,6] float32.
"""
import numpy as np


def _finish(P, N, T):
    return (np.ascontiguousarray(P, np.float32).reshape(-1, 9),
            np.ascontiguousarray(N, np.float32).reshape(-1, 9),
            np.ascontiguousarray(T, np.float32).reshape(-1, 6))


def _quad(p00, p10, p11, p01, n):
    """two triangles (p00,p10,p11), (p00,p11,p01) with a constant normal"""
    p00, p10, p11, p01, n = [np.asarray(v, np.float32) for v in (p00, p10, p11, p01, n)]
    P = np.stack([np.concatenate([p00, p10, p11]), np.concatenate([p00, p11, p01])])
    N = np.tile(np.concatenate([n, n, n]), (2, 1))
    T = np.array([[0, 0, 1, 0, 1, 1], [0, 0, 1, 1, 0, 1]], np.float32)
    return P, N, T


def cornell8():
    """S-cornell: 8 triangles = floor quad, back quad, front + top of a block."""
    parts = [
        _quad((-600, 1, 600), (600, 1, 600), (600, 1, -800), (-600, 1, -800), (0, 1, 0)),      # floor
        _quad((-600, 1, -800), (600, 1, -800), (600, 900, -800), (-600, 900, -800), (0, 0, 1)),  # back
        _quad((-250, 1, 0), (150, 1, 0), (150, 400, 0), (-250, 400, 0), (0, 0, 1)),            # block front
        _quad((-250, 400, 0), (150, 400, 0), (150, 400, -400), (-250, 400, -400), (0, 1, 0)),  # block top
    ]
    P = np.concatenate([p[0] for p in parts])
    N = np.concatenate([p[1] for p in parts])
    T = np.concatenate([p[2] for p in parts])
    return _finish(P, N, T)


def cornell_camera():
    """position / rotation (degrees) looking down -z at the block"""
    return dict(position=(0.0, 420.0, 1900.0), rotation_deg=(0.0, 0.0, 0.0))


def _grid_surface(fn, nu, nv, u0=0.0, u1=1.0, v0=0.0, v1=1.0, flip=False):
    """Tessellate the parametric surface fn(u,v)->(x,y,z) into 2*nu*nv triangles
    with smooth per-vertex normals from central differences."""
    u = np.linspace(u0, u1, nu + 1, dtype=np.float64)
    v = np.linspace(v0, v1, nv + 1, dtype=np.float64)
    U, V = np.meshgrid(u, v, indexing="ij")
    X = np.stack(fn(U, V), axis=-1)  # [nu+1, nv+1, 3]
    eu, ev = (u1 - u0) / nu * 0.5, (v1 - v0) / nv * 0.5
    Xu = np.stack(fn(U + eu, V), -1) - np.stack(fn(U - eu, V), -1)
    Xv = np.stack(fn(U, V + ev), -1) - np.stack(fn(U, V - ev), -1)
    Nn = np.cross(Xu, Xv)
    ln = np.linalg.norm(Nn, axis=-1, keepdims=True)
    Nn = np.where(ln > 1e-20, Nn / np.maximum(ln, 1e-20), np.array([0.0, 1.0, 0.0]))
    if flip:
        Nn = -Nn
    UV = np.stack([(U - u0) / (u1 - u0), (V - v0) / (v1 - v0)], -1)

    def corner(a, di, dj):
        return a[di:nu + di, dj:nv + dj].reshape(nu * nv, -1)

    tris_p, tris_n, tris_t = [], [], []
    for (c0, c1, c2) in (((0, 0), (1, 0), (1, 1)), ((0, 0), (1, 1), (0, 1))):
        tris_p.append(np.concatenate([corner(X, *c0), corner(X, *c1), corner(X, *c2)], 1))
        tris_n.append(np.concatenate([corner(Nn, *c0), corner(Nn, *c1), corner(Nn, *c2)], 1))
        tris_t.append(np.concatenate([corner(UV, *c0), corner(UV, *c1), corner(UV, *c2)], 1))
    # interleave the two triangles of each cell (face-minor order)
    P = np.stack(tris_p, 1).reshape(-1, 9)
    N = np.stack(tris_n, 1).reshape(-1, 9)
    T = np.stack(tris_t, 1).reshape(-1, 6)
    return P, N, T


def _box(lo, hi):
    """12 triangles of an axis-aligned box"""
    x0, y0, z0 = lo
    x1, y1, z1 = hi
    q = [
        _quad((x0, y0, z1), (x1, y0, z1), (x1, y1, z1), (x0, y1, z1), (0, 0, 1)),
        _quad((x1, y0, z0), (x0, y0, z0), (x0, y1, z0), (x1, y1, z0), (0, 0, -1)),
        _quad((x0, y0, z0), (x0, y0, z1), (x0, y1, z1), (x0, y1, z0), (-1, 0, 0)),
        _quad((x1, y0, z1), (x1, y0, z0), (x1, y1, z0), (x1, y1, z1), (1, 0, 0)),
        _quad((x0, y1, z1), (x1, y1, z1), (x1, y1, z0), (x0, y1, z0), (0, 1, 0)),
        _quad((x0, y0, z0), (x1, y0, z0), (x1, y0, z1), (x0, y0, z1), (0, -1, 0)),
    ]
    return (np.concatenate([a[0] for a in q]), np.concatenate([a[1] for a in q]),
            np.concatenate([a[2] for a in q]))


def _blob(centre, radius, nu, nv, seed, amp=0.18):
    """displaced lat-long sphere: radial sum-of-sines noise, poles trimmed"""
    rng = np.random.RandomState(seed)
    k = rng.randint(2, 9, size=(6, 2)).astype(np.float64)
    ph = rng.uniform(0, 2 * np.pi, size=6)
    a = rng.uniform(0.3, 1.0, size=6)
    a = a / a.sum() * amp
    cx, cy, cz = centre

    def fn(U, V):
        th = U * 2 * np.pi
        phi = 0.02 + V * (np.pi - 0.04)
        r = np.ones_like(U)
        for i in range(6):
            r = r + a[i] * np.sin(k[i, 0] * th + ph[i]) * np.sin(k[i, 1] * phi)
        r = r * radius
        return cx + r * np.sin(phi) * np.cos(th), cy + r * np.cos(phi), cz + r * np.sin(phi) * np.sin(th)

    return _grid_surface(fn, nu, nv)


def bunny70k(seed=1):
    """S-bunny70k: a noisy blob (2*187*187 = 69,938 triangles) standing in the
    Cornell-like set of cornell8() — 69,946 triangles in all."""
    parts = [cornell8(), _blob((-40.0, 640.0, -150.0), 230.0, 187, 187, seed)]
    return _finish(np.concatenate([p[0].reshape(-1, 9) for p in parts]),
                   np.concatenate([p[1].reshape(-1, 9) for p in parts]),
                   np.concatenate([p[2].reshape(-1, 6) for p in parts]))


def bunny_camera():
    return dict(position=(0.0, 520.0, 1900.0), rotation_deg=(0.0, 0.0, 0.0))


def sponza260k(seed=1):
    """S-sponza260k: a procedural two-storey atrium, x in [-1500,1500], y in
    [0,1000], z in [-900,900]: tessellated floor, colonnades (cylinders with
    entasis), arches, hanging curtains (sine sheets), vases, blobs, plus large
    wall/beam triangles and long thin railings that overlap many cells — the
    features that make Sponza's BVH hard for a median split."""
    rng = np.random.RandomState(seed)
    parts = []

    def add(p):
        parts.append(p)

    # floor: bumpy tessellated sheet, y ~ 2
    add(_grid_surface(lambda U, V: (-1500 + 3000 * U, 2 + 1.5 * np.sin(40 * U) * np.sin(31 * V), 900 - 1800 * V),
                      128, 96))
    # upper gallery slabs (two long balconies), y = 480
    for zc in (-640.0, 640.0):
        add(_grid_surface(lambda U, V, zc=zc: (-1400 + 2800 * U, 480 + 2 * np.sin(25 * U), zc - 130 + 260 * V),
                          96, 16))
    # colonnades: 2 storeys x 2 sides x 12 columns
    for storey, (y0, h) in enumerate(((2.0, 460.0), (484.0, 400.0))):
        for zc in (-520.0, 520.0):
            for i in range(12):
                xc = -1265.0 + 230.0 * i
                rad = 34.0 - 6.0 * storey

                def col(U, V, xc=xc, zc=zc, y0=y0, h=h, rad=rad):
                    th = U * 2 * np.pi
                    r = rad * (1.0 + 0.12 * np.sin(np.pi * V)) * (1 + 0.04 * np.cos(12 * th))
                    return xc + r * np.cos(th), y0 + h * V, zc + r * np.sin(th)

                add(_grid_surface(col, 32, 24))
                # arch from this column to the next one
                if i < 11:
                    def arch(U, V, xc=xc, zc=zc, y0=y0, h=h):
                        a = np.pi * U
                        cxm, R = xc + 115.0, 115.0 - 30.0
                        tube = 18.0
                        ang = 2 * np.pi * V
                        rr = R + tube * np.cos(ang)
                        return cxm - rr * np.cos(a), y0 + h - 20 + rr * np.sin(a) * 0.6, zc + tube * np.sin(ang)

                    add(_grid_surface(arch, 24, 16))
    # curtains: 8 hanging sine sheets across the nave
    for i in range(8):
        xc = -1225.0 + 350.0 * i
        ph = rng.uniform(0, 6.28)

        def curtain(U, V, xc=xc, ph=ph):
            z = -380 + 760 * U
            return xc + 22 * np.sin(9 * U * np.pi + ph) * (0.3 + V), 930 - 420 * V - 30 * np.sin(np.pi * U), z

        add(_grid_surface(curtain, 64, 64))
    # vases along the nave
    for i in range(16):
        xc = -1320.0 + 176.0 * i
        zc = 250.0 if i % 2 else -250.0

        def vase(U, V, xc=xc, zc=zc):
            th = U * 2 * np.pi
            r = 26 + 16 * np.sin(2.6 * V + 0.4) ** 2
            return xc + r * np.cos(th), 3 + 110 * V, zc + r * np.sin(th)

        add(_grid_surface(vase, 40, 20))
    # four noisy blobs ("lion heads") on the end walls
    for j, (xc, zc) in enumerate(((-1380.0, -300.0), (-1380.0, 300.0), (1380.0, -300.0), (1380.0, 300.0))):
        add(_blob((xc, 330.0, zc), 95.0, 56, 56, seed + 10 + j, amp=0.25))
    # large structural triangles: end walls and side walls behind the colonnades
    add(_quad((-1500, 0, -900), (1500, 0, -900), (1500, 1000, -900), (-1500, 1000, -900), (0, 0, 1)))
    add(_quad((1500, 0, 900), (-1500, 0, 900), (-1500, 1000, 900), (1500, 1000, 900), (0, 0, -1)))
    add(_quad((-1500, 0, 900), (-1500, 0, -900), (-1500, 1000, -900), (-1500, 1000, 900), (1, 0, 0)))
    add(_quad((1500, 0, -900), (1500, 0, 900), (1500, 1000, 900), (1500, 1000, -900), (-1, 0, 0)))
    # ceiling beams (long boxes spanning the nave) and long thin railings
    for i in range(20):
        xc = -1425.0 + 150.0 * i
        add(_box((xc - 12, 940, -880), (xc + 12, 975, 880)))
    for zc in (-512.0, 512.0):
        for i in range(60):
            y = 500.0 + 1.5 * (i % 20)
            x0 = -1400.0 + 46.6 * i
            add(_box((x0, y + 60, zc - 3), (min(x0 + 900.0, 1450.0) if i % 3 == 0 else x0 + 44.0, y + 66, zc + 3)))
    P = np.concatenate([p[0].reshape(-1, 9) for p in parts])
    N = np.concatenate([p[1].reshape(-1, 9) for p in parts])
    T = np.concatenate([p[2].reshape(-1, 6) for p in parts])
    return _finish(P, N, T)


def sponza_camera():
    """inside the atrium near one end, looking down the nave (-x to +x)"""
    return dict(position=(-1350.0, 260.0, 40.0), rotation_deg=(0.0, 82.0, 0.0))


def heightfield(n, extent=1600.0, height=120.0, y0=200.0):
    """sine heightfield of 2*n*n triangles (the survey's probe scenes were of this kind)"""
    return _finish(*_grid_surface(
        lambda U, V: (-extent / 2 + extent * U, y0 + height * np.sin(7 * U) * np.cos(5 * V), extent / 2 - extent * V),
        n, n))


def lattice(n=10):
    """2*n*n tilted quads' triangles whose coordinates are small integers (exact in
    float32, no transcendental functions): identical on every machine, used for
    the committed golden fixtures."""
    P, N, T = [], [], []
    for i in range(n):
        for j in range(n):
            x0, z0 = -900 + 180 * i, -700 + 140 * j
            y0 = 40 + 7 * ((i * 3 + j * 5) % 11)
            y1 = y0 + 9 * ((i + 2 * j) % 5)
            a, b = (x0, y0, z0), (x0 + 150, y0, z0 + 10)
            c, d = (x0 + 160, y1 + 30, z0 + 120), (x0 - 5, y1 + 25, z0 + 115)
            nn = (0.0, 1.0, 0.0) if (i + j) % 2 else (0.6, 0.8, 0.0)
            p, nq, t = _quad(a, b, c, d, nn)
            P.append(p), N.append(nq), T.append(t)
    return _finish(np.concatenate(P), np.concatenate(N), np.concatenate(T))


def lattice_camera():
    return dict(position=(0.0, 700.0, 1500.0), rotation_deg=(20.0, 0.0, 0.0))


SCENES = {
    "lattice": (lattice, lattice_camera),
    "cornell8": (cornell8, cornell_camera),
    "bunny70k": (bunny70k, bunny_camera),
    "sponza260k": (sponza260k, sponza_camera),
}
