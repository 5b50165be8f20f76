"""BruteForceTracer (core/integrators/integrators.cpp:9-186), the engine's default
integrator, on the device (vmx_render_bruteforce -> k_bruteforce) against the oracle's restatement:
whole frames bit for bit, with 0, 1 and 2 bound textures, both readings of `abs` (:170), sharded."""
import numpy as np
import pytest

import oracle_lib as O
import vermilion_amd as va
from vermilion_amd import scenes

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def same(a, b):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    return bool(np.all((bits(a) == bits(b)) | (np.isnan(a) & np.isnan(b))))


def tex(h, w, c, seed, lo=-0.4, hi=0.4):
    return np.random.RandomState(seed).uniform(lo, hi, size=(h, w, c)).astype(np.float32)


@pytest.mark.parametrize("name", ["cornell8", "lattice", "bunny70k", "sponza260k"])
@pytest.mark.parametrize("ntex", [0, 1, 2])
def test_bruteforce_frames_bit_exact(name, ntex):
    gen, camf = scenes.SCENES[name]
    pos, nrm, uv = gen()
    uv = uv * np.float32(2.3) - np.float32(0.6)  # leaves [0,1]: exercises VermiTexture::Sample's wrap
    g, c = va.Scene(pos, nrm, uv), O.OracleScene(pos, nrm, uv)
    if ntex >= 1:  # boundTextures[0] perturbs the normal (:98-106)
        t0 = tex(16, 24, 3, 1)
        g.bind_texture(t0), c.bind_texture(t0)
    if ntex >= 2:  # boundTextures[1] is the albedo (:141-147)
        t1 = tex(9, 5, 4, 2, 0.1, 1.0)
        g.bind_texture(t1), c.bind_texture(t1)
    cc = camf()
    W, H, spp = (160, 96, 24) if name in ("bunny70k", "sponza260k") else (128, 128, 24)
    cam = va.make_camera(cc["position"], cc["rotation_deg"], W, H, spp, back_size=(3.6, 3.6 * H / W))
    for flags in (0, va._lib.VMX_BF_ABS_INT):
        opts = va.make_opts(seed=13)
        img, st = g.render_bruteforce(cam, opts, flags)
        ref, rst = c.render_bruteforce(cam, opts, flags)
        assert img.shape == (H, W, 5)
        assert same(img, ref), f"{int((bits(img) != bits(ref)).any(axis=2).sum())} pixels differ (flags {flags})"
        assert st["samples"] == rst["samples"] and st["rays_primary"] == rst["rays_primary"] == st["samples"]
        assert st["rays_secondary"] == rst["rays_secondary"]
        if flags == 0:
            # the convergence break (:166-172) fires early on most pixels, and not on all of them
            assert 3 * W * H <= st["samples"] < spp * W * H
            assert 0.0 < img[:, :, :3].mean() < 1.0 and np.all(img[:, :, 3] == 1.0)  # the room is closed: every ray hits
        else:
            # |int(x)| < 0.001 for every |x| < 1: (nearly) every pixel stops at its 3rd sample
            assert 3 * W * H <= st["samples"] <= 3 * W * H + W * H // 100
    # sharded rendering returns the rank's rows of the same frame
    for r in range(3):
        part, _ = g.render_bruteforce(cam, va.make_opts(seed=13, rank=r, world=3, stripe_rows=7), 0)
        full, _ = g.render_bruteforce(cam, va.make_opts(seed=13), 0)
        assert same(part, full[va.local_row_indices(H, 7, r, 3)])
    g.close(), c.close()


def test_bruteforce_misses_and_depth_channel():
    """no sphere table: rays that leave the mesh miss -> alpha < 1, depth = INFINITY after a miss (meshEngine.cpp:507)"""
    pos, nrm, uv = scenes.cornell8()
    none = (va._lib.Sphere * 0)()
    g, c = va.Scene(pos, nrm, uv, spheres=none), O.OracleScene(pos, nrm, uv, spheres=none)
    cc = scenes.cornell_camera()
    cam = va.make_camera(cc["position"], cc["rotation_deg"], 96, 64, 16)
    img, st = g.render_bruteforce(cam, va.make_opts(seed=3))
    ref, rst = c.render_bruteforce(cam, va.make_opts(seed=3))
    assert same(img, ref) and st["samples"] == rst["samples"]
    assert np.isinf(img[:, :, 4]).any() and np.isfinite(img[:, :, 4]).any()
    assert img[:, :, 3].min() == 0.0 and img[:, :, 3].max() == 1.0
    # a pixel whose samples all miss: accum stays 0, the break fires at the third sample
    miss = np.isinf(img[:, :, 4]) & (img[:, :, 3] == 0)
    assert miss.any() and np.all(img[miss][:, :3] == 0)
    g.close(), c.close()


def test_bruteforce_argument_checks():
    pos, nrm, uv = scenes.cornell8()
    with va.Scene(pos, nrm, uv) as sc:
        cc = scenes.cornell_camera()
        for spp in (1, 2, 3, 5):  # no spp/4 rule here (that is PathTracer's, pathtracer.cpp:247)
            img, st = sc.render_bruteforce(va.make_camera(cc["position"], cc["rotation_deg"], 16, 8, spp), va.make_opts())
            assert st["samples"] <= 16 * 8 * spp and np.all(np.isfinite(img[:, :, :4]))
        with pytest.raises(va.VmxError):
            sc.render_bruteforce(va.make_camera(cc["position"], cc["rotation_deg"], 16, 8, 70000), va.make_opts())
        with pytest.raises(va.VmxError):
            sc.render_bruteforce(va.make_camera(cc["position"], cc["rotation_deg"], 16, 8, 8), va.make_opts(), flags=6)


def test_bruteforce_probe_asks_the_spheres_before_the_tree():
    """Round 3: the mirror probe's RayCast (integrators.cpp:121) is only asked whether it hit, so the kernel asks the sphere
    table first and walks the BVH only for probes that hit no sphere.  A partial table (two lights and the floor) makes
    both routes decide pixels of the same frame: probes going down hit the floor sphere, the others only the mesh, or
    nothing."""
    pos, nrm, uv = scenes.bunny70k()
    some = va.spheres_array([
        dict(centre=(15, 140, 25), radius=3.5, colour=(0, 7.5, 15), emit=True, normal_centre=(-55, 350, -150), normal_sign=-1),
        dict(centre=(0, 3300, 1300), radius=250, colour=(15.2, 15.2, 15.2), emit=True, normal_centre=(500, 800, 1300)),
        dict(centre=(0, -5e7, 0), radius=5e7)])
    g, c = va.Scene(pos, nrm, uv, spheres=some), O.OracleScene(pos, nrm, uv, spheres=some)
    cc = scenes.bunny_camera()
    cam = va.make_camera(cc["position"], cc["rotation_deg"], 192, 128, 32, back_size=(3.6, 2.4))
    img, st = g.render_bruteforce(cam, va.make_opts(seed=21))
    ref, rst = c.render_bruteforce(cam, va.make_opts(seed=21))
    assert same(img, ref) and st["samples"] == rst["samples"] and st["rays_secondary"] == rst["rays_secondary"]
    assert 0.0 < img[:, :, 3].min() < 1.0 or np.isinf(img[:, :, 4]).any()  # some camera rays leave the scene
    g.close(), c.close()
