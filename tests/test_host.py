"""Host-side logic that needs no GPU: the mirror of the reference's plugin
surface, the synthetic scenes and the stripe sharding."""
import os

import numpy as np
import pytest

import vermilion_amd as va
from vermilion_amd import dist as vdist
from vermilion_amd import scenes


def test_default_camera_is_the_reference_default():
    """RenderEngine::CreateInternalDefaultCamera, renderEngine.cpp:115-145"""
    s = va.cameraSettings()
    assert (s.imageResX, s.imageResY) == (648, 432) and s.raysPerPixel == 128
    assert tuple(s.position) == (-4000.0, 1600.0, 8000.0) and tuple(s.rotation) == (0.0, 25.0, 0.0)
    assert (s.fBackDistance, s.fBackSizeX, s.fBackSizeY) == (6.0, 3.6, 2.4)
    assert s.renderMode == va.vermRenderMode.RGBAZ
    cam = va.Camera(s)
    assert cam.RenderTargetSize == 648 * 432 and cam.mImage.size == 648 * 432 * 5
    d = cam._desc()
    assert list(d.image_res) == [648, 432] and d.rays_per_pixel == 128 and d.back_distance == 6.0


def test_set_pixel_value_layouts():
    """Camera::setPixelValue, camera.cpp:88-124"""
    pv = va.pixelValue(pixel=3, red=.1, green=.2, blue=.3, alpha=1.0, depth=7.0)
    for mode, ch in ((va.vermRenderMode.RGB, 3), (va.vermRenderMode.RGBA, 4), (va.vermRenderMode.RGBAZ, 5)):
        cam = va.Camera(va.cameraSettings(imageResX=4, imageResY=2, renderMode=mode))
        cam.setPixelValue(pv)
        assert np.allclose(cam.mImage[3 * ch:3 * ch + ch], [.1, .2, .3, 1.0, 7.0][:ch])
        assert cam.mImage.sum() == pytest.approx(sum([.1, .2, .3, 1.0, 7.0][:ch]), rel=1e-6)
    cam = va.Camera(va.cameraSettings(imageResX=4, imageResY=2, renderMode=va.vermRenderMode.Depth))
    cam.setPixelValue(pv)
    assert cam.mImage[3] == 7.0
    with pytest.raises(Exception):  # camera.cpp:74-80 throws for Depth64
        va.Camera(va.cameraSettings(renderMode=va.vermRenderMode.Depth64))


def test_render_engine_seam():
    """assignIntegrator / draw, renderEngine.cpp:70-78, 147-166"""
    calls = []

    class Probe(va.Integrator):
        def Render(self, cameraList, mEng):
            calls.append((len(cameraList), mEng))

    r = va.RenderEngine()
    r.assignIntegrator(Probe())
    r.draw()
    assert calls == [] and "without mesh engine" in r.log[0]
    m = va.MeshEngine()
    r.assignEngine(m)
    r.draw()
    assert calls == [(1, m)] and "Defaulting" in r.log[1] and r.mCameras[0].uImageU == 648
    with pytest.raises(NotImplementedError):
        va.Integrator().Render([], None)
    with pytest.raises(RuntimeError):
        va.PathTracer().Render(r.mCameras, m)  # no scene loaded: fails, never renders on the CPU


def test_scene_sizes_and_determinism():
    p, n, t = scenes.cornell8()
    assert p.shape == (8, 9) and n.shape == (8, 9) and t.shape == (8, 6)
    p, n, t = scenes.lattice()
    assert p.shape == (200, 9) and np.all(p == np.round(p))
    p, n, t = scenes.bunny70k()
    assert p.shape[0] == 69946 and np.all(np.isfinite(p)) and np.all(np.isfinite(n))
    p2, _, _ = scenes.bunny70k()
    assert np.array_equal(p, p2)
    p, n, t = scenes.sponza260k()
    assert 250000 < p.shape[0] < 270000 and np.all(np.isfinite(p))
    v = p.reshape(-1, 3)
    assert v[:, 0].min() >= -1500.5 and v[:, 0].max() <= 1500.5 and v[:, 1].min() >= 0 and v[:, 1].max() <= 1000.5
    ln = np.linalg.norm(n.reshape(-1, 3), axis=1)
    assert np.all(ln > 0.99) and np.all(ln < 1.01)


def test_stripe_assembly_roundtrip():
    rng = np.random.RandomState(0)
    for W, H, R, world in ((7, 37, 4, 3), (16, 64, 16, 8), (5, 9, 16, 2), (3, 10, 1, 4)):
        frame = rng.rand(H, W, 5).astype(np.float32)
        mrows = vdist.max_local_rows(H, R, world)
        parts = []
        for r in range(world):
            rows = va.local_row_indices(H, R, r, world)
            part = np.zeros((mrows, W, 5), np.float32)
            part[:len(rows)] = frame[rows]
            parts.append(part)
        assert np.array_equal(vdist.assemble_host(parts, W, H, R, world), frame)


def test_make_opts_and_spheres():
    o = va.make_opts(seed=2**40 + 1, early_stop=False, sampling=1, rank=2, world=4, stripe_rows=8)
    assert (o.seed, o.early_stop, o.sampling, o.rank, o.world, o.stripe_rows) == (2**40 + 1, 0, 1, 2, 4, 8)
    arr = va.spheres_array([dict(centre=(1, 2, 3), radius=4, colour=(5, 6, 7), emit=True, normal_sign=-1)])
    assert list(arr[0].centre) == [1, 2, 3] and arr[0].flags == 1 and arr[0].normal_sign == -1
    assert list(arr[0].normal_centre) == [1, 2, 3]


def test_fastdiv_equals_integer_division(tmp_path):
    """vmx_device.h: FastDiv (multiply-high + two shifts instead of a ~25-instruction division by the samples per pixel /
    the image width in the kernels) is exact for every 32-bit numerator: awkward and random divisors, numerators around
    every multiple boundary and across the range"""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        import pytest
        pytest.skip("no g++")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "fastdiv_test"
    subprocess.run(["g++", "-std=c++17", "-O2", "-Wall", "-Werror", "-I", os.path.join(root, "vermilion_amd", "csrc"),
                    os.path.join(root, "tests", "cpp", "fastdiv_test.cpp"), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0 and "mismatches 0" in out.stdout, out.stdout
