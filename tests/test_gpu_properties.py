"""Size-independent properties at BASELINE.json's full sizes, where the CPU
oracle is too slow to run the whole frame: determinism, invariance of the
frame to every scheduling parameter (pass size, pipeline form, tail threshold,
stripe sharding), and sample-count invariants."""
import os

import numpy as np
import pytest

import oracle_lib as O
import vermilion_amd as va
from vermilion_amd import dist as vdist
from vermilion_amd import scenes

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.fixture(scope="module")
def sponza():
    sc = va.Scene(*scenes.sponza260k())
    yield sc
    sc.close()


def sponza_cam(W, H, spp):
    c = scenes.sponza_camera()
    return va.make_camera(c["position"], c["rotation_deg"], W, H, spp, back_size=(3.6, 3.6 * H / W))


def test_full_size_frame_invariants(sponza):
    """1920x1080 Sponza stand-in (BASELINE config 4 geometry) at 16 spp"""
    cam = sponza_cam(1920, 1080, 16)
    a, sa = sponza.render(cam, va.make_opts(seed=1, early_stop=False))
    assert np.all(a[:, :, 4] == 16.0) and np.all(a[:, :, 3] == 1.0)
    assert sa["samples"] == 1920 * 1080 * 16 == sa["rays_primary"]
    assert np.all(a[:, :, :3] >= 0) and np.all(a[:, :, :3] <= 1) and np.all(np.isfinite(a))
    # scheduling must not change a single bit
    for kw in (dict(samples_per_batch=3), dict(pipeline=1), dict(pipeline=4), dict(tail_threshold=1),
               dict(max_paths=1 << 20), dict(refill_min=1, shade_min=1), dict(refill_min=64, shade_min=64),
               dict(refill_min=5, shade_min=40),
               dict(lds_entries=40), dict(lds_entries=1), dict(collect_counters=True)):
        b, sb = sponza.render(cam, va.make_opts(seed=1, early_stop=False, **kw))
        assert np.array_equal(bits(a), bits(b)), kw
        assert sb["rays_secondary"] == sa["rays_secondary"]
    # early stop: the reference's sample-count pattern (one stratum at a time), same rays twice
    e1, s1 = sponza.render(cam, va.make_opts(seed=1, early_stop=True))
    e2, s2 = sponza.render(cam, va.make_opts(seed=1, early_stop=True, pipeline=4))
    assert np.array_equal(bits(e1), bits(e2)) and s1["rays_primary"] == s2["rays_primary"]
    assert set(np.unique(e1[:, :, 4])).issubset({7.0, 10.0, 13.0, 16.0})
    dark = e1[:, :, :3].sum(-1) == 0
    assert np.all(e1[:, :, 4][dark] == 7.0)
    assert s1["samples"] == int(e1[:, :, 4].sum())
    e3, s3 = sponza.render(cam, va.make_opts(seed=1, early_stop=True, max_paths=1 << 21))
    assert np.array_equal(bits(e1), bits(e3))
    # oracle spot check on a crop of primary hits at full resolution
    tri, t = sponza.primary_ids(cam, va.make_opts(seed=1), 5)
    o, d = O.primary_rays(cam, va.make_opts(seed=1), 5)
    sel = np.arange(0, 1920 * 1080, 97)
    osc = O.OracleScene(*scenes.sponza260k())
    rtri, rt = osc.trace(o[sel], d[sel])
    assert np.array_equal(tri[sel], rtri) and np.array_equal(bits(t[sel]), bits(rt))


@pytest.mark.parametrize("world,stripe", [(2, 16), (3, 7), (8, 16), (5, 64)])
def test_sharded_frames_assemble_to_the_single_gpu_frame(sponza, world, stripe):
    W, H = 320, 200
    cam = sponza_cam(W, H, 16)
    full, sf = sponza.render(cam, va.make_opts(seed=6))
    parts, rays = [], 0
    mrows = vdist.max_local_rows(H, stripe, world)
    for r in range(world):
        img, st = sponza.render(cam, va.make_opts(seed=6, rank=r, world=world, stripe_rows=stripe))
        assert img.shape[0] == va.local_rows(H, stripe, r, world)
        rays += st["rays_primary"] + st["rays_secondary"]
        pad = np.zeros((mrows, W, 5), np.float32)
        pad[:img.shape[0]] = img
        parts.append(pad)
    assert np.array_equal(bits(vdist.assemble_host(parts, W, H, stripe, world)), bits(full))
    assert rays == sf["rays_primary"] + sf["rays_secondary"]


def test_device_assemble_kernel(sponza):
    import torch
    W, H, stripe, world = 200, 131, 16, 4
    cam = sponza_cam(W, H, 8)
    full, _ = sponza.render(cam, va.make_opts(seed=3))
    mrows = vdist.max_local_rows(H, stripe, world)
    stride = mrows * W * 5
    big = torch.zeros(world * stride, device="cuda")
    for r in range(world):
        rows = va.local_rows(H, stripe, r, world)
        out = torch.empty((rows, W, 5), device="cuda")
        sponza.render_device(cam, va.make_opts(seed=3, rank=r, world=world, stripe_rows=stripe), out.data_ptr())
        big[r * stride:r * stride + out.numel()] = out.reshape(-1)
    import ctypes as C
    frame = torch.empty((H, W, 5), device="cuda")
    va._lib.check(va._lib.lib().vmx_assemble_device(C.c_void_p(big.data_ptr()), stride, W, H, stripe, world,
                                                    C.c_void_p(frame.data_ptr()), 0, None))
    assert np.array_equal(bits(frame.cpu().numpy()), bits(full))


def test_early_stop_never_takes_more_samples_and_keeps_lit_pixels_lit(sponza):
    """The reference's early stop is a biased estimator (pixels whose first samples are
    black stop early), so only one-sided properties hold against the fixed-spp frame."""
    cam = sponza_cam(480, 270, 64)
    opts = dict(seed=2, sampling=va.VMX_SAMPLING_CORRECTED)
    a, sa = sponza.render(cam, va.make_opts(early_stop=False, **opts))
    b, sb = sponza.render(cam, va.make_opts(early_stop=True, **opts))
    assert sb["samples"] < sa["samples"] == 480 * 270 * 64
    assert np.all(b[:, :, 4] <= 64) and np.all(b[:, :, 4] >= 9)  # n > sqrt(64) before any stop
    assert b[:, :, :3].mean() > 0.01 and a[:, :, :3].mean() > 0.01


def test_4k_frame_config5_geometry(sponza):
    """BASELINE config 5 image size (3840x2160) on one GPU at low spp: sharding and pipeline forms agree"""
    cam = sponza_cam(3840, 2160, 8)
    full, sf = sponza.render(cam, va.make_opts(seed=4, early_stop=False))
    assert full.shape == (2160, 3840, 5) and sf["samples"] == 3840 * 2160 * 8
    assert np.all(full[:, :, 4] == 8.0) and np.all(np.isfinite(full))
    # rank 5 of 8 renders exactly its stripes of the same frame
    part, sp = sponza.render(cam, va.make_opts(seed=4, early_stop=False, rank=5, world=8, stripe_rows=16))
    rows = va.local_row_indices(2160, 16, 5, 8)
    assert np.array_equal(bits(part), bits(full[rows]))
    # primary-hit map of the full-size frame: exact vs the oracle on a sparse subset
    tri, t = sponza.primary_ids(cam, va.make_opts(seed=4), 3)
    o, d = O.primary_rays(cam, va.make_opts(seed=4), 3)
    sel = np.arange(0, 3840 * 2160, 1013)
    rtri, rt = O.OracleScene(*scenes.sponza260k()).trace(o[sel], d[sel])
    assert np.array_equal(tri[sel], rtri) and np.array_equal(bits(t[sel]), bits(rt))


def test_config4_full_frame_256spp(sponza):
    """BASELINE config 4 at full size and full spp (the bench frame): pass split, stripe sharding
    and the early-stop bookkeeping at the size the oracle cannot reach"""
    W, H, spp = 1920, 1080, 256
    cam = sponza_cam(W, H, spp)
    a, sa = sponza.render(cam, va.make_opts(seed=1, early_stop=False))
    assert sa["samples"] == W * H * spp == sa["rays_primary"] and sa["passes"] == 1
    assert np.all(a[:, :, 4] == float(spp)) and np.all(np.isfinite(a))
    b, sb = sponza.render(cam, va.make_opts(seed=1, early_stop=False, max_paths=100 << 20))
    assert sb["passes"] > 1 and np.array_equal(bits(a), bits(b)) and sb["rays_secondary"] == sa["rays_secondary"]
    part, _ = sponza.render(cam, va.make_opts(seed=1, early_stop=False, rank=3, world=8, stripe_rows=16))
    assert np.array_equal(bits(part), bits(a[va.local_row_indices(H, 16, 3, 8)]))
    # early stop (reference behaviour): 17 samples before the rule can fire, then at least the first
    # sample of each of the 3 following strata; black pixels take exactly those 20
    e, se = sponza.render(cam, va.make_opts(seed=1, early_stop=True))
    n = e[:, :, 4]
    assert n.min() == 20.0 and n.max() <= spp and se["samples"] == int(n.astype(np.int64).sum())
    assert np.all(n[e[:, :, :3].sum(-1) == 0] == 20.0)
    e2, se2 = sponza.render(cam, va.make_opts(seed=1, early_stop=True, max_paths=8 << 20, pipeline=4))
    assert np.array_equal(bits(e), bits(e2)) and se2["samples"] == se["samples"]
    parts = [sponza.render(cam, va.make_opts(seed=1, early_stop=True, rank=r, world=2, stripe_rows=16))[0] for r in range(2)]
    for r in range(2):
        assert np.array_equal(bits(parts[r]), bits(e[va.local_row_indices(H, 16, r, 2)]))


def _against_the_oracle_rows(sc, osc, cam, seed, rank, world, stripe, H):
    """whole full-size frames of the HIP path against the oracle on the pixel subset the oracle can do in seconds: the
    rows of rank `rank` of `world` (stripes of `stripe` rows spread over the whole image height)"""
    rows = va.local_row_indices(H, stripe, rank, world)
    ELIDE = va.VMX_SAMPLING_PARITY | va.VMX_SAMPLING_ELIDE_DEAD
    sub = dict(rank=rank, world=world, stripe_rows=stripe)
    # fixed count: the bench's own frame (default: the traversal kernels sort their finished rays), the frame with every
    # Radiance step shaded in full (bench.py's headline form), the two-phase form, and VMX_SAMPLING_ELIDE_DEAD
    ref, ost = osc.render(cam, va.make_opts(seed=seed, early_stop=False, **sub))
    assert ref.shape[0] == len(rows)
    for kw in (dict(), dict(pipeline=0x100), dict(pipeline=0x200), dict(sampling=ELIDE)):
        img, st = sc.render(cam, va.make_opts(seed=seed, early_stop=False, **kw))
        assert np.array_equal(bits(img[rows]), bits(ref)), kw
        # the same rows rendered as a rank of their own: identical again, and the counters are the oracle's
        part, sp = sc.render(cam, va.make_opts(seed=seed, early_stop=False, **sub, **kw))
        assert np.array_equal(bits(part), bits(ref)), kw
        assert sp["samples"] == ost["samples"]
        if "sampling" not in kw:  # (under ELIDE_DEAD only the traced rays are counted)
            assert sp["rays_primary"] == ost["rays_primary"] and sp["rays_secondary"] == ost["rays_secondary"], kw
            assert sp["primary"]["tri_hits"] + sp["bounce"]["tri_hits"] == ost["primary"]["tri_hits"], kw
            assert sp["primary"]["continued"] + sp["bounce"]["continued"] == ost["primary"]["continued"], kw
    # early stop on (what the adapter renders), default and elided
    ref, ost = osc.render(cam, va.make_opts(seed=seed, early_stop=True, **sub))
    for kw in (dict(), dict(sampling=ELIDE)):
        img, st = sc.render(cam, va.make_opts(seed=seed, early_stop=True, **kw))
        assert np.array_equal(bits(img[rows]), bits(ref)), kw
        part, sp = sc.render(cam, va.make_opts(seed=seed, early_stop=True, **sub, **kw))
        assert np.array_equal(bits(part), bits(ref)) and sp["samples"] == ost["samples"] == int(ref[:, :, 4].astype(np.int64).sum())
        if "sampling" not in kw:
            assert sp["rays_primary"] == ost["rays_primary"] + sp["samples_discarded"]


def test_config4_bench_frame_against_the_oracle(sponza):
    """BASELINE config 4 at full size and full spp, seed 1 — the frame bench.py times — put against the oracle itself
    (pathtracer.cpp:200-328 restated) on one sixteenth of its rows: the default form, the headline form, the two-phase
    form, VMX_SAMPLING_ELIDE_DEAD and the early-stop frames, bit for bit, with ray / sample / hit / continuation
    counters (the one link that was self-comparison only)"""
    W, H, spp = 1920, 1080, 256
    osc = O.OracleScene(*scenes.sponza260k())
    _against_the_oracle_rows(sponza, osc, sponza_cam(W, H, spp), 1, 5, 16, 4, H)


def test_config3_bunny_frame_against_the_oracle():
    """BASELINE config 3 (bunny stand-in, 1024 x 1024 x 128 spp) at full size against the oracle on a sixteenth of its rows"""
    sc = va.Scene(*scenes.bunny70k())
    osc = O.OracleScene(*scenes.bunny70k())
    c = scenes.SCENES["bunny70k"][1]()
    cam = va.make_camera(c["position"], c["rotation_deg"], 1024, 1024, 128)
    _against_the_oracle_rows(sc, osc, cam, 9, 11, 16, 4, 1024)
    sc.close()


def test_config3_bunny_1024_128spp():
    """BASELINE config 3 at full size: split wavefront and fused kernel agree (the first-generation kernels:
    test_first_generation_kernels_give_the_same_frames)"""
    sc = va.Scene(*scenes.bunny70k())
    c = scenes.SCENES["bunny70k"][1]()
    cam = va.make_camera(c["position"], c["rotation_deg"], 1024, 1024, 128)
    a, sa = sc.render(cam, va.make_opts(seed=9, early_stop=False))
    assert sa["samples"] == 1024 * 1024 * 128 and np.all(np.isfinite(a))
    for kw in (dict(pipeline=1), dict(pipeline=4, tail_threshold=1 << 20)):
        b, sb = sc.render(cam, va.make_opts(seed=9, early_stop=False, **kw))
        assert np.array_equal(bits(a), bits(b)), kw
        assert sb["rays_secondary"] == sa["rays_secondary"]
    sc.close()


def test_config5_stripes_of_the_4k_1024spp_frame(sponza):
    """BASELINE config 5 (3840x2160, 1024 spp, 8 ranks): what one rank renders equals its rows of the
    whole frame rendered on one GPU"""
    W, H, spp = 3840, 2160, 1024
    cam = sponza_cam(W, H, spp)
    full, sf = sponza.render(cam, va.make_opts(seed=8, early_stop=False))
    assert sf["samples"] == W * H * spp and sf["passes"] > 1
    for r in (0, 6):
        part, sp = sponza.render(cam, va.make_opts(seed=8, early_stop=False, rank=r, world=8, stripe_rows=16))
        assert np.array_equal(bits(part), bits(full[va.local_row_indices(H, 16, r, 8)]))
        assert sp["samples"] == part.shape[0] * W * spp


def test_config5_rank_against_the_oracle(sponza):
    """BASELINE config 5 (3840x2160, 1024 spp, sharded): what one rank of 128 renders — 4-row stripes spread over the
    whole image height, 66.8 M samples — against the oracle's same rows (pathtracer.cpp:200-328 restated), bit for bit,
    fixed count and early stop, default and headline form, with the ray / sample counters"""
    W, H, spp = 3840, 2160, 1024
    cam = sponza_cam(W, H, spp)
    osc = O.OracleScene(*scenes.sponza260k())
    sub = dict(rank=37, world=128, stripe_rows=4)
    rows = va.local_row_indices(H, 4, 37, 128)
    ref, ost = osc.render(cam, va.make_opts(seed=8, early_stop=False, **sub))
    assert ref.shape == (len(rows), W, 5) and ost["samples"] == len(rows) * W * spp
    for kw in (dict(), dict(pipeline=0x100), dict(sampling=va.VMX_SAMPLING_PARITY | va.VMX_SAMPLING_ELIDE_DEAD)):
        part, sp = sponza.render(cam, va.make_opts(seed=8, early_stop=False, **sub, **kw))
        assert np.array_equal(bits(part), bits(ref)), kw
        if "sampling" not in kw:
            assert sp["rays_primary"] == ost["rays_primary"] and sp["rays_secondary"] == ost["rays_secondary"], kw
    eref, eost = osc.render(cam, va.make_opts(seed=8, early_stop=True, **sub))
    epart, esp = sponza.render(cam, va.make_opts(seed=8, early_stop=True, **sub))
    assert np.array_equal(bits(epart), bits(eref)) and esp["samples"] == eost["samples"]


def test_per_kernel_timings_of_the_last_render(sponza):
    """vmx_scene_timings: what bench.py's roofline object takes its launch durations from"""
    cam = sponza_cam(480, 270, 64)
    _, st = sponza.render(cam, va.make_opts(seed=2, early_stop=False, pipeline=4, tail_threshold=1))
    t = sponza.timings()
    for k in ("raygen", "trace_camera", "shade_camera", "trace_bounce", "shade_bounce", "resolve"):
        assert t[k]["launches"] >= 1 and t[k]["ms"] > 0 and t[k]["longest_ms"] <= t[k]["ms"] + 1e-9, k
    assert t["tail"]["launches"] == 0 and t["fused"]["launches"] == 0 and t["bruteforce"]["launches"] == 0
    assert abs(t["trace_camera"]["ms"] - st["primary"]["ms"]) < 1e-6
    assert abs(t["trace_bounce"]["ms"] + t["tail"]["ms"] - st["bounce"]["ms"]) < 1e-6
    _, st2 = sponza.render(cam, va.make_opts(seed=2, early_stop=False, pipeline=1))
    t2 = sponza.timings()
    assert t2["fused"]["launches"] >= 1 and t2["trace_camera"]["launches"] == 0
    _, _ = sponza.render_bruteforce(sponza_cam(96, 64, 8), va.make_opts(seed=2))
    assert sponza.timings()["bruteforce"]["launches"] == 1


def test_pass_size_follows_the_memory_budget(sponza, monkeypatch):
    """render_impl sizes a pass from hipMemGetInfo (free memory + what the scene's workspace
    already holds) instead of assuming 67 GB of path state fit; VMX_MEM_BUDGET_MB simulates a small device.
    Frames do not depend on the pass size."""
    cam = sponza_cam(640, 360, 16)
    for es in (False, True):
        opts = va.make_opts(seed=3, early_stop=es)
        monkeypatch.delenv("VMX_MEM_BUDGET_MB", raising=False)
        a, sa = sponza.render(cam, opts)
        monkeypatch.setenv("VMX_MEM_BUDGET_MB", "200")  # room for ~1.5 samples per pixel of path state
        b, sb = sponza.render(cam, opts)
        assert np.array_equal(bits(a), bits(b))
        assert sb["passes"] > sa["passes"] and sb["samples"] == sa["samples"]
        if not es:
            assert sa["passes"] == 1 and sb["passes"] >= 4
        monkeypatch.setenv("VMX_MEM_BUDGET_MB", "1")  # below the fixed part: one sample per pixel and pass still renders
        c, sc_ = sponza.render(cam, opts)
        assert np.array_equal(bits(a), bits(c))
    monkeypatch.delenv("VMX_MEM_BUDGET_MB", raising=False)


AB_LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "build", "libvermilion_hip_ab.so")


@pytest.mark.skipif(not os.path.exists(AB_LIB), reason="build/libvermilion_hip_ab.so not built (make -C vermilion_amd/csrc ab)")
def test_first_generation_kernels_give_the_same_frames(sponza):
    """The product library holds one generation of render kernels; the first one (k_primary / k_bounce, pipeline forms
    2 and 3) lives on in the A/B library `make ab` builds from the same sources + vmx_kernels_ab.inc.  Loaded next to
    the product here, it must give the product's frames bit for bit — and the product must refuse the forms it lacks."""
    ab = va._lib.load(AB_LIB)
    pos, nrm, uv = scenes.sponza260k()
    cam = sponza_cam(1920, 1080, 16)
    with va.Scene(pos, nrm, uv, lib=ab) as old:
        for es in (False, True):
            a, sa = sponza.render(cam, va.make_opts(seed=1, early_stop=es))
            # ... and the bounce-reordering experiment (path_sort.hip) that shares the A/B library
            for kw in (dict(pipeline=2), dict(pipeline=3), dict(), dict(reorder=0x35, tail_threshold=1),
                       dict(reorder=0x108044, tail_threshold=4096, lds_entries=3)):
                b, sb = old.render(cam, va.make_opts(seed=1, early_stop=es, **kw))
                assert np.array_equal(bits(a), bits(b)), (es, kw)
                assert sb["samples"] == sa["samples"]
                if not es:
                    assert sb["rays_secondary"] == sa["rays_secondary"]
        c = scenes.SCENES["sponza260k"][1]()
        o, d = O.primary_rays(va.make_camera(c["position"], c["rotation_deg"], 256, 128, 16), va.make_opts(seed=5), 1)
        for sampling in (0, 1, 0x100):
            r0, _ = sponza.radiance(o, d, va.make_opts(seed=5, sampling=sampling))
            r2, _ = old.radiance(o, d, va.make_opts(seed=5, sampling=sampling, pipeline=2))
            assert np.array_equal(bits(r0), bits(r2))
    with pytest.raises(va.VmxError, match="A/B library"):
        sponza.render(cam, va.make_opts(seed=1, pipeline=2))
    with pytest.raises(va.VmxError, match="A/B library"):
        sponza.render(cam, va.make_opts(seed=1, early_stop=False, reorder=0x35, tail_threshold=1))


@pytest.mark.skipif(not os.path.exists(AB_LIB), reason="build/libvermilion_hip_ab.so not built (make -C vermilion_amd/csrc ab)")
def test_phase_pure_pool_probe_gives_the_same_frames(sponza):
    """VERDICT r3 item 1's probe (vmx_trace_pool.inc, A/B library only; vmx_opts.reserved[0] bit 10): a bounce traversal
    whose waves hold one phase, ray state in LDS by slot.  Per-ray test order is the reference's (bvh.cpp:47-145), so
    frames and ray counts equal the product's bit for bit — with every bounce generation through it (tail_threshold=1),
    with deep stacks in LDS and with most levels in the HBM slab, and with a pool barely larger than the block."""
    ab = va._lib.load(AB_LIB)
    pos, nrm, uv = scenes.sponza260k()
    cam = sponza_cam(960, 540, 64)
    a, sa = sponza.render(cam, va.make_opts(seed=2, early_stop=False, pipeline=4 | 0x100, tail_threshold=1))
    with va.Scene(pos, nrm, uv, lib=ab) as old:
        for env in (dict(), dict(VMX_AB_POOL_SLOTS="320", VMX_AB_POOL_LEVELS="3"), dict(VMX_AB_POOL_SLOTS="704", VMX_AB_POOL_LEVELS="12")):
            for k in ("VMX_AB_POOL_SLOTS", "VMX_AB_POOL_LEVELS"):
                os.environ.pop(k, None)
            os.environ.update(env)
            try:
                b, sb = old.render(cam, va.make_opts(seed=2, early_stop=False, pipeline=4 | 0x100 | 0x400, tail_threshold=1))
            finally:
                for k in env:
                    os.environ.pop(k, None)
            assert np.array_equal(bits(a), bits(b)), env
            assert sb["rays_secondary"] == sa["rays_secondary"] and sb["bounce"]["tri_hits"] == sa["bounce"]["tri_hits"]
    with pytest.raises(va.VmxError, match="A/B library"):
        sponza.render(cam, va.make_opts(seed=2, early_stop=False, pipeline=0x100 | 0x400))


def test_two_phase_shading_equals_one_phase(sponza):
    """Split passes settle the Radiance steps that end by their draws alone apart from the others: by default in the
    traversal kernel itself, which hands the other rays on as records (k_trace_w<.., SORT>; reference sampling only);
    with vmx_opts.reserved[0] bit 9 in a first shading phase (k_shade_ends, the form `corrected` sampling uses);
    bit 8 asks for plain one-phase shading, and a call with collect_counters always uses that.  Frames, ray counts, triangle-hit counts and continuation counts must agree —
    on the bench scene at 1080p, with bounce generations through both forms (tail_threshold=1: no fused tail), with a
    texture, and with a sphere table whose lights come after other spheres (SceneDev::emit_prefix)."""
    cam = sponza_cam(1920, 1080, 16)
    for es in (False, True):
        for kw in (dict(pipeline=4), dict(pipeline=4, tail_threshold=1)):
            a, sa = sponza.render(cam, va.make_opts(seed=3, early_stop=es, **kw))
            b, sb = sponza.render(cam, va.make_opts(seed=3, early_stop=es, **dict(kw, pipeline=4 | 0x100)))
            c, sc_ = sponza.render(cam, va.make_opts(seed=3, early_stop=es, collect_counters=True, **kw))
            d, sd = sponza.render(cam, va.make_opts(seed=3, early_stop=es, **dict(kw, pipeline=4 | 0x200)))
            assert np.array_equal(bits(a), bits(b)) and np.array_equal(bits(a), bits(c)) and np.array_equal(bits(a), bits(d)), (es, kw)
            for other in (sb, sc_, sd):
                for key in ("samples", "samples_discarded", "rays_primary", "rays_secondary"):
                    assert sa[key] == other[key], (es, kw, key)
                for stage in ("primary", "bounce"):
                    assert sa[stage]["tri_hits"] == other[stage]["tri_hits"], (es, kw, stage)
    # lights late in the table, a weak one among them, a texture
    pos, nrm, uv = scenes.bunny70k()
    table = va.spheres_array([
        dict(centre=(0, -5e7, 0), radius=5e7), dict(centre=(0, 5e7 + 1000, 0), radius=5e7),
        dict(centre=(0, 700, 300), radius=220, colour=(1.5, 1.2, 0.9), emit=True),
        dict(centre=(-5e7 + 2000, 0, 0), radius=5e7, normal_sign=-1),
        dict(centre=(300, 400, 300), radius=90, colour=(0.3, 0.2, 0.1), emit=True),
        dict(centre=(5e7 - 2000, 0, 0), radius=5e7, normal_sign=-1),
        dict(centre=(0, 0, -5e7 + 2000), radius=5e7, normal_sign=-1), dict(centre=(0, 0, 5e7 - 2000), radius=5e7)])
    # ... and a weak light that encloses everything: every ray has a light in reach, so the traversal kernel hands
    # every one of them on (the record list at its fullest)
    glow = va.spheres_array([
        dict(centre=(0, 300, 0), radius=6000, colour=(0.2, 0.25, 0.3), emit=True, normal_sign=-1),
        dict(centre=(0, 700, 300), radius=220, colour=(1.5, 1.2, 0.9), emit=True)])
    c = scenes.bunny_camera()
    cam = va.make_camera(c["position"], c["rotation_deg"], 256, 192, 32, back_size=(3.6, 2.7))
    tex = (np.random.RandomState(4).random_sample((16, 16, 3)) * 1.2).astype(np.float32)
    with va.Scene(pos, nrm, uv, spheres=glow) as g:
        osc = O.OracleScene(pos, nrm, uv, spheres=glow)
        for sampling in (0, va.VMX_SAMPLING_ELIDE_DEAD):
            ref, rst = osc.render(cam, va.make_opts(seed=8, early_stop=False))
            img, st = g.render(cam, va.make_opts(seed=8, early_stop=False, sampling=sampling, pipeline=4, tail_threshold=1))
            assert np.array_equal(bits(img), bits(ref)), sampling
            if sampling == 0:
                assert st["rays_secondary"] == rst["rays_secondary"] and st["rays_primary"] == rst["rays_primary"]
            else:  # next to nothing to elide: a few bounce directions with NaN components miss even this light
                assert st["rays_primary"] == rst["rays_primary"] and 0.99 * rst["rays_secondary"] < st["rays_secondary"] <= rst["rays_secondary"]
        osc.close()
    with va.Scene(pos, nrm, uv, spheres=table) as g:
        osc = O.OracleScene(pos, nrm, uv, spheres=table)
        for textured in (False, True):
            if textured:
                g.bind_texture(tex)
                osc.bind_texture(tex)
            for sampling in (0, 1):
                ref, rst = osc.render(cam, va.make_opts(seed=8, early_stop=False, sampling=sampling))
                for pipe in (4, 4 | 0x100, 4 | 0x200):
                    for tail in (0, 1):
                        img, st = g.render(cam, va.make_opts(seed=8, early_stop=False, sampling=sampling, pipeline=pipe, tail_threshold=tail))
                        assert np.array_equal(bits(img), bits(ref)), (textured, sampling, pipe, tail)
                        assert st["rays_secondary"] == rst["rays_secondary"] and st["rays_primary"] == rst["rays_primary"]
        osc.close()


def test_frames_do_not_depend_on_the_frames_before_them(sponza):
    """One scene, frames of different kinds in a row (fixed count / early stop, with and without
    VMX_SAMPLING_ELIDE_DEAD, a counted frame in between): every frame equals the first one of its kind and reports the
    same counts.  (Round 3: a fixed-count frame read FrameDev::lead, which only the early-stop branch of the pass loop
    set — it rendered 16 samples per pixel whenever the stack still held the previous early-stop frame's 17.)"""
    cam = sponza_cam(640, 360, 64)
    E = va.VMX_SAMPLING_ELIDE_DEAD
    seq = [(True, E), (False, E), (False, 0), (True, 0), (False, 0), (True, E), (False, E), (False, E), (True, 0), (False, 0),
           (True, E), (False, 0), (False, E)]
    ref, cnt = {}, {}
    for i, (es, samp) in enumerate(seq):
        if i == 5:
            sponza.render(cam, va.make_opts(seed=2, early_stop=True, collect_counters=True, pipeline=4))
        img, st = sponza.render(cam, va.make_opts(seed=2, early_stop=es, sampling=samp, pipeline=4))
        key = (st["rays_primary"], st["rays_secondary"], st["samples"], st["passes"])
        assert np.array_equal(bits(ref.setdefault(es, img)), bits(img)), (i, es, samp)
        assert cnt.setdefault((es, samp), key) == key, (i, es, samp)
        if not es:
            assert st["samples"] == 640 * 360 * 64 and st["samples_discarded"] == 0


def test_threads_render_and_introspect_concurrently():
    """The ABI is blocking and a scene serialises its own calls (vmx_scene::mu); different scenes may be driven from
    different host threads at the same time, and vmx_scene_describe / _timings / _bvh may be called while another
    thread renders on the same scene (flat_ready is atomic, timings are copied under the lock)."""
    import threading
    pos, nrm, uv = scenes.bunny70k()
    c = scenes.bunny_camera()
    cam = va.make_camera(c["position"], c["rotation_deg"], 192, 128, 32, back_size=(3.6, 2.4))
    opts = va.make_opts(seed=6, early_stop=True)
    with va.Scene(pos, nrm, uv) as ref_sc:
        ref, _ = ref_sc.render(cam, opts)
    scs = [va.Scene(pos, nrm, uv, builder=b) for b in (va._lib.VMX_BVH_REFERENCE, va._lib.VMX_BVH_REFERENCE,
                                                       va._lib.VMX_BVH_LBVH, va._lib.VMX_BVH_PLOC)]
    out, errs, stop = {}, [], threading.Event()

    def render(i):
        try:
            for _ in range(6):
                out[i], _ = scs[i].render(cam, opts)
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    def poke():
        try:
            while not stop.is_set():
                for sc in scs:
                    d = sc.describe()
                    assert d["ntris"] == pos.shape[0] and d["n_nodes"] > 0
                    sc.timings()
                scs[2].bvh()
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=render, args=(i,)) for i in range(4)] + [threading.Thread(target=poke)]
    for t in th:
        t.start()
    for t in th[:4]:
        t.join()
    stop.set()
    th[4].join()
    assert not errs, errs
    assert np.array_equal(bits(out[0]), bits(ref)) and np.array_equal(bits(out[1]), bits(ref))
    # device-built trees: their own (deterministic) frames, the same every time
    for i in (2, 3):
        again, _ = scs[i].render(cam, opts)
        assert np.array_equal(bits(out[i]), bits(again))
    for sc in scs:
        sc.close()
