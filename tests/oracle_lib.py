"""ctypes binding of the CPU oracle (oracle/libvmx_oracle.so) for tests, smoke()
and bench.py's cpu_baseline leg.  Test infrastructure only — never imported by
the vermilion_amd package."""
import ctypes as C
import os
import subprocess

import numpy as np

from vermilion_amd import _lib as L
from vermilion_amd.scene import RAYHIT_DTYPE

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ODIR = os.path.join(_ROOT, "oracle")

ORC_RNG_XOSHIRO_KEYED = 0
ORC_RNG_MT19937_64 = 1


class TraceCounters(C.Structure):
    _fields_ = [("inner_visits", C.c_uint64), ("tri_tests", C.c_uint64), ("pops", C.c_uint64),
                ("max_stack", C.c_uint64)]


def build():
    subprocess.run(["make", "-C", _ODIR], check=True, stdout=subprocess.DEVNULL)


_libs = {}


def lib(fast=False):
    name = "libvmx_oracle_fast.so" if fast else "libvmx_oracle.so"
    if name in _libs:
        return _libs[name]
    path = os.path.join(_ODIR, name)
    if not os.path.exists(path):
        build()
    l = C.CDLL(path)
    P = C.c_void_p
    l.orc_build_flags.restype = C.c_char_p
    l.orc_default_spheres.restype = C.POINTER(L.Sphere)
    l.orc_default_spheres.argtypes = [C.POINTER(C.c_uint32)]
    l.orc_scene_create.restype = P
    l.orc_scene_create.argtypes = [P, P, P, C.c_uint32, P, C.c_uint32, C.c_uint32]
    l.orc_scene_destroy.argtypes = [P]
    l.orc_scene_create_from_tree.restype = P
    l.orc_scene_create_from_tree.argtypes = [P, P, P, C.c_uint32, P, C.c_uint32, C.c_uint32, P, P, P, P, P]
    l.orc_scene_bind_texture.argtypes = [P, P, C.c_uint32, C.c_uint32, C.c_uint32]
    l.orc_scene_describe.argtypes = [P] + [C.POINTER(C.c_uint32)] * 3
    l.orc_scene_bvh.argtypes = [P] * 6
    l.orc_trace.argtypes = [P, P, P, C.c_uint32, P, P, C.POINTER(TraceCounters)]
    l.orc_raycast.argtypes = [P, P, P, C.c_uint32, P]
    l.orc_radiance.argtypes = [P, P, P, C.c_uint32, C.POINTER(L.Opts), P, C.POINTER(L.Stats)]
    l.orc_radiance_mt.argtypes = [P, P, P, C.c_uint32, P, C.c_uint32, P]
    l.orc_audit_elision.argtypes = [P, P, P, C.c_uint32, C.POINTER(L.Opts), P, P]
    l.orc_camera_matrix.argtypes = [C.POINTER(L.CameraDesc), P]
    l.orc_primary_rays.argtypes = [C.POINTER(L.CameraDesc), C.POINTER(L.Opts), C.c_uint32, P, P]
    l.orc_stream.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, P]
    l.orc_splitmix64.restype = C.c_uint64
    l.orc_splitmix64.argtypes = [C.POINTER(C.c_uint64)]
    l.orc_render.argtypes = [P, C.POINTER(L.CameraDesc), C.POINTER(L.Opts), C.c_int, C.c_int, P,
                             C.POINTER(L.Stats)]
    l.orc_render_bruteforce.argtypes = [P, C.POINTER(L.CameraDesc), C.POINTER(L.Opts), C.c_uint32, C.c_int, P,
                                        C.POINTER(L.Stats)]
    l.orc_max_threads.restype = C.c_int
    l.orc_quantize.argtypes = [P, C.c_uint64, P, P]
    l.orc_trig.argtypes = [P, C.c_uint32, P, P]
    l.orc_trig_compare_libm.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    l.orc_check_div_by_count.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    _libs[name] = l
    return l


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32).reshape(-1, 3)


class OracleScene:
    def __init__(self, pos, nrm, uv=None, spheres=None, leaf_size=4, fast=False, tree=None):
        """tree: dict as returned by Scene.bvh() — traverse that flat tree instead of building one"""
        self.l = lib(fast)
        pos = np.ascontiguousarray(pos, np.float32).reshape(-1, 9)
        nrm = np.ascontiguousarray(nrm, np.float32).reshape(-1, 9)
        uvp = None
        if uv is not None:
            uv = np.ascontiguousarray(uv, np.float32).reshape(-1, 6)
            uvp = uv.ctypes.data
        sp, nsp = (None, 0) if spheres is None else (C.addressof(spheres), len(spheres))
        self.ntris = pos.shape[0]
        if tree is not None:
            t = {k: np.ascontiguousarray(v) for k, v in tree.items()}
            self.h = self.l.orc_scene_create_from_tree(pos.ctypes.data, nrm.ctypes.data, uvp, self.ntris, sp, nsp,
                                                       len(t["start"]), t["start"].ctypes.data, t["nprims"].ctypes.data,
                                                       t["right_offset"].ctypes.data, t["bbox"].ctypes.data,
                                                       t["prim_order"].ctypes.data)
        else:
            self.h = self.l.orc_scene_create(pos.ctypes.data, nrm.ctypes.data, uvp, self.ntris, sp, nsp, leaf_size)
        if not self.h:
            raise RuntimeError("orc_scene_create failed")

    def close(self):
        if self.h:
            self.l.orc_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def bind_texture(self, data):
        data = np.ascontiguousarray(data, np.float32)
        h, w = data.shape[0], data.shape[1]
        c = 1 if data.ndim == 2 else data.shape[2]
        if self.l.orc_scene_bind_texture(self.h, data.ctypes.data, w, h, c) != 0:
            raise ValueError("bad texture")

    def describe(self):
        a, b, c = C.c_uint32(), C.c_uint32(), C.c_uint32()
        self.l.orc_scene_describe(self.h, C.byref(a), C.byref(b), C.byref(c))
        return {"n_nodes": a.value, "n_leaves": b.value, "max_depth": c.value}

    def bvh(self):
        n = self.describe()["n_nodes"]
        start, nprims, roff = (np.zeros(n, np.uint32) for _ in range(3))
        bbox = np.zeros((n, 6), np.float32)
        order = np.zeros(self.ntris, np.uint32)
        self.l.orc_scene_bvh(self.h, start.ctypes.data, nprims.ctypes.data, roff.ctypes.data, bbox.ctypes.data,
                             order.ctypes.data)
        return {"start": start, "nprims": nprims, "right_offset": roff, "bbox": bbox, "prim_order": order}

    def trace(self, o, d, counters=False):
        o, d = _f32(o), _f32(d)
        n = o.shape[0]
        tri = np.empty(n, np.int32)
        t = np.empty(n, np.float32)
        c = TraceCounters()
        self.l.orc_trace(self.h, o.ctypes.data, d.ctypes.data, n, tri.ctypes.data, t.ctypes.data,
                         C.byref(c) if counters else None)
        if counters:
            return tri, t, {k: getattr(c, k) for k, _ in c._fields_}
        return tri, t

    def raycast(self, o, d):
        o, d = _f32(o), _f32(d)
        out = np.zeros(o.shape[0], dtype=RAYHIT_DTYPE)
        self.l.orc_raycast(self.h, o.ctypes.data, d.ctypes.data, o.shape[0], out.ctypes.data)
        return out

    def radiance(self, o, d, opts):
        o, d = _f32(o), _f32(d)
        out = np.empty((o.shape[0], 4), np.float32)
        st = L.Stats()
        self.l.orc_radiance(self.h, o.ctypes.data, d.ctypes.data, o.shape[0], C.byref(opts), out.ctypes.data,
                            C.byref(st))
        return out, st.as_dict()

    def audit_elision(self, o, d, opts):
        """ElisionAudit (oracle/vmx_oracle.cpp) over explicit camera rays: (radiance, dict of the six counters)"""
        o, d = _f32(o), _f32(d)
        out = np.empty((o.shape[0], 4), np.float32)
        c = np.zeros(6, np.uint64)
        self.l.orc_audit_elision(self.h, o.ctypes.data, d.ctypes.data, o.shape[0], C.byref(opts), out.ctypes.data, c.ctypes.data)
        keys = ("steps", "predicted_last", "predicted_dead", "not_last", "dead_changed", "colour_mismatch")
        return out, {k: int(v) for k, v in zip(keys, c)}

    def radiance_mt(self, o, d, seeds, sampling=0):
        o, d = _f32(o), _f32(d)
        seeds = np.ascontiguousarray(seeds, np.uint64)
        out = np.empty((o.shape[0], 4), np.float32)
        self.l.orc_radiance_mt(self.h, o.ctypes.data, d.ctypes.data, o.shape[0], seeds.ctypes.data, sampling,
                               out.ctypes.data)
        return out

    def render_bruteforce(self, cam, opts, flags=0, threads=0):
        """BruteForceTracer::Render restated (integrators.cpp:9-186), whole image [H, W, 5]"""
        W, H = cam.image_res[0], cam.image_res[1]
        out = np.empty((H, W, 5), np.float32)
        st = L.Stats()
        self.l.orc_render_bruteforce(self.h, C.byref(cam), C.byref(opts), int(flags), threads, out.ctypes.data,
                                     C.byref(st))
        return out, st.as_dict()

    def render(self, cam, opts, rng_mode=ORC_RNG_XOSHIRO_KEYED, threads=0):
        """PathTracer::Render restated.  opts.world > 1: only this rank's stripes (vmx_render's sharding), packed in
        ascending row order [local_rows, W, 5] — a pixel subset of the same frame (streams are keyed by the global pixel)"""
        W, H = cam.image_res[0], cam.image_res[1]
        if opts.world > 1:
            stripe = opts.stripe_rows or 16
            H = sum(1 for y in range(H) if (y // stripe) % opts.world == opts.rank)
        out = np.empty((H, W, 5), np.float32)
        st = L.Stats()
        self.l.orc_render(self.h, C.byref(cam), C.byref(opts), rng_mode, threads, out.ctypes.data, C.byref(st))
        return out, st.as_dict()


def primary_rays(cam, opts, k=0):
    n = cam.image_res[0] * cam.image_res[1]
    o = np.empty((n, 3), np.float32)
    d = np.empty((n, 3), np.float32)
    lib().orc_primary_rays(C.byref(cam), C.byref(opts), k, o.ctypes.data, d.ctypes.data)
    return o, d


def camera_matrix(cam):
    m = np.empty(9, np.float32)
    lib().orc_camera_matrix(C.byref(cam), m.ctypes.data)
    return m.reshape(3, 3)  # [col][row]


def stream(seed, pixel, k, n):
    out = np.empty(n, np.uint64)
    lib().orc_stream(seed, pixel, k, n, out.ctypes.data)
    return out


def max_threads():
    return lib().orc_max_threads()


def quantize(frame):
    frame = np.ascontiguousarray(frame, np.float32).reshape(-1, 5)
    n = frame.shape[0]
    rgba = np.empty((n, 4), np.uint8)
    depth = np.empty(n, np.float32)
    lib().orc_quantize(frame.ctypes.data, n, rgba.ctypes.data, depth.ctypes.data)
    return rgba, depth


def trig(x):
    """restated cosf / sinf of the default reading of pathtracer.cpp:162 (oracle: libm_sincosf)"""
    x = np.ascontiguousarray(x, np.float32)
    cs, sn = np.empty_like(x), np.empty_like(x)
    lib().orc_trig(x.ctypes.data, x.size, cs.ctypes.data, sn.ctypes.data)
    return cs, sn


def trig_compare_libm(lo_bits, hi_bits):
    """(cosf, sinf) counts of floats in the bit-pattern range where the restatement differs from the host libm"""
    dc, ds = C.c_uint64(0), C.c_uint64(0)
    lib().orc_trig_compare_libm(int(lo_bits), int(hi_bits), C.byref(dc), C.byref(ds))
    return dc.value, ds.value
