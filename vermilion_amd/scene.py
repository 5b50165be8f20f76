"""Thin object wrapper over the C ABI: one `Scene` = one `vmx_scene*`.

numpy arrays in, numpy arrays out; all compute happens in libvermilion_hip.so.
"""
import ctypes as C

import numpy as np

from . import _lib as L


def _f32(a, shape_last=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if shape_last is not None and (a.ndim == 0 or a.shape[-1] != shape_last):
        a = a.reshape(-1, shape_last)
    return a


def make_camera(position, rotation_deg, width, height, spp, back_distance=6.0, back_size=(3.6, 2.4), rotation_rad=None):
    """cameraSettings subset (core/camera/camera.h:32-47); defaults follow
    RenderEngine::CreateInternalDefaultCamera (core/engines/renderEngine.cpp:135-139).
    rotation_rad: Camera::mRotation itself (radians, x and y already negated, camera.cpp:43-47) instead of the
    settings' degrees — what an Integrator holds (VMX_ROTATION_RADIANS)."""
    c = L.CameraDesc()
    c.position[:] = [float(v) for v in position]
    if rotation_rad is not None:
        c.rotation_units = L.VMX_ROTATION_RADIANS
        c.rotation_rad[:] = [float(v) for v in rotation_rad]
    else:
        c.rotation_deg[:] = [float(v) for v in rotation_deg]
    c.back_distance = float(back_distance)
    c.back_size[:] = [float(back_size[0]), float(back_size[1])]
    c.image_res[:] = [int(width), int(height)]
    c.rays_per_pixel = int(spp)
    return c


def make_opts(seed=1, early_stop=True, sampling=L.VMX_SAMPLING_PARITY, rank=0, world=1, stripe_rows=16,
              samples_per_batch=0, collect_counters=False, pipeline=0, max_paths=0, tail_threshold=0,
              refill_min=0, shade_min=0, reorder=0, lds_entries=0):
    o = L.Opts()
    o.seed = int(seed)
    o.early_stop = 1 if early_stop else 0
    o.sampling = int(sampling)
    o.rank, o.world, o.stripe_rows = int(rank), int(world), int(stripe_rows)
    o.samples_per_batch = int(samples_per_batch)
    o.collect_counters = 1 if collect_counters else 0
    o.reserved[0] = int(pipeline)      # 0 default routing, 1 fused kernel for every pass, 4 split wavefront for every pass; 2/3 first-generation kernels (A/B library only); | 0x100 one-phase shading, | 0x200 two-phase shading (k_shade_ends) instead of sorted rays
    o.reserved[1] = int(max_paths)     # paths in flight per pass (0 -> 16M)
    o.reserved[2] = int(tail_threshold)
    o.reserved[3] = int(refill_min)    # k_paths: refill when this many lanes idle (0 -> 16)
    o.reserved[4] = int(shade_min)     # k_paths: shade when this many lanes finished (0 -> 16)
    o.reserved[5] = int(reorder)       # bounce reordering key (vmx_api.cpp: Tuning::sort_mode); 0 = the library's default
    o.reserved[6] = int(lds_entries)   # stack levels kept in LDS, all kernels (0 -> 8 camera / 13 bounce / 10 fused); deeper ones spill to HBM
    return o


def spheres_array(spheres):
    """list of dicts/tuples -> ctypes array of vmx_sphere"""
    arr = (L.Sphere * len(spheres))()
    for i, s in enumerate(spheres):
        arr[i].centre[:] = [float(v) for v in s["centre"]]
        arr[i].radius = float(s["radius"])
        arr[i].colour[:] = [float(v) for v in s.get("colour", (0, 0, 0))]
        arr[i].flags = L.VMX_SPHERE_EMIT if s.get("emit", False) else 0
        arr[i].normal_centre[:] = [float(v) for v in s.get("normal_centre", s["centre"])]
        arr[i].normal_sign = float(s.get("normal_sign", 1.0))
    return arr


def default_spheres():
    n = C.c_uint32(0)
    p = L.lib().vmx_default_spheres(C.byref(n))
    out = (L.Sphere * n.value)()
    for i in range(n.value):
        C.memmove(C.byref(out[i]), C.byref(p[i]), C.sizeof(L.Sphere))
    return out


class Scene:
    """Device-resident scene: replaces MeshEngine::createBVH + BVH for the HIP path."""

    def __init__(self, pos, nrm, uv=None, spheres=None, leaf_size=4, device=0, builder=L.VMX_BVH_REFERENCE, lib=None):
        """lib: a library object from _lib.load(path) — the tests' A/B library; default: the product library"""
        self._lib = lib if lib is not None else L.lib()
        pos = _f32(pos).reshape(-1, 9)
        nrm = _f32(nrm).reshape(-1, 9)
        if pos.shape != nrm.shape:
            raise ValueError("pos and nrm must both be [ntris, 9]")
        uvp = None
        if uv is not None:
            uv = _f32(uv).reshape(-1, 6)
            uvp = uv.ctypes.data
        self._spheres = spheres
        sp, nsp = (None, 0) if spheres is None else (C.addressof(spheres), len(spheres))
        h = C.c_void_p()
        self._check(self._lib.vmx_scene_create_ex(pos.ctypes.data, nrm.ctypes.data, uvp, pos.shape[0], sp, nsp,
                                            int(leaf_size), int(builder), int(device), C.byref(h)))
        self._h = h
        self.ntris = pos.shape[0]
        self.device = int(device)

    def _check(self, code):
        if code != L.VMX_OK:
            raise L.VmxError(code, self._lib.vmx_last_error().decode("utf-8", "replace"))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.vmx_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def bind_texture(self, data):
        """MeshEngine::bindTexture (meshEngine.cpp:74-93): float image [H, W] or [H, W, C], C <= 4.
        Only the first bound texture is sampled (pathtracer.cpp:65)."""
        data = np.ascontiguousarray(data, dtype=np.float32)
        if data.ndim not in (2, 3):
            raise ValueError("texture must be [H, W] or [H, W, C]")
        c = 1 if data.ndim == 2 else data.shape[2]
        self._check(self._lib.vmx_scene_bind_texture(self._h, data.ctypes.data, data.shape[1], data.shape[0], c))
        return True

    # -- introspection ----------------------------------------------------
    def describe(self):
        d = L.SceneDesc()
        self._check(self._lib.vmx_scene_describe(self._h, C.byref(d)))
        return {k: getattr(d, k) for k, _ in d._fields_ if k != "pad"}

    def timings(self):
        """per-kernel device time of the last render on this scene (vmx_timings)"""
        t = L.Timings()
        self._check(self._lib.vmx_scene_timings(self._h, C.byref(t)))
        return t.as_dict()

    def bvh(self):
        n = self.describe()["n_nodes"]
        start = np.zeros(n, np.uint32)
        nprims = np.zeros(n, np.uint32)
        roff = np.zeros(n, np.uint32)
        bbox = np.zeros((n, 6), np.float32)
        order = np.zeros(self.ntris, np.uint32)
        self._check(self._lib.vmx_scene_bvh(self._h, start.ctypes.data, nprims.ctypes.data, roff.ctypes.data,
                                      bbox.ctypes.data, order.ctypes.data))
        return {"start": start, "nprims": nprims, "right_offset": roff, "bbox": bbox, "prim_order": order}

    # -- parity hooks -------------------------------------------------------
    def trace(self, origin, direction):
        o, d = _f32(origin, 3), _f32(direction, 3)
        n = o.shape[0]
        tri = np.empty(n, np.int32)
        t = np.empty(n, np.float32)
        self._check(self._lib.vmx_trace(self._h, o.ctypes.data, d.ctypes.data, n, tri.ctypes.data, t.ctypes.data))
        return tri, t

    def raycast(self, origin, direction):
        o, d = _f32(origin, 3), _f32(direction, 3)
        n = o.shape[0]
        out = np.zeros(n, dtype=RAYHIT_DTYPE)
        self._check(self._lib.vmx_raycast(self._h, o.ctypes.data, d.ctypes.data, n, out.ctypes.data))
        return out

    def primary_ids(self, cam, opts, k=0):
        n = cam.image_res[0] * cam.image_res[1]
        tri = np.empty(n, np.int32)
        t = np.empty(n, np.float32)
        self._check(self._lib.vmx_primary_ids(self._h, C.byref(cam), C.byref(opts), int(k), tri.ctypes.data,
                                        t.ctypes.data))
        return tri, t

    def radiance(self, origin, direction, opts):
        o, d = _f32(origin, 3), _f32(direction, 3)
        n = o.shape[0]
        out = np.empty((n, 4), np.float32)
        st = L.Stats()
        self._check(self._lib.vmx_radiance(self._h, o.ctypes.data, d.ctypes.data, n, C.byref(opts), out.ctypes.data,
                                     C.byref(st)))
        return out, st.as_dict()

    # -- render ---------------------------------------------------------------
    def render(self, cam, opts):
        """PathTracer::Render into a host array [local_rows, W, 5] (RGBAZ)."""
        rows = local_rows(cam.image_res[1], opts.stripe_rows, opts.rank, opts.world)
        out = np.empty((rows, cam.image_res[0], 5), np.float32)
        st = L.Stats()
        self._check(self._lib.vmx_render(self._h, C.byref(cam), C.byref(opts), out.ctypes.data, C.byref(st)))
        return out, st.as_dict()

    def render_bruteforce(self, cam, opts, flags=0):
        """BruteForceTracer::Render (core/integrators/integrators.cpp:9-186) into a host array
        [local_rows, W, 5]: r, g, b, alpha = hit fraction, depth = last sample's hit distance."""
        rows = local_rows(cam.image_res[1], opts.stripe_rows, opts.rank, opts.world)
        out = np.empty((rows, cam.image_res[0], 5), np.float32)
        st = L.Stats()
        self._check(self._lib.vmx_render_bruteforce(self._h, C.byref(cam), C.byref(opts), int(flags), out.ctypes.data,
                                              C.byref(st)))
        return out, st.as_dict()

    def render_device(self, cam, opts, d_out_ptr, stream_ptr=None):
        """Same, into device memory (e.g. a torch tensor's data_ptr()) on `stream_ptr`."""
        st = L.Stats()
        self._check(self._lib.vmx_render_device(self._h, C.byref(cam), C.byref(opts), C.c_void_p(d_out_ptr),
                                          C.c_void_p(stream_ptr or 0), C.byref(st)))
        return st.as_dict()


class MultiScene:
    """One process, several devices (vmx_multi_*): scene replicas render interleaved stripes, the packed
    stripes are gathered device-to-device on devices[0] and de-interleaved there."""

    def __init__(self, pos, nrm, uv=None, devices=(0,), spheres=None, leaf_size=4, builder=L.VMX_BVH_REFERENCE):
        pos = _f32(pos).reshape(-1, 9)
        nrm = _f32(nrm).reshape(-1, 9)
        uvp = None
        if uv is not None:
            uv = _f32(uv).reshape(-1, 6)
            uvp = uv.ctypes.data
        self._spheres = spheres
        sp, nsp = (None, 0) if spheres is None else (C.addressof(spheres), len(spheres))
        devs = (C.c_int * len(devices))(*[int(d) for d in devices])
        h = C.c_void_p()
        L.check(L.lib().vmx_multi_create(pos.ctypes.data, nrm.ctypes.data, uvp, pos.shape[0], sp, nsp, int(leaf_size),
                                         int(builder), devs, len(devices), C.byref(h)))
        self._h = h
        self.ntris = pos.shape[0]
        self.world = L.lib().vmx_multi_world(h)

    def close(self):
        if getattr(self, "_h", None):
            L.lib().vmx_multi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def routes(self):
        """[(device, route)] per replica: route 2 = the root's own device, 1 = direct peer copy, 0 = staged through the host"""
        d, r = (C.c_int * self.world)(), (C.c_int * self.world)()
        L.check(L.lib().vmx_multi_routes(self._h, d, r))
        return list(zip(list(d), list(r)))

    def timings(self):
        """the exchange step of the last render, timed apart from the rendering (vmx_multi_timings)"""
        t = L.MultiTimes()
        r, c = (C.c_double * self.world)(), (C.c_double * self.world)()
        L.check(L.lib().vmx_multi_timings(self._h, C.byref(t), r, c))
        d = {k: getattr(t, k) for k, _ in t._fields_ if k != "pad"}
        d["render_ms"], d["copy_ms"] = list(r), list(c)
        return d

    def bind_texture(self, data):
        data = np.ascontiguousarray(data, dtype=np.float32)
        c = 1 if data.ndim == 2 else data.shape[2]
        L.check(L.lib().vmx_multi_bind_texture(self._h, data.ctypes.data, data.shape[1], data.shape[0], c))

    def render(self, cam, opts):
        out = np.empty((cam.image_res[1], cam.image_res[0], 5), np.float32)
        st = L.Stats()
        L.check(L.lib().vmx_multi_render(self._h, C.byref(cam), C.byref(opts), out.ctypes.data, C.byref(st)))
        return out, st.as_dict()

    def render_device(self, cam, opts, d_out_ptr):
        st = L.Stats()
        L.check(L.lib().vmx_multi_render_device(self._h, C.byref(cam), C.byref(opts), C.c_void_p(d_out_ptr), C.byref(st)))
        return st.as_dict()

    def render_bruteforce(self, cam, opts, flags=0):
        out = np.empty((cam.image_res[1], cam.image_res[0], 5), np.float32)
        st = L.Stats()
        L.check(L.lib().vmx_multi_render_bruteforce(self._h, C.byref(cam), C.byref(opts), int(flags), out.ctypes.data,
                                                    C.byref(st)))
        return out, st.as_dict()


RAYHIT_DTYPE = np.dtype([
    ("location", np.float32, 3), ("distance", np.float32), ("normal", np.float32, 3), ("tri_id", np.int32),
    ("uv", np.float32, 2), ("tri_t", np.float32), ("flags", np.uint32), ("colour", np.float32, 3),
    ("pad", np.uint32),
])
assert RAYHIT_DTYPE.itemsize == 64


def local_rows(height, stripe_rows, rank, world):
    r = C.c_uint32(0)
    L.check(L.lib().vmx_local_rows(int(height), int(stripe_rows), int(rank), int(world), C.byref(r)))
    return r.value


def local_row_indices(height, stripe_rows, rank, world):
    """global row index of every local row of (rank, world) — pure host logic"""
    stripe_rows = stripe_rows or 16
    if world <= 1:
        return np.arange(height)
    rows = []
    n_stripes = (height + stripe_rows - 1) // stripe_rows
    for s in range(rank, n_stripes, world):
        rows.extend(range(s * stripe_rows, min((s + 1) * stripe_rows, height)))
    return np.asarray(rows, dtype=np.int64)
