"""ctypes binding of the C ABI declared in include/vermilion_hip.h.

The shared library is the product; this module only loads it and declares
argument types.  There is no CPU fallback: if ``libvermilion_hip.so`` is
missing or cannot be loaded, importing the package fails loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VMX_LIB") or os.path.join(_HERE, "libvermilion_hip.so")  # VMX_LIB: A/B builds

VMX_OK = 0
VMX_ERR_INVALID = 1
VMX_ERR_NO_DEVICE = 2
VMX_ERR_HIP = 3
VMX_ERR_DEPTH = 4
VMX_ERR_NOMEM = 5

VMX_SPHERE_EMIT = 1
VMX_SAMPLING_PARITY = 0
VMX_SAMPLING_CORRECTED = 1
VMX_SAMPLING_LIBM_DOUBLE = 0x100
VMX_SAMPLING_ELIDE_DEAD = 0x200
VMX_BVH_REFERENCE = 0
VMX_BVH_SAH = 1
VMX_BVH_LBVH = 2
VMX_BVH_PLOC = 3
VMX_BF_ABS_INT = 1
VMX_ROTATION_DEGREES = 0
VMX_ROTATION_RADIANS = 1


class Sphere(C.Structure):
    _fields_ = [
        ("centre", C.c_float * 3),
        ("radius", C.c_float),
        ("colour", C.c_float * 3),
        ("flags", C.c_uint32),
        ("normal_centre", C.c_float * 3),
        ("normal_sign", C.c_float),
    ]


class CameraDesc(C.Structure):
    _fields_ = [
        ("position", C.c_float * 3),
        ("rotation_deg", C.c_float * 3),
        ("back_distance", C.c_float),
        ("back_size", C.c_float * 2),
        ("image_res", C.c_uint32 * 2),
        ("rays_per_pixel", C.c_uint32),
        ("rotation_units", C.c_uint32),
        ("rotation_rad", C.c_float * 3),
    ]


class Opts(C.Structure):
    _fields_ = [
        ("seed", C.c_uint64),
        ("early_stop", C.c_uint32),
        ("sampling", C.c_uint32),
        ("rank", C.c_uint32),
        ("world", C.c_uint32),
        ("stripe_rows", C.c_uint32),
        ("samples_per_batch", C.c_uint32),
        ("collect_counters", C.c_uint32),
        ("reserved", C.c_uint32 * 7),
    ]


class StageStats(C.Structure):
    _fields_ = [
        ("rays", C.c_uint64),
        ("inner_visits", C.c_uint64),
        ("tri_tests", C.c_uint64),
        ("tri_hits", C.c_uint64),
        ("continued", C.c_uint64),
        ("launches", C.c_uint64),
        ("ms", C.c_double),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class Stats(C.Structure):
    _fields_ = [
        ("rays_primary", C.c_uint64),
        ("rays_secondary", C.c_uint64),
        ("samples", C.c_uint64),
        ("samples_discarded", C.c_uint64),
        ("passes", C.c_uint64),
        ("kernel_launches", C.c_uint64),
        ("ms_total", C.c_double),
        ("ms_device", C.c_double),
        ("primary", StageStats),
        ("bounce", StageStats),
        ("shade", StageStats),
    ]

    def as_dict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_ if k not in ("primary", "bounce", "shade")}
        d["primary"] = self.primary.as_dict()
        d["bounce"] = self.bounce.as_dict()
        d["shade"] = self.shade.as_dict()
        return d


class SceneDesc(C.Structure):
    _fields_ = [
        ("ntris", C.c_uint32),
        ("nspheres", C.c_uint32),
        ("leaf_size", C.c_uint32),
        ("n_nodes", C.c_uint32),
        ("n_leaves", C.c_uint32),
        ("n_inner", C.c_uint32),
        ("max_depth", C.c_uint32),
        ("stack_entries", C.c_uint32),
        ("device_bytes", C.c_uint64),
        ("device", C.c_int32),
        ("pad", C.c_uint32),
    ]


K_NAMES = ["raygen", "trace_camera", "shade_camera", "trace_bounce", "shade_bounce", "tail", "fused", "resolve",
           "bruteforce", "other"]  # VMX_K_* of vermilion_hip.h


class Timings(C.Structure):
    _fields_ = [("ms", C.c_double * 10), ("longest_ms", C.c_double * 10), ("launches", C.c_uint64 * 10)]

    def as_dict(self):
        return {n: {"ms": self.ms[i], "longest_ms": self.longest_ms[i], "launches": self.launches[i]}
                for i, n in enumerate(K_NAMES)}


class MultiTimes(C.Structure):
    _fields_ = [("slowest_render_ms", C.c_double), ("gather_ms", C.c_double), ("gather_sum_ms", C.c_double),
                ("assemble_ms", C.c_double), ("wall_ms", C.c_double), ("world", C.c_uint32), ("pad", C.c_uint32)]


class RayHit(C.Structure):
    _fields_ = [
        ("location", C.c_float * 3),
        ("distance", C.c_float),
        ("normal", C.c_float * 3),
        ("tri_id", C.c_int32),
        ("uv", C.c_float * 2),
        ("tri_t", C.c_float),
        ("flags", C.c_uint32),
        ("colour", C.c_float * 3),
        ("pad", C.c_uint32),
    ]


# every symbol include/vermilion_hip.h declares: name -> (restype, argtypes)
_P = C.c_void_p
SYMBOLS = {
    "vmx_abi_version": (C.c_int, []),
    "vmx_last_error": (C.c_char_p, []),
    "vmx_device_count": (C.c_int, []),
    "vmx_default_spheres": (C.POINTER(Sphere), [C.POINTER(C.c_uint32)]),
    "vmx_scene_create": (C.c_int, [_P, _P, _P, C.c_uint32, _P, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(_P)]),
    "vmx_scene_create_ex": (C.c_int, [_P, _P, _P, C.c_uint32, _P, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int,
                                     C.POINTER(_P)]),
    "vmx_scene_destroy": (C.c_int, [_P]),
    "vmx_scene_bind_texture": (C.c_int, [_P, _P, C.c_uint32, C.c_uint32, C.c_uint32]),
    "vmx_scene_describe": (C.c_int, [_P, C.POINTER(SceneDesc)]),
    "vmx_scene_timings": (C.c_int, [_P, C.POINTER(Timings)]),
    "vmx_scene_bvh": (C.c_int, [_P, _P, _P, _P, _P, _P]),
    "vmx_trace": (C.c_int, [_P, _P, _P, C.c_uint32, _P, _P]),
    "vmx_raycast": (C.c_int, [_P, _P, _P, C.c_uint32, _P]),
    "vmx_primary_ids": (C.c_int, [_P, C.POINTER(CameraDesc), C.POINTER(Opts), C.c_uint32, _P, _P]),
    "vmx_radiance": (C.c_int, [_P, _P, _P, C.c_uint32, C.POINTER(Opts), _P, C.POINTER(Stats)]),
    "vmx_trig": (C.c_int, [_P, C.c_uint32, _P, _P, C.c_int]),
    "vmx_local_rows": (C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]),
    "vmx_render": (C.c_int, [_P, C.POINTER(CameraDesc), C.POINTER(Opts), _P, C.POINTER(Stats)]),
    "vmx_render_device": (C.c_int, [_P, C.POINTER(CameraDesc), C.POINTER(Opts), _P, _P, C.POINTER(Stats)]),
    "vmx_render_bruteforce": (C.c_int, [_P, C.POINTER(CameraDesc), C.POINTER(Opts), C.c_uint32, _P, C.POINTER(Stats)]),
    "vmx_render_bruteforce_device": (C.c_int, [_P, C.POINTER(CameraDesc), C.POINTER(Opts), C.c_uint32, _P, _P,
                                              C.POINTER(Stats)]),
    "vmx_multi_create": (C.c_int, [_P, _P, _P, C.c_uint32, _P, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_int),
                                  C.c_uint32, C.POINTER(_P)]),
    "vmx_multi_destroy": (C.c_int, [_P]),
    "vmx_multi_world": (C.c_uint32, [_P]),
    "vmx_multi_routes": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "vmx_multi_timings": (C.c_int, [_P, C.POINTER(MultiTimes), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "vmx_multi_bind_texture": (C.c_int, [_P, _P, C.c_uint32, C.c_uint32, C.c_uint32]),
    "vmx_multi_render": (C.c_int, [_P, C.POINTER(CameraDesc), C.POINTER(Opts), _P, C.POINTER(Stats)]),
    "vmx_multi_render_device": (C.c_int, [_P, C.POINTER(CameraDesc), C.POINTER(Opts), _P, C.POINTER(Stats)]),
    "vmx_multi_render_bruteforce": (C.c_int, [_P, C.POINTER(CameraDesc), C.POINTER(Opts), C.c_uint32, _P,
                                             C.POINTER(Stats)]),
    "vmx_quantize_device": (C.c_int, [_P, C.c_uint64, _P, _P, C.c_int, _P]),
    "vmx_assemble_device": (C.c_int, [_P, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _P, C.c_int, _P]),
}


def load(path=None):
    """Load the HIP library and declare its prototypes.  Raises if it is absent."""
    path = path or LIB_PATH
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7.
    # If torch is importable, load it first so this library binds to the same
    # runtime (two runtimes in one process cannot both open the GPU).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(path):
        raise ImportError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  vermilion_amd has no CPU fallback."
        )
    lib = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    return lib


class VmxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"vermilion_hip error {code}: {msg}")
        self.code = code


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = load()
    return _lib


def check(code):
    if code != VMX_OK:
        raise VmxError(code, lib().vmx_last_error().decode("utf-8", "replace"))
