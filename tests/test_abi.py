"""The C-ABI library loads on a CPU-only box and exports every symbol
include/vermilion_hip.h declares; struct layouts match the header."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "vermilion_hip.h")

from vermilion_amd import _lib as L  # noqa: E402


def header_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vmx_[a-z0-9_]+)\s*\(", src)))


def test_header_is_valid_c99():
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-x", "c", HEADER], check=True)


def test_library_exports_every_declared_symbol(hip_lib):
    fns = header_functions()
    assert len(fns) >= 15
    for name in fns:
        assert hasattr(hip_lib, name), f"{name} declared in the header but not exported"
    assert sorted(L.SYMBOLS) == fns, "ctypes binding and header disagree on the function list"
    assert hip_lib.vmx_abi_version() == 2


def test_struct_layouts_match_header(tmp_path):
    names = {"vmx_sphere": L.Sphere, "vmx_camera": L.CameraDesc, "vmx_opts": L.Opts, "vmx_stage_stats": L.StageStats,
             "vmx_stats": L.Stats, "vmx_scene_desc": L.SceneDesc, "vmx_rayhit": L.RayHit, "vmx_timings": L.Timings,
             "vmx_multi_times": L.MultiTimes}
    prog = '#include <stdio.h>\n#include <stddef.h>\n#include "vermilion_hip.h"\nint main(void){\n'
    for n in names:
        prog += f'printf("{n} %zu\\n", sizeof({n}));\n'
    prog += 'printf("off_seed %zu\\n", offsetof(vmx_opts, seed));\n'
    prog += 'printf("off_primary %zu\\n", offsetof(vmx_stats, primary));\n'
    prog += 'printf("off_units %zu\\n", offsetof(vmx_camera, rotation_units));\n'
    prog += 'printf("off_flags %zu\\n", offsetof(vmx_rayhit, flags));\nreturn 0;}\n'
    src = tmp_path / "sz.c"
    src.write_text(prog)
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = dict(l.split() for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for n, ct in names.items():
        assert int(out[n]) == C.sizeof(ct), n
    assert int(out["off_seed"]) == L.Opts.seed.offset
    assert int(out["off_primary"]) == L.Stats.primary.offset
    assert int(out["off_flags"]) == L.RayHit.flags.offset
    assert int(out["off_units"]) == L.CameraDesc.rotation_units.offset
    assert C.sizeof(L.RayHit) == 64 and C.sizeof(L.Sphere) == 48
    src = open(HEADER).read()
    assert len(L.K_NAMES) == int(re.search(r"#define VMX_K_COUNT (\d+)", src).group(1))


def test_default_spheres_are_the_reference_eight(hip_lib):
    import oracle_lib as O
    n = C.c_uint32(0)
    p = hip_lib.vmx_default_spheres(C.byref(n))
    m = C.c_uint32(0)
    q = O.lib().orc_default_spheres(C.byref(m))
    assert n.value == 8 and m.value == 8
    a = np.frombuffer(C.string_at(p, 48 * 8), np.uint32)
    b = np.frombuffer(C.string_at(q, 48 * 8), np.uint32)
    assert np.array_equal(a, b)
    s = p[1]  # light 2, meshEngine.cpp:410-418
    assert list(s.centre) == [0.0, 3300.0, 1300.0] and s.radius == 250.0 and s.flags == 1
    assert list(p[3].centre) == [0.0, 50001000.0, 0.0]  # ceiling sphere, meshEngine.cpp:452


def test_argument_errors_do_not_need_a_gpu(hip_lib):
    h = C.c_void_p()
    pos = np.zeros((1, 9), np.float32)
    rc = hip_lib.vmx_scene_create(None, pos.ctypes.data, None, 1, None, 0, 4, 0, C.byref(h))
    assert rc == L.VMX_ERR_INVALID and b"positions" in hip_lib.vmx_last_error()
    rc = hip_lib.vmx_scene_create(pos.ctypes.data, pos.ctypes.data, None, 0, None, 0, 4, 0, C.byref(h))
    assert rc == L.VMX_ERR_INVALID
    rc = hip_lib.vmx_scene_create(pos.ctypes.data, pos.ctypes.data, None, 1, None, 3, 4, 0, C.byref(h))
    assert rc == L.VMX_ERR_INVALID
    r = C.c_uint32()
    assert hip_lib.vmx_local_rows(100, 16, 5, 4, C.byref(r)) == L.VMX_ERR_INVALID
    assert hip_lib.vmx_render(None, None, None, None, None) == L.VMX_ERR_INVALID
    assert hip_lib.vmx_scene_destroy(None) == L.VMX_OK
    # BruteForceTracer and the multi-device entry points follow the same convention
    assert hip_lib.vmx_render_bruteforce(None, None, None, 0, None, None) == L.VMX_ERR_INVALID
    assert hip_lib.vmx_multi_render(None, None, None, None, None) == L.VMX_ERR_INVALID
    assert hip_lib.vmx_multi_destroy(None) == L.VMX_OK and hip_lib.vmx_multi_world(None) == 0
    m = C.c_void_p()
    devs = (C.c_int * 2)(0, 1)
    assert hip_lib.vmx_multi_create(pos.ctypes.data, pos.ctypes.data, None, 1, None, 0, 4, 0, devs, 0, C.byref(m)) == L.VMX_ERR_INVALID
    assert hip_lib.vmx_multi_create(pos.ctypes.data, pos.ctypes.data, None, 1, None, 0, 4, 0, None, 2, C.byref(m)) == L.VMX_ERR_INVALID


def test_no_device_fails_loudly(hip_lib):
    """No CPU fallback: without a HIP device scene creation is an error."""
    if hip_lib.vmx_device_count() > 0:
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    pos = np.array([[0, 0, 0, 1, 0, 0, 0, 1, 0]], np.float32)
    rc = hip_lib.vmx_scene_create(pos.ctypes.data, pos.ctypes.data, None, 1, None, 0, 4, 0, C.byref(h))
    assert rc == L.VMX_ERR_NO_DEVICE and not h.value
    assert b"no CPU path" in hip_lib.vmx_last_error()
    import vermilion_amd as va
    with pytest.raises(va.VmxError):
        va.Scene(pos, pos)
    with pytest.raises(va.VmxError) as e:
        va.MultiScene(pos, pos, devices=[0, 1])
    assert e.value.code == L.VMX_ERR_NO_DEVICE


def test_local_rows_matches_host_logic(hip_lib):
    import vermilion_amd as va
    for H, R, world in ((1080, 16, 8), (37, 16, 3), (16, 16, 4), (5, 2, 2), (2160, 16, 8), (100, 7, 5)):
        total = 0
        seen = []
        for rank in range(world):
            rows = va.local_row_indices(H, R, rank, world)
            assert va.local_rows(H, R, rank, world) == len(rows)
            total += len(rows)
            seen.extend(rows.tolist())
        assert total == H and sorted(seen) == list(range(H))
    assert va.local_rows(123, 16, 0, 1) == 123 and va.local_rows(123, 0, 0, 0) == 123


def test_product_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "vermilion_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip", "Makefile")):
                text = open(os.path.join(dp, f), errors="replace").read()
                assert "oracle_lib" not in text and "vmx_oracle" not in text and "orc_" not in text, f
    syms = subprocess.run(["nm", "-D", L.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "orc_" not in syms
