"""The mesh walk of the reference-side adapter (vermilion_amd/adapter/flatten.h: the template HipPathTracer.cpp
instantiates with aiMesh*) EXECUTED on plain stand-in meshes: triangle order = MeshEngine::createBVH's push order
(mesh-major, face-minor, the face's three indices: meshEngine.cpp:659-718), and both readings of what a mesh without
UVs gets (UvRule).  Stand-in types pin no arithmetic — the walk has none; what runs is the index order and the
carry-over rule."""
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def expected(rule):
    """the same scene built independently: what createBVH would push"""
    tris = []
    last = np.zeros(6, np.float32)
    for m, (nv, nf, with_uv) in enumerate(((5, 3, True), (4, 2, False), (6, 4, True), (3, 1, False))):
        if rule == 0:
            last = np.zeros(6, np.float32)
        for f in range(nf):
            idx = (f % nv, (f + 2) % nv, (f + 1) % nv)
            pos = [c for i in idx for c in (100.0 * m + i, 0.5, -float(i))]
            nrm = [c for i in idx for c in (float(i), float(m), 1.0)]
            if with_uv:
                last = np.array([c for i in idx for c in (m + i / 16.0, i / 32.0)], np.float32)
            tris.append((np.float32(pos), np.float32(nrm), last.copy()))
    return tris


@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
def test_flatten_walk_runs_and_orders_triangles_as_createBVH(tmp_path):
    exe = tmp_path / "flatten_test"
    subprocess.run(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "vermilion_amd", "adapter"),
                    os.path.join(ROOT, "tests", "cpp", "flatten_test.cpp"), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines()
    got = {0: [], 1: []}
    rule = None
    for line in out:
        w = line.split()
        if w[0] == "rule":
            rule = int(w[1])
            assert int(w[3]) == 10 and [int(x) for x in w[5:8]] == [90, 90, 60]
        elif w[0] == "t":
            v = np.float32([float(x) for x in w[3:12]]), np.float32([float(x) for x in w[13:22]]), np.float32([float(x) for x in w[23:29]])
            got[rule].append(v)
    assert "empty 0" in out and "hollow 0 0" in out
    for rule in (0, 1):
        exp = expected(rule)
        assert len(got[rule]) == len(exp) == 10
        for t, (g, e) in enumerate(zip(got[rule], exp)):
            for a, b in zip(g, e):
                assert np.array_equal(a, b), (rule, t, a, b)
    # the two rules differ exactly on the meshes without UVs that follow a mesh with UVs (triangles 3, 4 and 9)
    differ = [t for t in range(10) if not np.array_equal(got[0][t][2], got[1][t][2])]
    assert differ == [3, 4, 9]
    assert np.array_equal(got[1][3][2], got[1][2][2]) and np.array_equal(got[1][9][2], got[1][8][2])  # carried over
    assert not got[0][3][2].any() and not got[0][9][2].any()                                          # zeros


def test_adapter_uses_the_template_and_passes_radians_through():
    src = open(os.path.join(ROOT, "vermilion_amd", "adapter", "HipPathTracer.cpp")).read()
    assert "flattenMeshes(mEng->sceneMeshes, kUvOfMeshesWithoutUvs, pos, nrm, uv)" in src
    assert "VMX_ROTATION_RADIANS" in src and "180 / 3.1415926535" not in src
