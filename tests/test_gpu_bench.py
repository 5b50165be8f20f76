"""bench.py's contract with the driver (one JSON line on rank 0, the fields BASELINE.json's metric needs, the
`roofline` and `cpu_baseline` objects) — on a small frame (large enough for the split pipeline: >= 8 M path slots), in
its three launch forms."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--width", "640", "--height", "360", "--spp", "64", "--steps", "2", "--warmup", "1", "--cpu-spp", "4",
         "--corrected-spp", "8"]


def run(cmd):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]  # ONE JSON line, from rank 0
    return json.loads(lines[0])


def check_common(d, n_gpus):
    assert d["metric"].startswith("Mrays/sec") and d["unit"] == "Mrays/s" and d["higher_is_better"] is True
    assert d["n_gpus"] == n_gpus and d["steps"] == 2 and d["warmup"] == 1
    assert d["value"] > 0 and d["ms_per_step"] > 0 and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - d["config"]["rays_per_frame"] / d["ms_per_step"] / 1e3) / d["value"] < 0.02
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r


def test_single_gpu_line_carries_roofline_cpu_baseline_and_the_extra_frames():
    d = run([sys.executable, "bench.py"] + SMALL)
    check_common(d, 1)
    assert d["scaling"] in ("weak", "strong")
    # (the PMC counters in profiles/ belong to the 1920x1080x256 frame: on another workload the issue fraction and the
    # traffic are null; `frac` is the work-based figure, computed from the live counters pass, and is always there)
    r = d["roofline"]
    assert r["traffic"] is None and "note" in r and r.get("issue_frac") is None and r["hbm_frac_whole_frame"] is None
    # (on this small frame the kernel that takes the most time need not be a traversal kernel; only those have a work model)
    tc = r["other_kernels"]["trace_camera"] if "trace_camera" in r.get("other_kernels", {}) else r
    assert 0 < tc["frac"] <= 1 and tc["frac"] == tc["useful_valu_frac"] and tc["work"]["rays"] == 640 * 360 * 64
    assert abs(tc["achieved"] / tc["peak"] - tc["frac"]) < 2e-3
    assert r["frac"] is None or 0 < r["frac"] <= 1
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == "Mrays/s" and cb["sample"]
    rf = d["reference_frame"]
    assert rf["ms_per_frame"] > 0 and rf["passes"] >= 1 and sum(rf["kernel_ms"].values()) <= rf["device_ms"] * 1.02
    assert rf["library_default_ms"] > 0
    cf = d["corrected_frame"]
    assert cf["spp"] == 8 and cf["rays_per_sample"] > 5 and cf["Mrays_per_s"] > 0
    cq = d["corrected_frame_quality_bvh"]
    assert cq["spp"] == 8 and cq["rays_per_frame"] > 0 and cq["Mrays_per_s"] > 0
    assert d["bruteforce_frame"]["ms_per_frame"] > 0 and d["quality_bvh"]["inner_visits_per_ray"] > 0
    # the headline counts full RayCasts; the library's default form and the elided form are the same frame,
    # bit for bit, in less time: reported beside it
    assert "RayCast" in d["config"]["ray"]
    so = d["sorted_frame"]
    assert so["frame_bit_identical_to_headline"] is True and so["rays_per_frame"] == d["config"]["rays_per_frame"] and so["ms_per_frame"] > 0
    el = d["elided_frame"]
    assert el["frames_bit_identical_to_headline"] is True
    assert 0 < el["fixed_count"]["rays_per_frame"] < 0.5 * d["config"]["rays_per_frame"]
    assert 0 < el["early_stop"]["rays_per_frame"] < rf["rays_per_frame"]
    fm = d["frame_ms"]
    assert fm["bit_identical"] is True and fm["every_ray_a_full_RayCast (headline)"] == d["ms_per_step"]
    assert fm["VMX_SAMPLING_ELIDE_DEAD"] == el["fixed_count"]["ms_per_frame"]


def test_two_ranks_with_the_extra_frames():
    """... and once with the extra frames (early stop, full shading, elision) through the sharded path"""
    d = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
             "--master-port", "29534", "bench.py", "--gpus", "2", "--backend", "gloo", "--device", "0", "--no-cpu-baseline"] + SMALL)
    check_common(d, 2)
    assert d["elided_frame"]["frames_bit_identical_to_headline"] is True and d["sorted_frame"]["frame_bit_identical_to_headline"] is True
    x = d["exchange"]  # the exchange step timed apart from the rendering
    assert len(x["rank_render_ms"]) == 2 and x["slowest_rank_render_ms"] == max(x["rank_render_ms"]) > 0
    assert x["gather_ms"] > 0 and x["assemble_ms"] > 0 and len(x["gather_ms_by_rank"]) == 2


def test_multi_device_form_in_one_process():
    d = run([sys.executable, "bench.py", "--multi", "0,0"] + SMALL)
    check_common(d, 2)
    assert d["config"]["distinct_devices"] == 1 and "vmx_multi" in d["config"]["parallelism"]
    assert d["reference_frame"]["ms_per_frame"] > 0
    x = d["exchange"]
    assert x["slowest_rank_render_ms"] > 0 and x["gather_ms"] > 0 and x["assemble_ms"] > 0 and x["wall_ms"] >= x["slowest_rank_render_ms"]
    assert [r["device"] for r in x["routes"]] == [0, 0] and "parallel_efficiency" not in d
    d2 = run([sys.executable, "bench.py", "--multi", "0,0", "--n1-ms", "10.0"] + SMALL)
    assert abs(d2["parallel_efficiency"] - 10.0 / (2 * d2["ms_per_step"])) < 1e-3


def test_two_ranks_under_torch_distributed_run():
    """the driver's N > 1 launch line, rehearsed on the one GPU (both ranks on device 0, frames gathered with gloo)"""
    d = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
             "--master-port", "29533", "bench.py", "--gpus", "2", "--backend", "gloo", "--device", "0", "--no-extras"] + SMALL)
    check_common(d, 2)
    assert d["scaling"] == "strong" and d["config"]["parallelism"].startswith("stripes")
    assert d["exchange"]["gather_ms"] > 0 and d["exchange"]["assemble_ms"] > 0
