"""profiles/traffic.json from the PMC passes of tools/pmc.sh (gpurun_out/pmc_<tag>_N).

Picks, in every pass, the longest launch of the camera-ray traversal kernel
(k_trace_w<0> — the single 256-samples-per-pixel launch of the fixed-spp frame) and records its counters.  HBM bytes follow
MI355X_MICROARCH.md's HBM section: FETCH_SIZE and WRITE_SIZE are collected in
separate --pmc passes, are in KiB, and FETCH_SIZE is doubled on gfx950 for
16-byte-per-lane loads (128-B requests tallied at 64 B).

usage: python tools/make_traffic.py <tag> <algorithmic_bytes_per_launch> <rays_per_launch>
"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = "k_trace_w<0>"


def longest_launch(pass_dir):
    out = {}
    for f in glob.glob(os.path.join(pass_dir, "*", "*counter_collection.csv")):
        kt = f.replace("counter_collection", "kernel_trace")
        dur = {r["Dispatch_Id"]: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
               for r in csv.DictReader(open(kt)) if KERNEL in r["Kernel_Name"]}
        if not dur:
            continue
        best = max(dur, key=dur.get)
        out["ms"] = dur[best]
        for r in csv.DictReader(open(f)):
            if r["Dispatch_Id"] == best:
                out[r["Counter_Name"]] = out.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return out


def main():
    tag = sys.argv[1]
    alg = int(sys.argv[2])
    rays = int(sys.argv[3])
    c = {}
    ms = []
    for d in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", f"pmc_{tag}_*"))):
        if os.path.isdir(d):
            v = longest_launch(d)
            if "ms" in v:
                ms.append(v.pop("ms"))
            c.update(v)
    fetch = c["FETCH_SIZE"] * 1024 * 2
    write = c["WRITE_SIZE"] * 1024
    valu, lanes = c["SQ_INSTS_VALU"], c["SQ_THREAD_CYCLES_VALU"]
    out = {
        "round": 1,
        "tag": tag,
        "command": "rocprofv3 --kernel-trace --pmc <one group per pass> --output-format csv -- python3 bench.py "
                   f"--steps 1 --warmup 0 --no-cpu-baseline  (tools/pmc.sh, TAG={tag}; this file: tools/make_traffic.py)",
        "kernel": f"{KERNEL}: the single 256-samples-per-pixel launch of the fixed-spp frame ({rays / 1e6:.1f}M depth-0 rays)",
        "FETCH_SIZE_KB_per_launch": c["FETCH_SIZE"],
        "WRITE_SIZE_KB_per_launch": c["WRITE_SIZE"],
        "correction": "FETCH_SIZE x2 (gfx950 tallies 128-B requests at 64 B for 16-B-per-lane loads, "
                      "MI355X_MICROARCH.md 'HBM'); WRITE_SIZE as read; separate --pmc passes; KB -> bytes x1024",
        "trace_kernel_hbm_bytes_per_launch": int(fetch + write),
        "algorithmic_bytes_per_launch": alg,
        "expected_stream_bytes": f"16-B camera-ray read + 8-B hit write per ray = {rays * 24 / 1e9:.1f} GB",
        "l2_hit_rate": c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]),
        "waves_per_launch": c["SQ_WAVES"],
        "valu_insts_per_launch": valu,
        "salu_insts_per_launch": c["SQ_INSTS_SALU"],
        "vmem_read_insts_per_launch": c["SQ_INSTS_VMEM_RD"],
        "lds_insts_per_launch": c["SQ_INSTS_LDS"],
        "valu_lane_utilization": lanes / (valu * 64),
        # wave-level instruction counts per ray: what the loop's cost is made of
        "valu_wave_insts_per_64_rays": valu / (rays / 64),
        "salu_wave_insts_per_64_rays": c["SQ_INSTS_SALU"] / (rays / 64),
        "valu_lane_ops_per_ray": lanes / rays,
        "wait_any_frac_per_wave": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"],
        "avg_ms_under_pmc": sum(ms) / len(ms),
    }
    json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
