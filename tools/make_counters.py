"""profiles/counters.json from the PMC passes of tools/pmc.sh (gpurun_out/pmc_<tag>_N): for every kernel
class of the frame (vmx_timings names) the counters of its LONGEST launch, and what bench.py's roofline
object needs from them: wave-level VALU instructions per launch (VALU issue roof), vector-L1 accesses per
launch (lookup roof), HBM bytes per launch.  HBM bytes follow MI355X_MICROARCH.md's HBM section:
FETCH_SIZE and WRITE_SIZE come from separate --pmc passes, are in KiB, FETCH_SIZE is doubled on gfx950
(128-B requests tallied at 64 B for 16-B-per-lane loads).  Also copies the kernel-trace stats CSV of the
same command to profiles/<tag>_kernel_stats.csv.

usage: python tools/make_counters.py <tag> [workload description]
"""
import csv
import glob
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# (vmx_timings name, substring of the demangled kernel name) — the kernels of the bench's headline frame: every Radiance
# step shaded in full (vmx_opts.reserved[0] bit 8, bench.py HEADLINE_FORM)
CLASSES = [
    ("raygen", "k_raygen<0>"), ("trace_camera", "k_trace_w<0, false, false>"), ("shade_camera", "k_shade<0, false, false, 0>"),
    # (one-phase bounce generations go through dense ray records, all kept: the SORT instantiation with WorkDev::keep_all)
    ("trace_bounce", "k_trace_w<1, false, true>"), ("shade_bounce", "k_shade<1, false, false, 2>"),
    ("tail", "k_paths<false, 2"), ("fused", "k_paths<false, 0"), ("resolve", "k_resolve"),
]
HEADLINE_FORM = 0x100
N_SIMD, N_CU, NOMINAL_HZ = 1024, 256, 2.4e9  # MI355X_MICROARCH.md: 256 CUs x 4 SIMD-32, 2.4 GHz max clock


def longest(pass_dir, pat):
    out = {}
    # (gpurun merges a call's output into the local directory: keep only the newest run of a pass)
    files = sorted(glob.glob(os.path.join(pass_dir, "*", "*counter_collection.csv")), key=os.path.getmtime)
    for f in files[-1:]:
        kt = f.replace("counter_collection", "kernel_trace")
        dur = {r["Dispatch_Id"]: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
               for r in csv.DictReader(open(kt)) if pat in r["Kernel_Name"]}
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
    workload = sys.argv[2] if len(sys.argv) > 2 else "sponza260k 1920x1080 256spp, reference sampling, fixed spp"
    kernels = {}
    for name, pat in CLASSES:
        c, ms = {}, []
        for d in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", f"pmc_{tag}_*"))):
            if os.path.isdir(d):
                v = longest(d, pat)
                if "ms" in v:
                    ms.append(v.pop("ms"))
                c.update(v)
        if not ms:
            continue
        t = sum(ms) / len(ms) * 1e-3
        k = {"kernel": pat, "ms_under_pmc": round(t * 1e3, 4), "counters": c}
        d = {}
        if "GRBM_GUI_ACTIVE" in c:  # summed over the 8 XCDs (MI355X_MICROARCH.md, DVFS give-back)
            d["effective_clock_GHz"] = c["GRBM_GUI_ACTIVE"] / 8 / t / 1e9
        if "SQ_INSTS_VALU" in c:
            d["valu_wave_insts_per_launch"] = c["SQ_INSTS_VALU"]
            if "SQ_INSTS_SALU" in c:  # scalar instructions take issue slots beside the vector ones (tools/sload_probe.hip)
                d["salu_wave_insts_per_launch"] = c["SQ_INSTS_SALU"]
            # a wave64 VALU instruction occupies its SIMD-32 for 2 cycles: peak = SIMDs x clock / 2
            d["valu_issue_frac_at_nominal_clock"] = c["SQ_INSTS_VALU"] / t / (N_SIMD * NOMINAL_HZ / 2)
            if "effective_clock_GHz" in d:
                d["valu_issue_frac_at_effective_clock"] = c["SQ_INSTS_VALU"] / t / (N_SIMD * d["effective_clock_GHz"] * 1e9 / 2)
            if "SQ_THREAD_CYCLES_VALU" in c:
                d["valu_lane_utilization"] = c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_INSTS_VALU"] * 64)
        if "TCP_TOTAL_CACHE_ACCESSES_sum" in c:
            d["l1_accesses_per_launch"] = c["TCP_TOTAL_CACHE_ACCESSES_sum"]
            d["l1_accesses_per_clk_per_cu_at_nominal_clock"] = c["TCP_TOTAL_CACHE_ACCESSES_sum"] / t / NOMINAL_HZ / N_CU
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            d["hbm_bytes_per_launch"] = int(c["FETCH_SIZE"] * 1024 * 2 + c["WRITE_SIZE"] * 1024)
            d["hbm_GBs"] = d["hbm_bytes_per_launch"] / t / 1e9
        if "TCC_HIT_sum" in c:
            d["l2_hit_rate"] = c["TCC_HIT_sum"] / max(c["TCC_HIT_sum"] + c.get("TCC_MISS_sum", 0.0), 1.0)
        if "SQ_WAIT_ANY" in c and "SQ_WAVE_CYCLES" in c:
            d["wait_any_frac_per_wave"] = c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]
        k["derived"] = d
        kernels[name] = k
    out = {
        "round": 4,
        "tag": tag,
        "form": HEADLINE_FORM,
        "workload": workload,
        "command": "rocprofv3 --kernel-trace --pmc <one group per pass> --output-format csv -- python3 bench.py "
                   f"--steps 2 --warmup 1 --no-cpu-baseline --no-extras  (tools/pmc.sh TAG={tag}; this file: tools/make_counters.py)",
        "hbm_correction": "FETCH_SIZE x2 (gfx950 tallies 128-B requests at 64 B for 16-B-per-lane loads, "
                          "MI355X_MICROARCH.md 'HBM'); WRITE_SIZE as read; separate --pmc passes; KiB -> bytes x1024",
        "valu_issue_peak": "1024 SIMD-32 x 2.4 GHz / 2 cycles per wave64 instruction = 1.2288e12 wave-instructions/s "
                           "(MI355X_MICROARCH.md 'Wave scheduling', 'Per-instruction cycle constants')",
        "kernels": kernels,
    }
    json.dump(out, open(os.path.join(ROOT, "profiles", "counters.json"), "w"), indent=1)
    for f in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", f"trace_{tag}", "*", "*kernel_stats.csv")), key=os.path.getmtime)[-1:]:
        shutil.copy(f, os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.csv"))
    for name, k in kernels.items():
        d = k["derived"]
        print(f"{name:14s} {k['ms_under_pmc']:8.3f} ms  VALU issue {d.get('valu_issue_frac_at_nominal_clock', float('nan')):.3f} "
              f"(eff clk {d.get('effective_clock_GHz', float('nan')):.2f} GHz: {d.get('valu_issue_frac_at_effective_clock', float('nan')):.3f})  "
              f"L1 acc/clk/CU {d.get('l1_accesses_per_clk_per_cu_at_nominal_clock', float('nan')):.3f}  HBM {d.get('hbm_GBs', float('nan')):.0f} GB/s  "
              f"L2 hit {d.get('l2_hit_rate', float('nan')):.2f}  lane util {d.get('valu_lane_utilization', float('nan')):.2f}")


if __name__ == "__main__":
    main()
