"""Counters of k_trace_pool and k_trace_w<1> from the passes of tools/pool_pmc.sh (longest dispatch of each kernel)."""
import csv, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04pool"
KERNELS = (("k_trace_w<1>", "k_trace_w<1, false, false>"), ("k_trace_pool", "k_trace_pool"))
res = {k: {} for k, _ in KERNELS}
for d in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", f"pmc_{tag}_*"))):
    if not os.path.isdir(d):
        continue
    for f in sorted(glob.glob(os.path.join(d, "*", "*counter_collection.csv")), key=os.path.getmtime)[-1:]:
        kt = f.replace("counter_collection", "kernel_trace")
        rows = list(csv.DictReader(open(kt)))
        for name, pat in KERNELS:
            dur = {r["Dispatch_Id"]: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows if pat in r["Kernel_Name"]}
            if not dur:
                continue
            best = max(dur, key=dur.get)
            res[name].setdefault("ms", []).append(dur[best])
            for r in csv.DictReader(open(f)):
                if r["Dispatch_Id"] == best:
                    res[name][r["Counter_Name"]] = res[name].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
for name, c in res.items():
    if "ms" not in c:
        continue
    ms = sum(c["ms"]) / len(c["ms"])
    print(f"== {name}: {ms:.3f} ms under the counters")
    for k in sorted(c):
        if k != "ms":
            print(f"   {k:28s} {c[k]:.6g}")
    if "SQ_INSTS_VALU" in c and "SQ_INSTS_SALU" in c:
        print(f"   VALU + SALU wave-instructions {c['SQ_INSTS_VALU'] + c['SQ_INSTS_SALU']:.6g}; lane utilisation "
              f"{c.get('SQ_THREAD_CYCLES_VALU', 0) / max(c['SQ_INSTS_VALU'] * 64, 1):.3f}")
