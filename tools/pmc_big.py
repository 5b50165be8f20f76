"""Per-launch PMC figures for the large k_primary / k_bounce launches of the fixed-spp frame."""
import csv, glob, sys, os, re, json
from collections import defaultdict
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
res = defaultdict(lambda: defaultdict(list))
for d in sorted(glob.glob(os.path.join(root, "gpurun_out", f"pmc_{tag}_*"))):
    if not os.path.isdir(d): continue
    for f in glob.glob(os.path.join(d, "*", "*counter_collection.csv")):
        kt = f.replace("counter_collection", "kernel_trace")
        dur = {}
        for r in csv.DictReader(open(kt)):
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        for r in csv.DictReader(open(f)):
            m = re.search(r"(k_[a-z0-9_]+(?:<[^>]*>)?)", r["Kernel_Name"])
            if not m: continue
            k = m.group(1)
            ms = dur.get(r["Dispatch_Id"], 0)
            grid = int(r["Grid_Size"]) if "Grid_Size" in r else 0
            if (k.startswith("k_trace_q<false, 0>") and ms > 10.0) or (k == "k_trace_q<false, 1>" and ms > 2.0) \
                    or (k.startswith("k_shade<0>") and ms > 2.0):
                res[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                res[k]["_ms_" + r["Counter_Name"]].append(ms)
out = {}
for k in res:
    print(k)
    out[k] = {}
    for c in sorted(res[k]):
        v = res[k][c]
        if c.startswith("_ms_"): continue
        ms = res[k]["_ms_" + c]
        out[k][c] = {"per_launch": sum(v) / len(v), "launches": len(v), "avg_ms_under_pmc": sum(ms) / len(ms)}
        print(f"   {c:26s} per-launch={sum(v)/len(v):.6g}  n={len(v)}  avg_ms={sum(ms)/len(ms):.3f}")
json.dump(out, open(os.path.join(root, "gpurun_out", f"pmc_{tag}_big.json"), "w"), indent=1)
