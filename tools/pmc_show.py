"""Print per-kernel PMC sums from gpurun_out/<prefix>_<tag>_*/ (rocprofv3 --pmc CSVs)."""
import csv, glob, re, sys, os
from collections import defaultdict
prefix, tag = sys.argv[1], sys.argv[2]
pat = sys.argv[3] if len(sys.argv) > 3 else "trace"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
agg = defaultdict(lambda: defaultdict(float))
for f in glob.glob(os.path.join(root, "gpurun_out", f"{prefix}_{tag}_*", "*", "*counter_collection.csv")):
    kt = f.replace("counter_collection", "kernel_trace")
    dur = {r["Dispatch_Id"]: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in csv.DictReader(open(kt))}
    seen = set()
    for r in csv.DictReader(open(f)):
        m = re.search(r"k_[a-z_]+(<[^>]*>)?", r["Kernel_Name"])
        k = m.group(0) if m else r["Kernel_Name"][:30]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if (f, r["Dispatch_Id"]) not in seen:
            seen.add((f, r["Dispatch_Id"]))
            agg[k]["_ms_" + os.path.basename(os.path.dirname(os.path.dirname(f)))] += dur.get(r["Dispatch_Id"], 0)
for k in sorted(agg):
    if pat in k:
        print(k)
        for c in sorted(agg[k]):
            print("   %-30s %.6g" % (c, agg[k][c]))
