"""Summarise rocprofv3 --pmc CSVs (gpurun_out/pmc_<tag>_N) per kernel family."""
import csv, glob, sys, json, os
from collections import defaultdict
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
agg = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for d in sorted(glob.glob(os.path.join(root, "gpurun_out", f"pmc_{tag}_*"))):
    if not os.path.isdir(d): continue
    for f in glob.glob(os.path.join(d, "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            import re
            m = re.search(r"(k_[a-z0-9_]+(?:<[^>]*>)?)", name)
            short = m.group(1) if m else name.split("(")[0]
            # big launches only for k_primary: separate fixed-spp frames (grid >= 1000 blocks, many items)
            agg[short][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[short][r["Counter_Name"]] += 1
out = {}
for k in sorted(agg):
    out[k] = {c: {"sum": agg[k][c], "dispatches": cnt[k][c]} for c in sorted(agg[k])}
    print(k)
    for c in sorted(agg[k]):
        print(f"   {c:28s} sum={agg[k][c]:.6g}  dispatches={cnt[k][c]}  per-dispatch={agg[k][c]/cnt[k][c]:.6g}")
json.dump(out, open(os.path.join(root, "gpurun_out", f"pmc_{tag}_summary.json"), "w"), indent=1)
