"""Per-launch PMC figures for launches longer than MINMS ms, grouped by kernel (gpurun_out/pmcp_<tag>_*)."""
import csv, glob, sys, os, re
from collections import defaultdict
tag = sys.argv[1]; minms = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
res = defaultdict(lambda: defaultdict(list))
for d in sorted(glob.glob(os.path.join(root, "gpurun_out", f"pmcp_{tag}_*"))):
    if not os.path.isdir(d): continue
    for f in glob.glob(os.path.join(d, "*", "*counter_collection.csv")):
        kt = f.replace("counter_collection", "kernel_trace")
        dur = {r["Dispatch_Id"]: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in csv.DictReader(open(kt))}
        for r in csv.DictReader(open(f)):
            m = re.search(r"(k_[a-z0-9_]+(?:<[^>]*>)?)", r["Kernel_Name"])
            if not m: continue
            ms = dur.get(r["Dispatch_Id"], 0)
            if ms >= minms:
                res[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
                res[m.group(1)]["ms"].append(ms)
for k in sorted(res):
    v = res[k]
    g = lambda c: sum(v[c]) / len(v[c]) if v.get(c) else float("nan")
    print(f"{k}: launches~{len(v.get('SQ_WAVES', []))} avg_ms={g('ms'):.3f}")
    print(f"   VALU insts {g('SQ_INSTS_VALU'):.4g}  SALU {g('SQ_INSTS_SALU'):.4g}  LDS {g('SQ_INSTS_LDS'):.4g}  VMEM_RD {g('SQ_INSTS_VMEM_RD'):.4g}  SMEM {g('SQ_INSTS_SMEM'):.4g}")
    print(f"   lane util {g('SQ_THREAD_CYCLES_VALU') / (g('SQ_ACTIVE_INST_VALU') * 64):.3f}  waves {g('SQ_WAVES'):.0f}")
    wc = g('SQ_WAVE_CYCLES')
    print(f"   wave_cycles {wc:.4g}: wait_any {g('SQ_WAIT_ANY') / wc:.3f}  wait_inst_any {g('SQ_WAIT_INST_ANY') / wc:.3f}  active_inst_any {g('SQ_ACTIVE_INST_ANY') / wc:.3f}  active_valu {g('SQ_ACTIVE_INST_VALU') / wc:.3f}  active_sca {g('SQ_ACTIVE_INST_SCA') / wc:.3f}  busy_cycles {g('SQ_BUSY_CYCLES'):.4g}")
    print(f"   L2 hit {g('TCC_HIT_sum') / (g('TCC_HIT_sum') + g('TCC_MISS_sum')):.3f} (hits {g('TCC_HIT_sum'):.4g} miss {g('TCC_MISS_sum'):.4g})  TCP->TCC read req {g('TCP_TCC_READ_REQ_sum'):.4g}  TCP accesses {g('TCP_TOTAL_CACHE_ACCESSES_sum'):.4g}")
