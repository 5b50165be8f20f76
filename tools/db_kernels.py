"""Per-kernel durations of the LAST frame in a rocprofv3 results db (rocpd sqlite).   python tools/db_kernels.py results.db"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name,start,end,grid_x,lds_size,scratch_size from kernels order by start"))
idx = [i for i, r in enumerate(rows) if "k_camera_tables" in r[0]][-1]
t0 = rows[idx][1]; prev = t0
for r in rows[idx:]:
    name = r[0].replace("vmx::(anonymous namespace)::", "").split("(")[0][:60]
    print(f"{(r[1]-t0)/1e3:9.1f} us  gap {(r[1]-prev)/1e3:7.1f}  dur {(r[2]-r[1])/1e3:9.1f}  grid {r[3]:8d} lds {r[4]:6d} scr {r[5]:4d}  {name}")
    prev = r[2]
