"""Under-filled launches in a rocprofv3 rocpd database: kernels whose grid has fewer workgroups than `cus` (default 256), by time.
usage: python tools/rocpd_grids.py <db> <steps> [cus]"""
import sqlite3
import sys

db, steps = sqlite3.connect(sys.argv[1]), int(sys.argv[2])
cus = int(sys.argv[3]) if len(sys.argv) > 3 else 256
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
gx = next((c for c in cols if c.lower() in ("grid_x", "grid_size_x", "grid_size")), None)
wx = next((c for c in cols if c.lower() in ("workgroup_x", "workgroup_size_x", "workgroup_size")), None)
if gx is None or wx is None:
    print("columns:", cols)
    sys.exit(0)
gy, gz = gx.replace("x", "y") if "x" in gx else None, gx.replace("x", "z") if "x" in gx else None
wy, wz = wx.replace("x", "y") if "x" in wx else None, wx.replace("x", "z") if "x" in wx else None
expr = f"(({gx})/({wx}))" + (f"*(({gy})/({wy}))*(({gz})/({wz}))" if gy in cols and wy in cols else "")
q = f"select name, {expr} as wgs, count(*), sum(end-start) from kernels group by name, wgs having wgs < {cus} order by 4 desc limit 40"
tot = 0.0
for name, wgs, n, t in db.execute(q):
    tot += t
    print(f"{t / steps / 1e6:8.3f} ms {n / steps:7.1f} launches  {int(wgs):5d} workgroups  {name[:100]}")
print(f"total in launches with < {cus} workgroups: {tot / steps / 1e6:.2f} ms/step")
