#!/usr/bin/env python3
"""Print the kernel launches of the last optimisation step of a rocprofv3 --kernel-trace database
(start offset, duration in us, grid, name): python3 tools/step_trace.py <results.db> [marker-kernel]"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
marker = sys.argv[2] if len(sys.argv) > 2 else "adam_kernel"
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = db.execute(f"select s.kernel_name,d.start,d.end,d.grid_size_x,d.grid_size_y from {kd} d join {ks} s "
                  "on d.kernel_id=s.id order by d.start").fetchall()
idx = [i for i, r in enumerate(rows) if marker in r[0]]
a, b = idx[-2], idx[-1]
t0 = rows[a][2]
busy = 0.0
for r in rows[a + 1:b + 1]:
    name = re.sub(r"^_ZN2rn\d+", "", r[0])[:70]
    busy += (r[2] - r[1]) / 1e3
    print(f"{(r[1] - t0) / 1e3:9.1f} {(r[2] - r[1]) / 1e3:8.1f}  gx={r[3]:8d} gy={r[4]:3d} {name}")
print(f"# kernel time of the step {busy:.1f} us over {(rows[b][2] - t0) / 1e3:.1f} us")
