#!/usr/bin/env python
"""Per-(kernel, grid) view of a rocprofv3 (rocpd sqlite) kernel trace: which SHAPES of a templated kernel the time goes to.
    tools/trace_by_grid.py trace.db nsteps [name substring]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _names import clean  # noqa: E402
import sqlite3

db, nsteps = sys.argv[1], int(sys.argv[2])
sub = sys.argv[3] if len(sys.argv) > 3 else ""
c = sqlite3.connect(db).cursor()
cols = [r[1] for r in c.execute("pragma table_info(kernels)").fetchall()]
gx = next((n for n in ("grid_x", "grid_size_x", "grid_size") if n in cols), None)
gy = "grid_y" if "grid_y" in cols else ("grid_size_y" if "grid_size_y" in cols else None)
gz = "grid_z" if "grid_z" in cols else ("grid_size_z" if "grid_size_z" in cols else None)
if gx is None:
    print("columns:", cols); sys.exit(1)
g = f"{gx}" + (f"*{gy}*{gz}" if gy and gz else "")
rows = c.execute(f"select name, {g} as grid, count(*), sum(end-start)/1e6, avg(end-start)/1e3 from kernels group by name, grid order by 4 desc").fetchall()
print(f"{'ms/step':>8} {'calls/step':>10} {'avg us':>9} {'blocks':>8}  kernel")
for name, grid, n, tot, avg in rows[:60]:
    nm = clean(name)
    if sub and sub not in nm:
        continue
    print(f"{tot / nsteps:8.3f} {n / nsteps:10.1f} {avg:9.1f} {int(grid) // 256:8d}  {nm}")
