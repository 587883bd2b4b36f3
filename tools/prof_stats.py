#!/usr/bin/env python
"""Aggregate a rocprofv3 (rocpd sqlite) kernel trace into a per-kernel stats CSV: tools/prof_stats.py db out.csv nsteps"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _names import clean  # noqa: E402
import sqlite3

db, out, nsteps = sys.argv[1], sys.argv[2], int(sys.argv[3])
c = sqlite3.connect(db).cursor()
rows = c.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 "
                 "from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from celebbasis_amd.build import source_digest  # noqa: E402
stamp = source_digest()
lines = [f"# stamp src_sha={stamp['src_sha']} plan_table_sha={stamp['plan_table_sha']} nsteps={nsteps}",
         f"# rocprofv3 --kernel-trace; {nsteps} training steps in the trace; total kernel time {tot:.1f} ms = {tot / nsteps:.2f} ms/step",
         "name,calls,total_ms,ms_per_step,avg_us,min_us,max_us,pct"]
for r in rows:
    n = clean(r[0])
    lines.append(f'"{n}",{r[1]},{r[2]:.3f},{r[2] / nsteps:.3f},{r[3]:.2f},{r[4]:.2f},{r[5]:.2f},{100 * r[2] / tot:.2f}')
open(out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:32]))
