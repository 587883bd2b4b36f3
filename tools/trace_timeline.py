#!/usr/bin/env python
"""tools/trace_timeline.py trace.db — the timeline of ONE steady-state training step out of a rocprofv3 kernel trace (rocpd sqlite):
per stream the busy time, the union over streams, the idle gaps, and where the streams wait for each other (fork / join of the identity
branch, the serial tail). Steps are delimited by the AdamW kernel."""
import sqlite3
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _names import clean  # noqa: E402

db = sys.argv[1]
which = int(sys.argv[2]) if len(sys.argv) > 2 else -3
c = sqlite3.connect(db).cursor()
rows = c.execute("select start, end, name, stream_id, grid_x*grid_y*grid_z, workgroup_x from kernels order by start").fetchall()
ends = [r[1] for r in rows if "adamw" in r[2]]
# two adamw launches per step (weight, bias): step boundary = every second one
bounds = ends[1::2]
lo, hi = bounds[which - 1], bounds[which]
step = [r for r in rows if lo < r[0] <= hi]
t0 = min(r[0] for r in step); t1 = max(r[1] for r in step)
print(f"step wall {1e-6 * (t1 - t0):.3f} ms, {len(step)} kernels")
streams = sorted({r[3] for r in step})
for sid in streams:
    ks = [r for r in step if r[3] == sid]
    print(f"  stream {sid}: {len(ks)} kernels, busy {1e-6 * sum(r[1] - r[0] for r in ks):.3f} ms, span {1e-6 * (ks[0][0] - t0):.3f} .. {1e-6 * (ks[-1][1] - t0):.3f} ms")
# union busy
iv = sorted((r[0], r[1]) for r in step)
busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
gaps = []
for s, e in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; gaps.append((cur_e, s)); cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"union busy {1e-6 * busy:.3f} ms, idle {1e-6 * (t1 - t0 - busy):.3f} ms in {len(gaps)} gaps (median {1e-3 * sorted(g[1] - g[0] for g in gaps)[len(gaps) // 2]:.2f} us, "
      f"max {1e-3 * max(g[1] - g[0] for g in gaps):.1f} us)")
big = sorted(gaps, key=lambda g: g[0] - g[1])[:8]
for g in sorted(big):
    before = max((r for r in step if r[1] <= g[0]), key=lambda r: r[1])
    after = min((r for r in step if r[0] >= g[1]), key=lambda r: r[0])
    print(f"   gap {1e-3 * (g[1] - g[0]):7.1f} us at {1e-6 * (g[0] - t0):7.3f} ms: {clean(before[2])[:50]} -> {clean(after[2])[:50]}")
# small-kernel time: kernels with fewer than 256 workgroups (cannot fill the chip) while nothing else runs
def overlaps(r):
    return any(o is not r and o[0] < r[1] and o[1] > r[0] for o in step)
lowocc = [r for r in step if r[4] // max(r[5], 1) < 256]
alone = [r for r in lowocc if not overlaps(r)]
print(f"kernels with < 256 workgroups: {len(lowocc)} ({1e-6 * sum(r[1] - r[0] for r in lowocc):.3f} ms); of them running ALONE on the chip: {len(alone)} ({1e-6 * sum(r[1] - r[0] for r in alone):.3f} ms)")
# phase marks
def first(sub): return next((r for r in step if sub in r[2]), None)
def last(sub):
    xs = [r for r in step if sub in r[2]]
    return xs[-1] if xs else None
for tag, sub in (("q_sample (UNet start)", "qsample"), ("mse loss", "mse_loss"), ("row_surgery_bwd (tail start)", "row_surgery_bwd")):
    r = first(sub)
    if r: print(f"  {tag}: {1e-6 * (r[0] - t0):.3f} ms")
