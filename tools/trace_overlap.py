#!/usr/bin/env python
"""tools/trace_overlap.py trace.db [n] — round 6: where the three streams of the training step run relative to each other, out of a
rocprofv3 kernel trace (rocpd sqlite). A step window = from one `mse_loss_kernel` (end of the UNet forward, context stream) to the
next; for every stream: kernels, busy time, first start / last end relative to the window; and for the deferred tail (the stream that
runs `adamw_kernel`) the span from its first kernel after the loss to AdamW, next to the VAE encode's span on the context stream."""
import os
import sqlite3
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _names import clean  # noqa: E402

db = sys.argv[1]
which = int(sys.argv[2]) if len(sys.argv) > 2 else -2
c = sqlite3.connect(db).cursor()
rows = c.execute("select start, end, name, stream_id, grid_x*grid_y*grid_z from kernels order by start").fetchall()
rows = [(r[0], r[1], clean(r[2]), r[3], r[4]) for r in rows]
loss = [r for r in rows if "mse_loss" in r[2]]
lo, hi = loss[which - 1][0], loss[which][0]
win = [r for r in rows if lo <= r[0] < hi]
t0 = lo
print(f"window (loss -> next loss) {1e-6 * (hi - lo):.3f} ms, {len(win)} kernels")
for sid in sorted({r[3] for r in win}):
    ks = [r for r in win if r[3] == sid]
    print(f"  stream {sid}: {len(ks):5d} kernels, busy {1e-6 * sum(r[1] - r[0] for r in ks):7.3f} ms, span {1e-6 * (ks[0][0] - t0):7.3f} .. {1e-6 * (ks[-1][1] - t0):7.3f} ms"
          f"   first {ks[0][2][:40]} last {ks[-1][2][:40]}")
adam = [r for r in win if "adamw" in r[2]]
if adam:
    sid = adam[0][3]
    ks = [r for r in win if r[3] == sid and r[0] <= adam[-1][0]]
    # the tail = the run of kernels on AdamW's stream that ends with AdamW and starts after the last kernel of that stream before the UNet backward ended
    print(f"  AdamW on stream {sid} ends at {1e-6 * (adam[-1][1] - t0):.3f} ms")
    tail = [r for r in ks if "layernorm_bwd" in r[2] or "quickgelu_bwd" in r[2]]
    if tail:
        print(f"  CLIP backward (quickgelu / layernorm bwd kernels on that stream): {1e-6 * (tail[0][0] - t0):.3f} .. {1e-6 * (tail[-1][1] - t0):.3f} ms")
big = [r for r in win if r[4] >= 256 * 1024 and "gemm" in r[2]]
if big:
    print(f"  large conv launches (>= 1024 blocks: the VAE encode): {1e-6 * (big[0][0] - t0):.3f} .. {1e-6 * (big[-1][1] - t0):.3f} ms on stream(s) {sorted({r[3] for r in big})}")
iv = sorted((r[0], r[1]) for r in win)
busy, cs, ce = 0, iv[0][0], iv[0][1]
for s, e in iv[1:]:
    if s > ce:
        busy += ce - cs; cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
print(f"  union busy {1e-6 * busy:.3f} ms of {1e-6 * (hi - lo):.3f}")
