#!/usr/bin/env python
"""tools/gemm_histogram.py — the GEMM-class launch histogram of one training step WITHOUT a GPU: a dry context runs the step's host
code with CB_GEMM_LOG set (cb_gemm_glds.hip::gemm_log writes one "<plan_key> <BM> <BN> <split>" line per problem handed to the LDS-DMA
kernels), and the lines are joined with the measured times of the offline autotune records.

    python tools/gemm_histogram.py [profiles/r04_autotune.txt]       top shapes by launches x measured time, totals, split-K count
"""
import collections
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    recs = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r04_autotune.txt")
    rec = {}
    for l in open(recs):
        p = l.split()
        if len(p) >= 10 and p[5] == "|":
            rec[p[0]] = ((int(p[1]), int(p[2]), int(p[3])), float(p[4]), float(p[9]))      # best plan, its us, the model plan's us
    with tempfile.TemporaryDirectory() as d:
        log = os.path.join(d, "gemm.log")
        subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                       env=dict(os.environ, CB_GEMM_LOG=log), check=True, capture_output=True, cwd=ROOT)
        cnt, plan = collections.Counter(), {}
        for l in open(log):
            if l.strip():
                k, bm, bn, sp = l.split()
                cnt[k] += 1; plan[k] = (int(bm), int(bn), int(sp))
    rows, total, nsplit, n80 = [], 0.0, 0, 0
    for k, c in cnt.items():
        us = None
        if k in rec:
            us = rec[k][1] if plan[k] == rec[k][0] else rec[k][2]
            total += c * us
        nsplit += c if plan[k][2] > 1 else 0
        n80 += c if plan[k][1] == 80 else 0
        f = k.split("|")
        rows.append((c * (us or 0.0), c, f"{'conv' if f[2] == '2' else 'gemm'} M={f[3]} N={f[4]} K={f[5]}", plan[k], us))
    rows.sort(reverse=True)
    for t, c, name, pl, us in rows[:40]:
        print(f"{t:9.1f} us  n={c:3d}  {name:34s} {pl[0]:3d}x{pl[1]:<3d}/{pl[2]:<2d} {'' if us is None else f'{us:7.1f} us'}")
    print(f"{sum(cnt.values())} LDS-DMA GEMM-class launches per step, {len(cnt)} shapes; {total / 1e3:.2f} ms by the records; "
          f"{nsplit} with a split-K reduce; {n80} on the 80-column tiles")


if __name__ == "__main__":
    main()
