#!/usr/bin/env python
"""Per-dispatch means of rocprofv3 --pmc passes: tools/pmc_summary.py <dir> [<dir> …] — prints kernel, grid, counter means"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _names import clean  # noqa: E402
import collections
import csv
import glob

acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for d in sys.argv[1:]:
    for p in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(p)):
            n = clean(r["Kernel_Name"])
            if "gemm" not in n and "flash" not in n and not n.startswith("f2_"):
                continue
            key = (n, int(r["Grid_Size"]))
            a = acc[key][r["Counter_Name"]]
            a[0] += 1; a[1] += float(r["Counter_Value"])
for key in sorted(acc):
    print(f"== {key[0]} grid={key[1]}")
    for c, (n, s) in sorted(acc[key].items()):
        print(f"   {c:28s} {s / n:16.1f}   (n={n})")
