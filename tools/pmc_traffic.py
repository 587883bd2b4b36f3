#!/usr/bin/env python
"""HBM traffic of the hot path's kernels from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a pass on
gfx950: /opt/skills/guides/MI355X_MICROARCH.md "rocprofv3 PMC slots").

    tools/pmc_traffic.py <fetch_dir> <write_dir> out.json

Reads the *_counter_collection.csv of both passes, averages per kernel symbol and per (symbol, grid size) so that one
shape can be singled out. FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB and stored here as reported (bytes);
see the "note" field for the gfx950 correction the guide prescribes.
"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _names import clean  # noqa: E402
import collections
import csv
import glob
import json


def load(d, counter):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for p in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(p)):
            if r["Counter_Name"] != counter:
                continue
            n = clean(r["Kernel_Name"])
            n = n.replace("(cb::_anonymous_namespace_)::", "")
            if not ("gemm_" in n or "flash" in n or n.startswith("gn_") or "layernorm" in n or "splitk" in n):
                continue
            key = (n, int(r["Grid_Size"]))
            acc[key][0] += 1
            acc[key][1] += float(r["Counter_Value"])
    return acc


def main():
    fd, wd, out = sys.argv[1:4]
    f = load(fd, "FETCH_SIZE")
    w = load(wd, "WRITE_SIZE")
    rows, sym = [], collections.defaultdict(lambda: [0, 0.0, 0, 0.0])
    for key in sorted(set(f) | set(w)):
        nf, sf = f.get(key, [0, 0.0])
        nw, sw = w.get(key, [0, 0.0])
        rows.append({"kernel": key[0], "grid": key[1], "launches": max(nf, nw),
                     "fetch_bytes_per_launch": 1024.0 * sf / nf if nf else None,
                     "write_bytes_per_launch": 1024.0 * sw / nw if nw else None})
        a = sym[key[0]]
        a[0] += nf; a[1] += 1024.0 * sf; a[2] += nw; a[3] += 1024.0 * sw
    per_symbol = {k: {"launches": max(a[0], a[2]), "fetch_bytes_per_launch": a[1] / a[0] if a[0] else None,
                      "write_bytes_per_launch": a[3] / a[2] if a[2] else None} for k, a in sym.items()}
    note = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), KiB -> bytes, AS REPORTED. Guide correction for gfx950 "
            "(MI355X_MICROARCH.md, HBM / rocprofv3 section): FETCH_SIZE counts a 128-B request of a wide coalesced read or a "
            "`buffer_load ... lds` as 64 B, so the LDS-DMA GEMM / conv kernels (16-B lanes, 128-B rows) read 2 x FETCH_SIZE; "
            "bench.py applies that factor to `roofline.traffic` and keeps the raw counters in `traffic_detail`. FETCH_SIZE counts "
            "L2 misses that the Infinity Cache may still serve, so it is an upper bound of the HBM reads proper.")
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from celebbasis_amd.build import source_digest
    json.dump({"stamp": source_digest(), "note": note, "per_symbol": per_symbol, "per_grid": rows}, open(out, "w"), indent=1)
    for k, v in sorted(per_symbol.items(), key=lambda kv: -(kv[1]["fetch_bytes_per_launch"] or 0) * kv[1]["launches"])[:12]:
        print(k, v)


if __name__ == "__main__":
    main()
