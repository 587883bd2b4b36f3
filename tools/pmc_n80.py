#!/usr/bin/env python
"""Fixed workload for the 80-column fp32 tiles (cb_gemm_glds.hip::gemm_glds80_kernel) against the 64x64 tile they replace, with the
plan pinned through cb_debug_force_plan: the 64x64-level ResBlock conv and Linear, the 32x32 and 16x16 level Linear.
    python tools/pmc_n80.py            HIP-event timings per (shape, plan)   -> gpurun_out/pmc_n80_times.json
    rocprofv3 --kernel-trace --pmc ... -- python tools/pmc_n80.py --reps 5    (tools/gpurun.sh pmcsq)
(A three-stage ring was measured slower in round 4 and removed: profiles/r04_n80_times_interleaved_st3.json.)"""
import argparse
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from celebbasis_amd import lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=30)
a = ap.parse_args()
L = lib.load()
ctx = lib.Context()
DEV = "cuda:0"
out = []


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.reps * 1e3


PLANS = [(64, 64, 1), (64, 64, 4), (64, 80, 1), (32, 80, 1), (16, 80, 1), (32, 80, 2)]
for (H, Ci, Co) in [(64, 320, 320), (32, 640, 640)]:
    x = torch.randn(1, H, H, Ci, device=DEV); w = torch.randn(Co, 9 * Ci, device=DEV) * 0.01; b = torch.zeros(Co, device=DEV)
    y = torch.empty(1, H, H, Co, device=DEV)
    fn = lambda: lib.check(L.cb_conv2d_fwd(ctx.h, lib.ptr(x), lib.ptr(w), lib.ptr(b), None, None, lib.ptr(y), 1, H, H, Ci, Co, 3, 0))
    for (bm, bn, sp) in PLANS:
        lib.check(L.cb_debug_force_plan(bm, bn, sp))
        us = timeit(fn)
        out.append(dict(op=f"conv {H}x{H} {Ci}->{Co}", plan=f"{bm}x{bn}/{sp}", us=round(us, 2), tflops=round(2.0 * H * H * Co * 9 * Ci / us / 1e6, 1)))
        print(out[-1], flush=True)
for (M, N, K) in [(4096, 320, 320), (1024, 640, 640), (256, 1280, 1280), (4096, 320, 1280)]:
    A = torch.randn(M, K, device=DEV); B = torch.randn(N, K, device=DEV) * K ** -0.5; Cc = torch.empty(M, N, device=DEV)
    d = lib.GemmDesc()
    d.A, d.B, d.C = A.data_ptr(), B.data_ptr(), Cc.data_ptr()
    d.M, d.N, d.K = M, N, K
    d.lda = d.ldb = K; d.ldc = N; d.batch = 1; d.alpha = 1.0
    fn = lambda: lib.check(L.cb_gemm(ctx.h, C.byref(d)))
    for (bm, bn, sp) in PLANS:
        if sp > 1 and K < 640:
            continue
        lib.check(L.cb_debug_force_plan(bm, bn, sp))
        us = timeit(fn)
        out.append(dict(op=f"gemm {M}x{N}x{K}", plan=f"{bm}x{bn}/{sp}", us=round(us, 2), tflops=round(2.0 * M * N * K / us / 1e6, 1)))
        print(out[-1], flush=True)
lib.check(L.cb_debug_force_plan(0, 0, 0))
os.makedirs("gpurun_out", exist_ok=True)
with open("gpurun_out/pmc_n80_times.json", "w") as f:
    json.dump(out, f, indent=1)
