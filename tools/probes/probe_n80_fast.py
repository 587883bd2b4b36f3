#!/usr/bin/env python
"""probe: the FAST (scalar-offset) pieces of gemm_glds80_kernel on Linear shapes with K = 640 — timing by (M, N, K, ld, plan)"""
import ctypes as C
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from celebbasis_amd import lib  # noqa: E402
L = lib.load(); ctx = lib.Context(); DEV = "cuda:0"


def timeit(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for (M, N, K, ld) in [(1024, 640, 640, 640), (1024, 640, 640, 672), (2048, 640, 640, 640), (1024, 320, 640, 640), (1024, 640, 320, 320), (1024, 640, 1280, 1280),
                      (1024, 640, 960, 960), (1024, 1280, 640, 640), (512, 640, 640, 640)]:
    A = torch.randn(M, ld, device=DEV); B = torch.randn(N, ld, device=DEV) * K ** -0.5; Cc = torch.empty(M, N, device=DEV)
    d = lib.GemmDesc()
    d.A, d.B, d.C = A.data_ptr(), B.data_ptr(), Cc.data_ptr()
    d.M, d.N, d.K = M, N, K
    d.lda = d.ldb = ld; d.ldc = N; d.batch = 1; d.alpha = 1.0
    fn = lambda: lib.check(L.cb_gemm(ctx.h, C.byref(d)))
    row = []
    for (bm, bn, sp) in [(64, 80, 1), (32, 80, 1), (16, 80, 1)]:
        lib.check(L.cb_debug_force_plan(bm, bn, sp))
        row.append(f"{bm}x{bn}/{sp}: {timeit(fn):7.2f}")
    print(f"M={M} N={N} K={K} ld={ld}  " + "  ".join(row), flush=True)
lib.check(L.cb_debug_force_plan(0, 0, 0))
