#!/usr/bin/env python
"""tools/probes/probe_p8_ksweep.py — round 6: time of the phase-pipelined bf16 tiles against K at fixed M x N (one / two / four blocks per
CU in sequence): the slope is the cost of a k-tile, the intercept the per-tile prologue + epilogue that nothing overlaps at one block per CU."""
import ctypes as C
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from celebbasis_amd import lib  # noqa: E402

DEV = torch.device("cuda:0")
BF = torch.bfloat16


def timeit(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def main():
    ctx = lib.Context(dtype="bf16")
    L = lib.load()
    g = torch.Generator().manual_seed(0)
    out = []
    for tile in ((256, 256), (512, 128), (128, 128)):
        lib.check(L.cb_debug_force_plan(tile[0], tile[1], 1))
        for (M, N) in ((4096, 4096), (16384, 4096), (65536, 512), (262144, 128)):
            row = dict(tile=tile, M=M, N=N, us={})
            for K in (64, 128, 256, 512, 1024, 2048, 4096):
                A = (torch.rand(M, K, generator=g) * 2 - 1).to(BF).to(DEV); B = (torch.rand(N, K, generator=g) * 2 - 1).to(BF).to(DEV)
                Cc = torch.empty(M, N, dtype=BF, device=DEV)
                d = lib.GemmDesc()
                d.A, d.B, d.C = A.data_ptr(), B.data_ptr(), Cc.data_ptr()
                d.M, d.N, d.K = M, N, K
                d.lda = d.ldb = K; d.ldc = N; d.batch = 1; d.alpha = 1.0; d.dtype = lib.DTYPE_BF16
                row["us"][K] = round(timeit(lambda: lib.check(L.cb_gemm(ctx.h, C.byref(d)))) * 1e6, 1)
                del A, B, Cc
            ks = sorted(row["us"])
            slope = (row["us"][4096] - row["us"][1024]) / ((4096 - 1024) / 64)
            row["us_per_ktile_launch"] = round(slope, 3)
            row["intercept_us"] = round(row["us"][1024] - slope * 16, 1)
            out.append(row)
            print(tile, M, N, row["us"], "slope/k-tile", row["us_per_ktile_launch"], "intercept", row["intercept_us"])
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "p8_ksweep.json"), "w"), indent=1)
    os._exit(0)


if __name__ == "__main__":
    main()
