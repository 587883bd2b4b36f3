#!/usr/bin/env python
"""SQ-counter workload for tools/gpurun.sh pmcsq: the 64x64-level conv and Linear under pinned plans (distinct kernel symbols / grids)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from celebbasis_amd import lib  # noqa: E402

L = lib.load()
ctx = lib.Context()
DEV = "cuda:0"
H, Ci, Co = 64, 320, 320
x = torch.randn(1, H, H, Ci, device=DEV); w = torch.randn(Co, 9 * Ci, device=DEV) * 0.01; b = torch.zeros(Co, device=DEV)
y = torch.empty(1, H, H, Co, device=DEV)
for (bm, bn, sp) in [(64, 80, 1), (32, 80, 1), (64, 64, 4), (128, 64, 3)]:
    lib.check(L.cb_debug_force_plan(bm, bn, sp))
    for _ in range(6):
        lib.check(L.cb_conv2d_fwd(ctx.h, lib.ptr(x), lib.ptr(w), lib.ptr(b), None, None, lib.ptr(y), 1, H, H, Ci, Co, 3, 0))
    torch.cuda.synchronize()
M, N, K = 4096, 320, 1280
A = torch.randn(M, K, device=DEV); B = torch.randn(N, K, device=DEV) * K ** -0.5; Cc = torch.empty(M, N, device=DEV)
d = lib.GemmDesc()
d.A, d.B, d.C = A.data_ptr(), B.data_ptr(), Cc.data_ptr()
d.M, d.N, d.K = M, N, K
d.lda = d.ldb = K; d.ldc = N; d.batch = 1; d.alpha = 1.0
for (bm, bn, sp) in [(64, 80, 1), (32, 80, 1), (64, 64, 1)]:
    lib.check(L.cb_debug_force_plan(bm, bn, sp))
    for _ in range(6):
        lib.check(L.cb_gemm(ctx.h, C.byref(d)))
    torch.cuda.synchronize()
lib.check(L.cb_debug_force_plan(0, 0, 0))
