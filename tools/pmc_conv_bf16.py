#!/usr/bin/env python
"""Fixed bf16 conv / GEMM workloads (the dominant config-3 kernel on three representative shapes) for PMC collection:
   rocprofv3 --kernel-trace --pmc <counters> -d out -- python tools/pmc_conv_bf16.py   (variant: CB_BF16_GEMM)"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from celebbasis_amd import lib  # noqa: E402

L = lib.load()
ctx = lib.Context(dtype="bf16")
DEV, BF = "cuda:0", torch.bfloat16
for (N, H, W, Ci, Co) in [(4, 512, 512, 128, 128), (4, 64, 64, 320, 320), (4, 64, 64, 512, 512)]:
    x = torch.randn(N, H, W, Ci, device=DEV).to(BF); w = (torch.randn(Co, 9 * Ci, device=DEV) * 0.01).to(BF); b = torch.zeros(Co, device=DEV)
    y = torch.empty(N, H, W, Co, device=DEV, dtype=BF)
    for _ in range(4):
        lib.check(L.cb_conv2d_fwd(ctx.h, lib.ptr(x, BF), lib.ptr(w, BF), lib.ptr(b), None, None, lib.ptr(y, BF), N, H, W, Ci, Co, 3, 0))
    torch.cuda.synchronize()
M, Nn, K = 16384, 2560, 320
A = torch.randn(M, K, device=DEV).to(BF); B = torch.randn(Nn, K, device=DEV).to(BF); Cc = torch.empty(M, Nn, device=DEV, dtype=BF)
d = lib.GemmDesc()
d.A, d.B, d.C = A.data_ptr(), B.data_ptr(), Cc.data_ptr()
d.M, d.N, d.K = M, Nn, K
d.lda = d.ldb = K; d.ldc = Nn; d.batch = 1; d.alpha = 1.0; d.dtype = lib.DTYPE_BF16
for _ in range(4):
    lib.check(L.cb_gemm(ctx.h, C.byref(d)))
torch.cuda.synchronize()
del ctx
