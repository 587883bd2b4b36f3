#!/usr/bin/env python
"""tools/pmc_p8.py — round 6: fixed workloads for the PMC passes of the phase-pipelined bf16 tiles (cb_gemm_bf16_p8.hip) next to the
128 x 128 two-barrier kernel they replace, on the VAE's three large config-3 convolutions and a square GEMM:
   bash tools/gpurun.sh pmcsq r06_p8 tools/pmc_p8.py     (SQ busy / waits, instruction mix + LDS bank conflicts, L2 hit + TA busy)
Plans are pinned with cb_debug_force_plan: 128 x 128 first, then the tile the plan table picks for the shape."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from celebbasis_amd import lib  # noqa: E402

L = lib.load()
ctx = lib.Context(dtype="bf16")
DEV, BF = "cuda:0", torch.bfloat16
g = torch.Generator().manual_seed(0)
for (N, H, W, Ci, Co, tile) in [(4, 512, 512, 128, 128, (512, 128)), (4, 256, 256, 256, 256, (256, 256)), (4, 128, 128, 512, 512, (256, 256))]:
    x = torch.randn(N, H, W, Ci, generator=g).to(BF).to(DEV)
    w = (torch.randn(Co, Ci, 3, 3, generator=g) * (Ci * 9) ** -0.5).to(DEV)
    wp = torch.empty(Co * 9 * Ci, dtype=BF, device=DEV)
    lib.check(L.cb_pack_conv_weight(ctx.h, lib.ptr(w), lib.ptr(wp, BF), Co, Ci, 3, 0))
    b = torch.zeros(Co, device=DEV)
    y = torch.empty(N, H, W, Co, device=DEV, dtype=BF)
    for plan in ((128, 128), tile):
        lib.check(L.cb_debug_force_plan(plan[0], plan[1], 1))
        for _ in range(3):
            lib.check(L.cb_conv2d_fwd(ctx.h, lib.ptr(x, BF), lib.ptr(wp, BF), lib.ptr(b), None, None, lib.ptr(y, BF), N, H, W, Ci, Co, 3, 0))
        torch.cuda.synchronize()
M = Nn = K = 4096
A = (torch.rand(M, K, generator=g) * 2 - 1).to(BF).to(DEV); B = (torch.rand(Nn, K, generator=g) * 2 - 1).to(BF).to(DEV)
Cc = torch.empty(M, Nn, device=DEV, dtype=BF)
d = lib.GemmDesc()
d.A, d.B, d.C = A.data_ptr(), B.data_ptr(), Cc.data_ptr()
d.M, d.N, d.K = M, Nn, K
d.lda = d.ldb = K; d.ldc = Nn; d.batch = 1; d.alpha = 1.0; d.dtype = lib.DTYPE_BF16
for plan in ((128, 128), (256, 256)):
    lib.check(L.cb_debug_force_plan(plan[0], plan[1], 1))
    for _ in range(3):
        lib.check(L.cb_gemm(ctx.h, C.byref(d)))
    torch.cuda.synchronize()
lib.check(L.cb_debug_force_plan(0, 0, 0))
del ctx
