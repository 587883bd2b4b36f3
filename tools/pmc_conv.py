#!/usr/bin/env python
"""One fixed conv workload (the dominant GEMM kernel on its biggest shapes) for PMC collection:
   rocprofv3 --kernel-trace --pmc <counters> -d out -- python tools/pmc_conv.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from celebbasis_amd import lib  # noqa: E402

L = lib.load()
ctx = lib.Context()
DEV = "cuda:0"
for (N, H, W, Ci, Co) in [(1, 256, 256, 256, 256), (1, 512, 512, 128, 128), (1, 128, 128, 512, 512), (1, 64, 64, 320, 320), (1, 16, 16, 1280, 1280)]:
    x = torch.randn(N, H, W, Ci, device=DEV); w = torch.randn(Co, 9 * Ci, device=DEV) * 0.01; b = torch.zeros(Co, device=DEV)
    y = torch.empty(N, H, W, Co, device=DEV)
    for _ in range(5):
        lib.check(L.cb_conv2d_fwd(ctx.h, lib.ptr(x), lib.ptr(w), lib.ptr(b), None, None, lib.ptr(y), N, H, W, Ci, Co, 3, 0))
    torch.cuda.synchronize()
