#!/usr/bin/env python
"""Experiment: how much of the machine does the bs=1 UNet fwd+bwd leave idle? Runs UNet fwd+bwd (ctx A, stream A) and the VAE
encode (ctx B, stream B) back to back on ONE stream vs concurrently on TWO streams and reports ms per (UNet + VAE) pair."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from celebbasis_amd import lib, synthetic  # noqa: E402

dev = "cuda:0"
shapes = {k: v for k, v in synthetic.trainer_shapes().items() if k.startswith(("model.diffusion_model.", "first_stage_model."))}
W = synthetic.synthetic_state(shapes, seed=0, device=dev)
wt = lib.WeightTable(W)
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
with torch.cuda.stream(sA):
    ctxA = lib.Context(); u = lib.UNet(ctxA, wt, lib.unet_cfg(), prefix="model.diffusion_model.")
with torch.cuda.stream(sB):
    ctxB = lib.Context(); v = lib.VAE(ctxB, wt, lib.vae_cfg(), prefix="first_stage_model.")
with torch.cuda.stream(sA):
    ctxA2 = lib.Context(); v2 = lib.VAE(ctxA2, wt, lib.vae_cfg(), prefix="first_stage_model.")     # VAE on the UNet's stream (serial case)
g = torch.Generator().manual_seed(0)
x = torch.randn(1, 4, 64, 64, generator=g).to(dev); t = torch.tensor([500], device=dev); c = torch.randn(1, 77, 768, generator=g).to(dev)
dout = torch.randn(1, 4, 64, 64, generator=g).to(dev)
img = (torch.rand(1, 512, 512, 3, generator=g) * 2 - 1).to(dev); pn = torch.randn(1, 4, 64, 64, generator=g).to(dev)
torch.cuda.synchronize()


def run(mode, iters):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters):
        if mode in ("serial", "concurrent", "unet"):
            with torch.cuda.stream(sA):
                u.forward(x, t, c, need_grad=True); u.backward_context(dout)
        if mode == "serial":
            with torch.cuda.stream(sA):
                v2.encode(img, pn)
        if mode in ("concurrent", "vae"):
            with torch.cuda.stream(sB):
                v.encode(img, pn)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


for m in ("unet", "vae", "serial", "concurrent"):
    run(m, 3)
res = {m: round(min(run(m, 10) for _ in range(2)), 3) for m in ("unet", "vae", "serial", "concurrent")}
print(res, "reclaimed ms:", round(res["serial"] - res["concurrent"], 3))
