#!/usr/bin/env python
"""Sampling path timing (SURVEY §8 A10 / BASELINE config 5 second half): DDIM-S with classifier-free guidance + VAE decode at
SD v1-4 sizes, fp32, synthetic weights. Prints one JSON line: seconds per image and the UNet-pass rate.
    python tools/bench_sampler.py [--n 1] [--steps 50] [--scale 10]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from celebbasis_amd import lib, synthetic  # noqa: E402
from celebbasis_amd.ldm.models.diffusion.ddim import ddim_tables, ddim_timesteps  # noqa: E402
from celebbasis_amd.ldm.models.diffusion.ddpm import make_beta_schedule_linear  # noqa: E402
import numpy as np  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1)
ap.add_argument("--steps", type=int, default=50)
ap.add_argument("--scale", type=float, default=10.0)
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"], help="UNet / VAE element type (bf16 = the MI355X counterpart of the reference's fp16 autocast sampling)")
a = ap.parse_args()
dev = "cuda:0"
ctx = lib.Context(dtype=a.dtype)
shapes = {k: v for k, v in synthetic.trainer_shapes().items() if k.startswith("model.diffusion_model.")}
shapes.update(synthetic.decoder_shapes())
W = synthetic.synthetic_state(shapes, seed=0, device=dev)
wt = lib.WeightTable(W)
u = lib.UNet(ctx, wt, lib.unet_cfg(), prefix="model.diffusion_model.")
d = lib.VAEDecoder(ctx, wt, lib.vae_cfg(), prefix="first_stage_model.")
acp = torch.tensor(np.cumprod(1.0 - make_beta_schedule_linear(1000, 0.00085, 0.012)), dtype=torch.float32)
ts = ddim_timesteps("uniform", a.steps, 1000)
tb = ddim_tables(acp, ts, 0.0)
g = torch.Generator(device="cpu").manual_seed(1)
x_T = torch.randn(a.n, 4, 64, 64, generator=g).to(dev); c = torch.randn(a.n, 77, 768, generator=g).to(dev); uc = torch.randn(a.n, 77, 768, generator=g).to(dev)
best = None
for r in range(a.reps + 1):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    z, _ = u.ddim_sample(x_T, c, uc, ts, tb["ddim_alphas"], tb["ddim_alphas_prev"], tb["ddim_sigmas"], tb["ddim_sqrt_one_minus_alphas"], scale=a.scale)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    img = d.decode(z, 0.18215, to01=True)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    if r > 0 and (best is None or t2 - t0 < best[0]):
        best = (t2 - t0, t1 - t0, t2 - t1)
print(json.dumps({"metric": f"DDIM sampling seconds per image @ SD v1-4 512x512 {a.dtype}", "dtype": a.dtype, "n_samples": a.n, "ddim_steps": a.steps, "cfg_scale": a.scale,
                  "sec_per_batch": round(best[0], 4), "sec_per_image": round(best[0] / a.n, 4), "ddim_loop_s": round(best[1], 4),
                  "vae_decode_s": round(best[2], 4), "unet_passes_per_s": round(a.steps * 2 * a.n / best[1], 2),
                  "unet_tflops": round(a.steps * 2 * a.n * 0.80327 / best[1], 2), "finite": bool(torch.isfinite(img).all())}))
