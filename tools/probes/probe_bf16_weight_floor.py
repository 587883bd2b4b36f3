#!/usr/bin/env python
"""tools/probes/probe_bf16_weight_floor.py — CPU only (round 6): how far do the learned coefficients move when NOTHING but the frozen UNet /
VAE weights is rounded to bf16? Config 3 ("bf16 UNet + fp32 coeff grad") holds those weights in bf16 by definition, so this distance — fp32
arithmetic everywhere, the product's own 40-step mini replay stream, the fp32 oracle against itself — is a floor under ANY implementation
of config 3, whatever it does about activation rounding (fp32 residual stream included). It is compared with north_star's 1e-3 (stated for
config 2) and with what the bf16 path measures on the GPU (profiles/r05_bf16_mini40_summation_order_samples.json: 1.9e-3 .. 2.9e-3).

    python tools/probes/probe_bf16_weight_floor.py [--steps 40] [--also-activations]      -> one JSON line (+ profiles/ file by hand)

--also-activations additionally rounds the OUTPUT of every UNet / VAE conv / linear module to bf16 (weights bf16, residual adds, norms and
softmax in fp32): an estimate of what an fp32-residual-stream variant would measure.
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--lr", type=float, default=5e-3)
    args = ap.parse_args()
    from oracle import cb_oracle as O
    import tests.test_gpu_models as T
    m, B = T.MINI, 2
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    shapes = O.trainer_param_shapes(m["unet"], m["vae"], m["clip"], m["fr"], es=m["es"], inner=m["inner"], max_ids=m["max_ids"])
    W = O.init_weights(shapes, seed=args.seed)
    lat = m["image"] // (2 ** (len(m["vae"]["ch_mult"]) - 1))
    batches = [T.make_batch(B, m["image"], lat, m["clip"]["vocab"], m["placeholder"], 100 + s, m["max_ids"]) for s in range(args.steps)]

    def run(Wx):
        st = O.TrainState(Wx)
        c0 = st.id_coef.clone()
        for b in batches:
            O.train_step(Wx, st, b, args.lr, m["unet"], m["vae"], m["clip"], m["fr"], es=m["es"], inner=m["inner"], placeholder=m["placeholder"])
        return st.id_coef.clone(), (st.id_coef - c0).norm(dim=1).max().item()

    ref, travelled = run(W)
    Wb = {k: (v.to(torch.bfloat16).to(torch.float32) if (k.startswith("model.diffusion_model.") or k.startswith("first_stage_model.")) and v.is_floating_point() else v.clone())
          for k, v in W.items()}
    got, _ = run(Wb)
    l2 = (got - ref).norm(dim=1).max().item()
    # a second, independent perturbation of the same size (round-to-nearest replaced by truncation of the low 16 bits): the distance is a
    # property of the perturbation's SIZE, not of one rounding pattern
    Wt = {k: ((v.view(torch.int32) & ~0xFFFF).view(torch.float32) if (k.startswith("model.diffusion_model.") or k.startswith("first_stage_model.")) and v.dtype == torch.float32 else v.clone())
          for k, v in W.items()}
    got_t, _ = run(Wt)
    l2_t = (got_t - ref).norm(dim=1).max().item()
    out = dict(workload="mini config, bs=2, %d steps, lr %g, seed %d (the stream of test_train_40_steps_mini_bf16_*)" % (args.steps, args.lr, args.seed),
               travelled=travelled, weights_rounded_to_bf16_only=l2, weights_truncated_to_bf16_only=l2_t, north_star_fp32_bar=1e-3,
               within_1e_3=bool(l2 <= 1e-3), gpu_bf16_path_samples="1.9e-3 .. 2.9e-3 (profiles/r05_bf16_mini40_summation_order_samples.json)")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
