#!/usr/bin/env python
"""BASELINE config 5, end to end in one run: identities sharded over the ranks; each identity is personalised (bs=2 training
steps through cb_trainer_step) and then sampled (eval-mode manager -> CLIP -> DDIM-50 with guidance 10, n_samples 8 ->
VAE decode), exactly the two halves `01_start_train.sh` + `02_start_test.sh` run one after the other in the reference.
Synthetic weights / faces (no checkpoints or datasets offline). One JSON line with the end-to-end wall-clock.

    python tools/run_config5.py --identities 4 --train-steps 100                                   # 1 GPU, scaled down
    python tools/run_config5.py --gpus 8 --identities 32                                           # 8 ranks (launches itself under torch.distributed.run)
    python tools/run_config5.py --identities 1 --train-steps 3 --ddim-steps 2 --n-samples 2 --dump-first f.pt   # + the oracle replay in tests/
    python tools/run_config5.py --gpus 2 --identities 5 --dry                                       # no GPU: launcher / sharding / merge only
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=0, help="ranks of this node; > 1 without a launcher environment: re-executes itself under "
                                                         "torch.distributed.run (same rule as bench.py); 0 = take WORLD_SIZE as it is")
    ap.add_argument("--identities", type=int, default=4, help="total identities of the job (config 5: 32)")
    ap.add_argument("--train-steps", type=int, default=800, help="steps per identity (aigc_id.yaml:135 max_steps 800)")
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--ddim-steps", type=int, default=50)
    ap.add_argument("--n-samples", type=int, default=8)
    ap.add_argument("--scale", type=float, default=10.0)
    ap.add_argument("--precision", default="f32", choices=["f32", "bf16"])
    ap.add_argument("--dry", action="store_true", help="no GPU: dry contexts + gloo — launcher, identity sharding, per-identity control flow, "
                                                        "checkpoint merge and the JSON line only (computes nothing); used by the CPU test")
    ap.add_argument("--dump-first", type=str, default=None,
                    help="write the FIRST identity's learned coefficient row, its decoded images' statistics and the replay description "
                         "(weight seed, batch seeds, lr, steps) to this .pt file: tests/test_gpu_models.py replays exactly that training on "
                         "the CPU oracle and holds the coefficients to the precision's bar (the checker lives in tests/, not here)")
    a = ap.parse_args()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:            # `python tools/run_config5.py --gpus 8`: become the launcher
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        os.execvp(cmd[0], cmd)
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus and world != a.gpus:
        raise SystemExit(f"run_config5.py --gpus {a.gpus} was started with WORLD_SIZE={world}")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    if a.dry:
        dev = torch.device("cpu")
        if world > 1:
            dist.init_process_group("gloo")
    else:
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
        if world > 1:
            dist.init_process_group("nccl", device_id=dev)
    from celebbasis_amd import lib, synthetic, sharding
    if a.dry:
        return dry_run(a, rank, world, dev, dist, lib, synthetic, sharding)
    from celebbasis_amd.ldm.models.diffusion.ddim import ddim_tables, ddim_timesteps
    from celebbasis_amd.ldm.models.diffusion.ddpm import make_beta_schedule_linear

    shapes = dict(synthetic.trainer_shapes()); shapes.update(synthetic.decoder_shapes())
    state = synthetic.synthetic_state(shapes, seed=0)
    ctx = lib.Context(dtype=a.precision)
    wt = lib.WeightTable(state)
    tr = lib.Trainer(ctx, wt)
    unet, clip = tr.unet(), tr.clip()                    # views on the trainer's weights: one copy of SD v1-4 per GPU
    dec = lib.VAEDecoder(ctx, wt, lib.vae_cfg(), prefix="first_stage_model.")
    basis = state["celeb_embeddings"].to(dev).contiguous()
    es, K, W = 2, 512, 768
    mlp_w0, mlp_b0 = state["embedding_manager.meta_id_net.stylegan_mlp.net.0.weight"], state["embedding_manager.meta_id_net.stylegan_mlp.net.0.bias"]
    acp = torch.tensor(np.cumprod(1.0 - make_beta_schedule_linear(1000, 0.00085, 0.012)), dtype=torch.float32)
    ts = ddim_timesteps("uniform", a.ddim_steps, 1000)
    tb = ddim_tables(acp, ts, 0.0)
    L = lib.load()
    mine = sharding.identities_for_rank(a.identities, rank, world)
    B = a.batch
    batches = [synthetic.synthetic_batch(B, step=s, device=dev, identity=0, seed=23 + rank) for s in range(4)]
    prompt = torch.full((a.n_samples, 77), 49407, dtype=torch.int64); prompt[:, :len(synthetic.PROMPT_IDS)] = torch.tensor(synthetic.PROMPT_IDS)
    empty = torch.full((a.n_samples, 77), 49407, dtype=torch.int64); empty[:, 0] = 49406
    prompt, empty = prompt.to(dev), empty.to(dev)
    zeros = lambda n: torch.zeros(n)
    coefs, t_train, t_sample, finite, checked = [], 0.0, 0.0, True, None
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for ident in mine:
        slot = ident % 10
        # a fresh coefficient MLP + optimiser per identity (each identity is its own optimisation problem, SURVEY §8 e)
        tr.set_state("mlp.weight", mlp_w0); tr.set_state("mlp.bias", mlp_b0); tr.set_state("step", torch.zeros(1))
        for n in ("adam.m_weight", "adam.v_weight"):
            tr.set_state(n, zeros(es * K * 512))
        for n in ("adam.m_bias", "adam.v_bias"):
            tr.set_state(n, zeros(es * K))
        ta = time.perf_counter()
        for s in range(a.train_steps):
            b = batches[s % len(batches)]
            ids = torch.full((B, 2), slot, dtype=torch.int64, device=dev)
            tr.step(b["image"], b["faces"], ids, b["tokens"], b["t"], b["noise"], b["post_noise"], 5e-3 * B)
        coef = tr.get_state("id_coefficients").reshape(10, es * K)[slot].clone()
        tb_ = time.perf_counter()
        first = rank == 0 and ident == mine[0]
        # ---- sampling with the learned coefficients (test_mode 'coefficient', embedding_manager.py:464-475)
        c_dev = coef.reshape(es, K).to(dev).contiguous()
        z = torch.empty(es, W, device=dev)
        for e in range(es):
            d = lib.GemmDesc()
            d.A, d.B, d.C = c_dev[e].data_ptr(), basis[e, 1:].data_ptr(), z[e].data_ptr()
            d.M, d.N, d.K = 1, W, K
            d.lda, d.ldb, d.ldc = K, W, W
            d.a_layout, d.b_layout, d.batch = 0, 1, 1
            d.bias = basis[e, 0].data_ptr(); d.alpha, d.beta = 1.0, 0.0
            lib.check(L.cb_gemm(ctx.h, C.byref(d)))
        emb = clip.embed_tokens(prompt)
        out = torch.empty_like(emb); mp = torch.empty(a.n_samples * 77, dtype=torch.int32, device=dev)
        zrep = z.unsqueeze(0).repeat(a.n_samples, 1, 1).contiguous()
        lib.check(L.cb_row_surgery(ctx.h, lib.ptr(emb), lib.ptr(zrep), lib.ptr(prompt, torch.int64), lib.ptr(out), C.c_void_p(mp.data_ptr()),
                                   a.n_samples, 77, W, es, synthetic.SKS))
        cond = clip.forward(out)
        uncond = clip.forward(clip.embed_tokens(empty))
        x_T = torch.randn(a.n_samples, 4, 64, 64, device=dev)
        lat, _ = unet.ddim_sample(x_T, cond, uncond, ts, tb["ddim_alphas"], tb["ddim_alphas_prev"], tb["ddim_sigmas"],
                                  tb["ddim_sqrt_one_minus_alphas"], scale=a.scale)
        img = dec.decode(lat, 0.18215, to01=True)
        torch.cuda.synchronize()
        tc = time.perf_counter()
        finite = finite and bool(torch.isfinite(img).all()) and bool(torch.isfinite(coef).all())
        if a.dump_first and first:
            torch.save(dict(coef=coef.cpu(), slot=slot, steps=a.train_steps, batch=B, lr=5e-3 * B, weight_seed=0, batch_seed=23 + rank,
                            n_batches=len(batches), precision=a.precision, image_mean=float(img.mean()), image_std=float(img.std()),
                            image_shape=list(img.shape)), a.dump_first)
            checked = a.dump_first
        coefs.append(coef); t_train += tb_ - ta; t_sample += tc - tb_
    merged = sharding.gather_identity_blocks(mine, torch.stack(coefs) if coefs else torch.zeros(0, es * K), a.identities, device=dev)
    torch.cuda.synchronize()
    elapsed = torch.tensor([time.perf_counter() - t0], device=dev)
    if world > 1:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    if rank == 0:
        n_merged = int((merged.abs().sum(dim=1) > 0).sum().item())
        assert n_merged == a.identities, f"checkpoint merge returned {n_merged} of {a.identities} identities"
        el = float(elapsed.item())
        print(json.dumps({"metric": "config 5 end-to-end: personalise + DDIM sample, identities/sec", "value": round(a.identities / el, 5),
                          "unit": "identities/s", "n_gpus": world, "identities": a.identities, "train_steps_per_identity": a.train_steps,
                          "batch": B, "ddim_steps": a.ddim_steps, "n_samples": a.n_samples, "cfg_scale": a.scale, "wall_s": round(el, 2),
                          "rank0_train_s": round(t_train, 2), "rank0_sample_s": round(t_sample, 2),
                          "rank0_train_steps_per_s": round(len(mine) * a.train_steps / max(t_train, 1e-9), 2),
                          "merged_identities": int((merged.abs().sum(dim=1) > 0).sum().item()), "finite": finite, "dtype": a.precision,
                          "first_identity_dump": checked, "data": "synthetic"}))
    if world > 1:
        dist.destroy_process_group()


def dry_run(a, rank, world, dev, dist, lib, synthetic, sharding):
    """--dry: the job's host logic on dry contexts (shape checks + accounting, no device) with the gloo backend"""
    shapes = dict(synthetic.trainer_shapes()); shapes.update(synthetic.decoder_shapes())
    ctx = lib.Context(device=-1)
    tr = lib.Trainer(ctx, lib.WeightTable(shapes))
    mine = sharding.identities_for_rank(a.identities, rank, world)
    es, K = 2, 512
    coefs = []
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for ident in mine:
        for _ in range(min(a.train_steps, 2)):
            tr.step(a.batch, None, None, None, None, None, None, 5e-3 * a.batch)       # a dry context plans and accounts, computes nothing
        coefs.append(torch.full((es * K,), float(ident + 1)))                          # stand-in block that identifies its owner
    merged = sharding.gather_identity_blocks(mine, torch.stack(coefs) if coefs else torch.zeros(0, es * K), a.identities, device=dev)
    elapsed = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    if rank == 0:
        for i in range(a.identities):
            assert torch.all(merged[i] == i + 1), f"checkpoint merge: identity {i} came back as {merged[i][:2].tolist()}"
        print(json.dumps({"metric": "config 5 end-to-end: personalise + DDIM sample, identities/sec", "value": None, "unit": "identities/s",
                          "n_gpus": world, "identities": a.identities, "train_steps_per_identity": a.train_steps, "batch": a.batch,
                          "merged_identities": a.identities, "identities_per_rank": [len(sharding.identities_for_rank(a.identities, r, world)) for r in range(world)],
                          "dry": True, "dtype": a.precision, "data": "synthetic"}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
