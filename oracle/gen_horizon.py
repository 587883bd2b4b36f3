"""CPU ORACLE — TEST INFRASTRUCTURE ONLY. Long-horizon oracle trajectories at the full SD v1-4 size, computed ONCE on host cores
(≈10 s per sample and step) and committed as fixtures, so that the `-m gpu` horizon tests spend their time on the HIP path
instead of on the CPU oracle:

    python oracle/gen_horizon.py fp32   -> tests/golden/horizon_sd14_b1.pt   BASELINE config 2: bs=1, 100 optimiser steps
    python oracle/gen_horizon.py bf16   -> tests/golden/horizon_sd14_b4.pt   BASELINE config 3: bs=4, 20 optimiser steps (fp32 oracle)
    python oracle/gen_horizon.py fp32alt  -> horizon_sd14_b1_alt.pt   20 steps of `fp32` on 3 instead of 8 host threads (torch-CPU's convolutions
                                            do not change their summation order with the thread count: the two runs agree to 1.5e-6)
    python oracle/gen_horizon.py fp32pert -> horizon_sd14_b1_pert.pt  `fp32` with every step's MLP gradient perturbed by 4e-5 relative noise
    python oracle/gen_horizon.py fp32long -> horizon_sd14_b1_800.pt   `fp32` continued to 800 steps (config 2's own length; 3.5 h)
    python oracle/gen_horizon.py fp32photo -> horizon_sd14_b1_photo.pt 800 steps of the same stream with low-pass ("photo-like") face crops (2.5 h)
    python oracle/gen_horizon.py fp32photo2 / fp32photo3 -> horizon_sd14_b1_photo{2,3}.pt  the same with other frozen weights and another batch stream
    python oracle/gen_horizon.py fp32noise2 -> horizon_sd14_b1_noise2.pt  the default (white-noise) stream with other frozen weights and another batch stream
    python oracle/gen_horizon.py fp32photo2head / fp32photo3head / fp32noise2head -> horizon_sd14_b1_{photo2,photo3,noise2}_head.pt  their first 120 steps
                                            with the per-step coefficients kept (30 min each)

The trajectory is `oracle.cb_oracle.train_step` (pinned against the reference by oracle/gen_golden.py::gen_step) driven over
`oracle.replay.make_batch(seed = 100 + step)` with `init_weights(seed)`; the first steps of the fp32 run are the ones
tests/test_gpu_models.py::test_train_2_steps_sd14_live_oracle recomputes live, which ties the fixture to the live oracle.
Per step the file holds the loss, the coefficients of every face in the batch (meta_net.py:289, [2B, es, inner]) and the EMA
dictionary rows of the identities in the batch (embedding_manager.py:484-489); the MLP gradient is kept for a few steps on every
16th row (a 32 768-element sample: the full 2 MB per step would not be a small fixture)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import cb_oracle as O                      # noqa: E402
from oracle.replay import make_batch                   # noqa: E402

RUNS = dict(fp32=dict(B=1, steps=100, lr=5e-3, seed=8, out="horizon_sd14_b1.pt", grad_steps=(1, 2, 5, 10, 25, 50, 100)),
            # the SAME oracle run on a different number of host threads, i.e. with another fp32 summation order inside torch's CPU GEMMs /
            # convolutions: how far two correct fp32 evaluations of the reference's arithmetic drift apart under AdamW (slim: no gradients)
            fp32alt=dict(B=1, steps=20, lr=5e-3, seed=8, out="horizon_sd14_b1_alt.pt", grad_steps=(), threads=3, slim=True),
            # SENSITIVITY of the reference's own arithmetic to gradient round-off: the same oracle run with the MLP gradient of every step
            # perturbed by seeded noise of relative L2 size 4e-5 — what the HIP path's dW measures against the oracle at SD size
            # (summation order inside 1.7 TFLOP of fp32 convolutions). AdamW turns that into weight differences of order lr on the
            # elements whose gradient is round-off sized; this trajectory shows how fast ANY fp32 evaluation with such an error separates.
            fp32pert=dict(B=1, steps=100, lr=5e-3, seed=8, out="horizon_sd14_b1_pert.pt", grad_steps=(), slim=True, grad_noise=4e-5),
            # BASELINE config 2's own length (aigc_id.yaml:135 max_steps: 800): the `fp32` run continued to 800 optimiser steps (≈ 2 h on 8 host
            # threads); its first 100 steps are the `fp32` fixture's. slim800 keeps loss, step coefficients and dictionary rows per step.
            fp32long=dict(B=1, steps=800, lr=5e-3, seed=8, out="horizon_sd14_b1_800.pt", grad_steps=(), slim=True, coef_steps=200),
            # the fp32 stream with LOW-PASS ("photo-like") face crops: the HIP path's 4e-5 step-coefficient difference on the default stream is
            # the fp32 rounding of the warp's sampling coordinate on WHITE-NOISE faces; on smooth faces it is 5e-6 (probe_warp_rounding.py)
            fp32photo=dict(B=1, steps=800, lr=5e-3, seed=8, out="horizon_sd14_b1_photo.pt", grad_steps=(), slim=True, photo=True, coef_steps=200),
            # two more draws of the 800-step claim (round-4 verdict: one stream, one seed is one sample of a random event): other frozen
            # weights AND another batch stream; slim fixtures (loss per step, learned rows every 4th step: 0.9 MB each)
            fp32photo2=dict(B=1, steps=800, lr=5e-3, seed=9, batch_seed0=3000, out="horizon_sd14_b1_photo2.pt", grad_steps=(), slim=True, photo=True,
                            coef_steps=0, id_coef_stride=4),
            fp32photo3=dict(B=1, steps=800, lr=5e-3, seed=10, batch_seed0=5000, out="horizon_sd14_b1_photo3.pt", grad_steps=(), slim=True, photo=True,
                            coef_steps=0, id_coef_stride=4),
            # the DEFAULT (white-noise) stream drawn again: other frozen weights, another batch stream
            fp32noise2=dict(B=1, steps=800, lr=5e-3, seed=11, batch_seed0=7000, out="horizon_sd14_b1_noise2.pt", grad_steps=(), slim=True,
                            coef_steps=0, id_coef_stride=4),
            # round 6: the default stream on the face-warp table of ANOTHER host — the GPU box's EPYC, whose F.affine_grid differs from this
            # host's in 14 % of the elements by one ulp (tests/golden/warp_grid_gpubox_512.npz, captured there by
            # tools/probes/capture_host_affine_grid.py). This is the oracle on the table the PRODUCT'S DEFAULT path (lib.Trainer builds the
            # table with the host's own F.affine_grid) uses on that box: tests/test_gpu_models.py::test_train_120_steps_sd14_default_table
            fp32boxgrid=dict(B=1, steps=120, lr=5e-3, seed=8, out="horizon_sd14_b1_boxgrid.pt", grad_steps=(), slim=True, coef_steps=120,
                             grid_npz="warp_grid_gpubox_512.npz"),
            # round 6: the first 120 steps of the three slim draws AGAIN, this time keeping the per-step coefficients of every face (the slim
            # 800-step files hold none): the implementation's own invariant — per-step agreement, asserted at 2e-5 — on all five draws
            # (tests/test_gpu_models.py::test_train_800_steps_sd14_more_draws; the losses / learned rows of these files must equal the slim ones')
            # (`threads` = what the slim run had — three generators shared the host then: torch-CPU's GEMM partitioning, hence the round-off, follows it)
            fp32photo2head=dict(B=1, steps=120, lr=5e-3, seed=9, batch_seed0=3000, out="horizon_sd14_b1_photo2_head.pt", grad_steps=(), slim=True, photo=True,
                                coef_steps=120, id_coef_stride=4, threads=5),
            fp32photo3head=dict(B=1, steps=120, lr=5e-3, seed=10, batch_seed0=5000, out="horizon_sd14_b1_photo3_head.pt", grad_steps=(), slim=True, photo=True,
                                coef_steps=120, id_coef_stride=4, threads=5),
            fp32noise2head=dict(B=1, steps=120, lr=5e-3, seed=11, batch_seed0=7000, out="horizon_sd14_b1_noise2_head.pt", grad_steps=(), slim=True,
                                coef_steps=120, id_coef_stride=4, threads=3),
            bf16=dict(B=4, steps=20, lr=2e-2, seed=5, out="horizon_sd14_b4.pt", grad_steps=(1, 2, 5, 10, 20)),
            # round 6: config 3's stream continued to 100 steps (its first 20 are the `bf16` fixture's): a horizon beyond 20 for the bf16 path
            bf16long=dict(B=4, steps=100, lr=2e-2, seed=5, out="horizon_sd14_b4_100.pt", grad_steps=(), slim=True, coef_steps=100))
ROW_STRIDE = 16


def main(which, steps=None):
    r = dict(RUNS[which])
    if steps:
        r["steps"] = int(steps)
    if r.get("threads"):
        torch.set_num_threads(r["threads"])
    if r.get("grid_npz"):
        import numpy as np
        from oracle.replay import fixture_host_warp_grid
        z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", r["grid_npz"]))
        S = int(z["S"])
        table = fixture_host_warp_grid(S).numpy().copy()
        table.reshape(-1)[np.unpackbits(z["mask"])[:table.size].astype(bool)] = z["values"]
        import hashlib
        assert hashlib.sha1(table.tobytes()).hexdigest()[:12] == str(z["sha1_host"]), "captured table does not reproduce its digest"
        table_t = torch.from_numpy(table)
        real_grid = O.F.affine_grid

        def host_grid(theta, size, align_corners=None):          # the reference's call with its constant matrix at S x S -> the other host's table
            if list(size[-2:]) == [S, S] and align_corners and torch.equal(theta[0], torch.tensor(O.TRANS_MATRIX, dtype=torch.float32)):
                return table_t[None].expand(theta.shape[0], S, S, 2).contiguous()
            return real_grid(theta, size, align_corners=align_corners)
        O.F.affine_grid = host_grid
    if r.get("grad_noise"):
        eps, gen, real = float(r["grad_noise"]), torch.Generator().manual_seed(77), O.adamw_step

        def noisy_adamw(p, g, m, v, step, lr, **kw):
            g = g + eps * g.norm() / (g.numel() ** 0.5) * torch.randn(g.shape, generator=gen)
            return real(p, g, m, v, step, lr, **kw)
        O.adamw_step = noisy_adamw
    m = dict(unet=O.UNET_SD14, vae=O.VAE_SD14, clip=O.CLIP_L14, fr=O.FR_R100, es=2, inner=512, max_ids=10, image=512,
             placeholder=O.PLACEHOLDER_SKS)
    shapes = O.trainer_param_shapes(m["unet"], m["vae"], m["clip"], m["fr"], es=m["es"], inner=m["inner"], max_ids=m["max_ids"])
    W = O.init_weights(shapes, seed=r["seed"])
    st = O.TrainState(W)
    ids = sorted({1 + (b % (m["max_ids"] - 1)) for b in range(r["B"])})
    seed0 = int(r.get("batch_seed0", 100))
    rec = dict(cfg=dict(B=r["B"], lr=r["lr"], weight_seed=r["seed"], batch_seed0=seed0, grid_npz=r.get("grid_npz"), id_coef_stride=int(r.get("id_coef_stride", 1)), row_stride=ROW_STRIDE, ids=ids, photo=bool(r.get("photo")),
                        torch=torch.__version__, threads=torch.get_num_threads()),
               loss=[], coef=[], id_coef=[], id_emb_norm=[], gW_rows={}, gb={})
    W0 = st.Wm.clone()
    t0 = time.time()
    for s in range(r["steps"]):
        batch = make_batch(r["B"], 512, 64, m["clip"]["vocab"], m["placeholder"], seed0 + s, m["max_ids"], photo=bool(r.get("photo")))
        loss, aux = O.train_step(W, st, batch, r["lr"], m["unet"], m["vae"], m["clip"], m["fr"], es=2, inner=512,
                                 placeholder=m["placeholder"], return_aux=True)
        rec["loss"].append(float(loss))
        rec["coef"].append(aux["coef"].reshape(2 * r["B"], -1).clone())
        rec["id_coef"].append(st.id_coef[ids].clone())
        rec["id_emb_norm"].append(st.id_emb[ids].norm(dim=1).clone())
        if s + 1 in r["grad_steps"]:
            rec["gW_rows"][s + 1] = aux["gW"][::ROW_STRIDE].clone()
            rec["gb"][s + 1] = aux["gb"].clone()
        print(f"[gen_horizon {which}] step {s + 1}/{r['steps']} loss {float(loss):.6f} ({time.time() - t0:.0f} s)", flush=True)
        if (s + 1) % 10 == 0 or s + 1 == r["steps"]:
            out = dict(rec, steps=s + 1, loss=torch.tensor(rec["loss"], dtype=torch.float64), coef=torch.stack(rec["coef"]),
                       id_coef=torch.stack(rec["id_coef"]), id_emb_norm=torch.stack(rec["id_emb_norm"]),
                       Wm_rows=st.Wm[::ROW_STRIDE].clone(), W0_rows=W0[::ROW_STRIDE].clone(), bm=st.bm.clone())
            if r.get("slim"):
                # coef_steps: the per-step coefficients are what the kink-crossing analysis reads (the first crossings); beyond that
                # horizon only the learned (EMA dictionary) rows and the losses are kept — 5 MB instead of 10 for the 800-step run
                out = dict(cfg=out["cfg"], steps=out["steps"], loss=out["loss"], coef=out["coef"][:r.get("coef_steps", out["steps"])].clone(),
                           id_coef=out["id_coef"][r.get("id_coef_stride", 1) - 1::r.get("id_coef_stride", 1)].clone())   # rows = steps k, 2k, ...
            path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", r["out"])
            torch.save(out, path + ".tmp")
            os.replace(path + ".tmp", path)


if __name__ == "__main__":
    main(*sys.argv[1:])
