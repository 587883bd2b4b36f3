"""CPU ORACLE — TEST INFRASTRUCTURE ONLY. BASELINE config 2's horizon (aigc_id.yaml:135 max_steps: 800) run by the REFERENCE ITSELF at
mini size: 800 optimiser steps of oracle/gen_golden.py::step_harness (the imported LatentDiffusion.forward / p_losses / q_sample /
get_loss, UNetModel, Encoder, EmbeddingManagerId + MetaIdNet, torch.optim.AdamW — the harness tests/golden/step_mini.pt pins for
3 steps), recorded per step as (loss, EMA dictionary rows = the `id_coefficients` the checkpoint holds, embedding_manager.py:484-489).

    python oracle/gen_horizon_ref.py  ->  tests/golden/horizon_mini_ref800.pt  (≈ 10 min; needs /root/reference, like gen_golden.py)

Runs in the file:
  ref            the reference, 8 host threads
  ref_fr3e-6, ref_fr2.5e-5, ref_fr2e-4   the SAME reference code with seeded noise of that relative L2 size on the FR-net's features:
                 3e-6 is what two correct fp32 evaluations of the backbone differ by (the oracle's fp32 vs fp64 backbone: 2e-6),
                 2.5e-5 is what the HIP path's feature difference measures at SD size on the white-noise replay faces (the fp32 rounding
                 of a 512-pixel warp coordinate, profiles/r04_warp_rounding.json), 2e-4 a decade above
  oracle         oracle.cb_oracle.train_step on the same stream (what the `-m gpu` tests hold the HIP path to at SD size)
The tests hold the oracle AND the HIP path to `ref` (tests/test_oracle_golden.py, tests/test_gpu_models.py) and compare their
distance from it with the distance of the perturbed reference runs."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gen_golden as GG                                  # noqa: E402  (puts /root/reference on sys.path)
from oracle import cb_oracle as O                        # noqa: E402
from oracle.replay import ref_stream_batch               # noqa: E402

K = 800
NOISE = (("ref", 0.0), ("ref_fr3e-6", 3e-6), ("ref_fr2.5e-5", 2.5e-5), ("ref_fr2e-4", 2e-4))


def sensitivity(inner):
    """--inner N: the same four reference runs with an N-wide head (N = 512: the real head on the mini networks, 1024 learned coefficients
    per identity): how the distance between two runs of the REFERENCE depends on the size of an FR-feature perturbation when the head has
    the real number of LeakyReLU kinks. Evidence only (profiles/r04_horizon_ref_w<N>_sensitivity.json), not a fixture."""
    import json
    import transformers  # noqa: F401
    from transformers import CLIPTokenizer, CLIPTextModel  # noqa: F401
    GG.install_shims()
    torch.set_num_threads(4)
    out, run = GG.step_harness(inner=inner)
    ids = lambda s_: torch.tensor([[1, 1], [2, 2]])
    for tag, eps in NOISE:
        run(False, K, 3, ids, tag, slim=True, fr_noise=eps)
    ref = torch.stack([s["id_coef"] for s in out["ref"]["steps"]])
    res = dict(inner=inner, steps=K, lr=out["lr"], learned_coefficient_l2_vs_unperturbed_reference={})
    for tag, eps in NOISE[1:]:
        d = (torch.stack([s["id_coef"] for s in out[tag]["steps"]]) - ref).norm(dim=2).max(dim=1).values
        res["learned_coefficient_l2_vs_unperturbed_reference"][tag] = {str(k): float(d[k - 1]) for k in (1, 10, 50, 100, 200, 400, 800)}
        res["learned_coefficient_l2_vs_unperturbed_reference"][tag]["max"] = float(d.max())
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", f"r04_horizon_ref_w{inner}_sensitivity.json")
    with open(path, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res, indent=1))


def main(only_oracle=False):
    import transformers  # noqa: F401
    from transformers import CLIPTokenizer, CLIPTextModel  # noqa: F401
    GG.install_shims()
    torch.set_num_threads(8)
    out, run = GG.step_harness()
    ids = lambda s_: torch.tensor([[1, 1], [2, 2]])
    t0 = time.time()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "horizon_mini_ref800.pt")
    if only_oracle:                                          # --oracle: keep the reference runs of the existing file, redo the oracle's
        old = torch.load(path, weights_only=False)
        run(False, 1, 3, ids, "ref", slim=True)              # (one step: the initial state is drawn inside run())
        for tag, _ in NOISE:
            out[tag] = dict(out["ref"], steps=[dict(loss=float(l), id_coef=c) for l, c in zip(old[tag]["loss"], old[tag]["id_coef"])],
                            W_final=old[tag]["W_final"], b_final=old[tag]["b_final"])
    else:
        for tag, eps in NOISE:
            run(False, K, 3, ids, tag, slim=True, fr_noise=eps)
            print(f"[gen_horizon_ref] {tag} done ({time.time() - t0:.0f} s)", flush=True)
    fx = dict(cfg={k: out[k] for k in ("unet", "vae", "clip", "fr", "es", "inner", "lr", "scale", "seeds")}, steps=K, torch=torch.__version__)
    ref = out["ref"]
    fx["init"] = {k: ref[k] for k in ("Wm", "bm", "id_coef0", "id_emb0")}
    for tag, _ in NOISE:
        r = out[tag]
        fx[tag] = dict(loss=torch.tensor([s["loss"] for s in r["steps"]], dtype=torch.float64), id_coef=torch.stack([s["id_coef"] for s in r["steps"]]),
                       W_final=r["W_final"], b_final=r["b_final"])
    # the oracle on the same stream
    sd = out["seeds"]
    W = {}
    W.update(O.init_weights(O.unet_param_shapes(out["unet"], "model.diffusion_model."), seed=sd["unet"]))
    W.update(O.init_weights(O.vae_encoder_param_shapes(out["vae"], "first_stage_model."), seed=sd["vae"]))
    W.update(O.init_weights(O.clip_param_shapes(out["clip"], "cond_stage_model.transformer.text_model."), seed=sd["clip"]))
    W["cond_stage_model.transformer.text_model.embeddings.token_embedding.weight"] = O.formula_table(out["clip"]["vocab"], out["clip"]["width"])
    W.update(O.init_weights(O.frnet_param_shapes(out["fr"], "embedding_manager.meta_id_net.id_model."), seed=sd["fr"]))
    W["celeb_embeddings"] = O.init_weights({"celeb_embeddings": (out["es"], 1 + out["inner"], out["clip"]["width"])}, seed=sd["basis"])["celeb_embeddings"]
    W["embedding_manager.meta_id_net.stylegan_mlp.net.0.weight"] = ref["Wm"].clone()
    W["embedding_manager.meta_id_net.stylegan_mlp.net.0.bias"] = ref["bm"].clone()
    W["id_coefficients"] = ref["id_coef0"].clone(); W["id_embeddings"] = ref["id_emb0"].clone()
    st = O.TrainState(W)
    lat = 64 // (2 ** (len(out["vae"]["ch_mult"]) - 1))
    losses, rows = [], []
    for s in range(K):
        loss = O.train_step(W, st, ref_stream_batch(s, lat), out["lr"], out["unet"], out["vae"], out["clip"], out["fr"], es=out["es"], inner=out["inner"],
                            scale=out["scale"])
        losses.append(float(loss)); rows.append(st.id_coef.clone())
    fx["oracle"] = dict(loss=torch.tensor(losses, dtype=torch.float64), id_coef=torch.stack(rows), W_final=st.Wm.clone(), b_final=st.bm.clone())
    print(f"[gen_horizon_ref] oracle done ({time.time() - t0:.0f} s)", flush=True)
    torch.save(fx, path)
    for tag in [t for t, _ in NOISE[1:]] + ["oracle"]:
        d = (fx[tag]["id_coef"] - fx["ref"]["id_coef"]).norm(dim=2).max(dim=1).values
        print(tag, "learned-coefficient L2 vs ref at steps 1/10/100/200/400/800:", [f"{float(d[i - 1]):.2e}" for i in (1, 10, 100, 200, 400, 800)])


if __name__ == "__main__":
    if "--inner" in sys.argv:
        sensitivity(int(sys.argv[sys.argv.index("--inner") + 1]))
    else:
        main(only_oracle="--oracle" in sys.argv)
