"""CPU: the oracle (oracle/cb_oracle.py) against the golden vectors produced by the reference's own modules
(oracle/gen_golden.py, run in the build container). Tolerances are fp32 round-off class."""
import json
import os

import torch
import torch.nn.functional as F

from oracle import cb_oracle as O
from oracle.gen_golden import keys_digest


def _load(golden_dir, name):
    return torch.load(os.path.join(golden_dir, name), weights_only=False)


def test_schedule_known_answers(golden_dir):
    fx = _load(golden_dir, "schedule.pt")
    s = O.make_schedule()
    for k in ("betas", "alphas_cumprod", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod"):
        assert torch.equal(s[k], fx[k]), k
    # SURVEY §8 A1 known answers
    assert abs(float(s["betas"][0]) - 0.00085) < 1e-9 and abs(float(s["betas"][999]) - 0.012) < 1e-8
    assert abs(float(s["sqrt_alphas_cumprod"][500]) - 0.5256736) < 1e-6
    assert abs(float(s["alphas_cumprod"][999]) - 0.0046601) < 1e-7
    t = fx["q_t"]
    xt = s["sqrt_alphas_cumprod"][t].reshape(-1, 1, 1, 1) * fx["q_x0"] + s["sqrt_one_minus_alphas_cumprod"][t].reshape(-1, 1, 1, 1) * fx["q_eps"]
    assert torch.equal(xt, fx["q_xt"])
    # fp32 cos / sin of arguments up to 999 rad: libm builds differ in the LAST BIT OF THE ARGUMENT t * freq across hosts (the product's
    # rounding and the exp that makes freq), and d cos / d arg <= 1: the bar is per element — two ulps of that element's argument plus
    # two ulps of the result — so the low-frequency channels (arguments << 1) are held to 5e-7, and a wrong frequency or channel order
    # anywhere still fails (round-5 advisor finding: a global atol of 2e-4 would have hidden one there)
    import math
    t = fx["temb_t"].to(torch.float64)
    half = 160
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float64) / half)
    arg = t[:, None] * freqs[None]
    tol = (2.0 * arg.abs() * 2.0 ** -23 + 2.0 * 2.0 ** -23).to(torch.float32)
    tol = torch.cat([tol, tol], dim=-1)
    got, ref = O.timestep_embedding(fx["temb_t"], 320), fx["temb"]
    assert got.shape == ref.shape == tol.shape and bool(((got - ref).abs() <= tol).all()), float(((got - ref).abs() - tol).max())
    # ... and against the float64 evaluation of the same formula (util.py:151-167: cos first, then sin): the reference's fp32 frequency
    # exp(-ln(1e4) j / half) carries the rounding of an argument of magnitude up to 9.2, i.e. up to ~8 ulps of relative error
    exact = torch.cat([torch.cos(arg), torch.sin(arg)], dim=-1).to(torch.float32)
    assert bool(((got - exact).abs() <= 8.0 * tol).all()), float(((got - exact).abs() / tol).max())


def test_shift_rows_bit_exact(golden_dir):
    for case in _load(golden_dir, "shift_rows.pt"):
        out = O.shift_rows(case["tokens"], case["embedded"], case["z"], case["placeholder"], case["reps"])
        assert torch.equal(out, case["out"]), case["tokens"][:12]


def test_shift_rows_multi_bit_exact(golden_dir):
    for case in _load(golden_dir, "shift_rows_multi.pt"):
        out = O.shift_rows_multi(case["tokens"], case["embedded"], case["z"], case["placeholders"], case["reps"])
        assert torch.equal(out, case["out"]), case["placeholders"]


def test_state_dict_tables_match_reference(golden_dir):
    with open(os.path.join(golden_dir, "state_dict_digests.json")) as f:
        d = json.load(f)
    assert keys_digest(O.unet_param_shapes(O.UNET_SD14)) == d["unet_sd14"]
    assert keys_digest(O.vae_encoder_param_shapes(O.VAE_SD14)) == d["vae_encoder_sd14"]
    assert keys_digest(O.frnet_param_shapes(O.FR_R100)) == d["iresnet100"]
    assert keys_digest(O.clip_param_shapes(O.CLIP_L14)) == d["clip_l14_text"]
    assert d["unet_sd14"]["params"] == 859520964
    # trainable tensors: EqualLinear(512 -> 1024) = 525 312 floats
    s = O.trainer_param_shapes()
    n = 1
    for v in s["embedding_manager.meta_id_net.stylegan_mlp.net.0.weight"]:
        n *= v
    assert n + s["embedding_manager.meta_id_net.stylegan_mlp.net.0.bias"][0] == 525312


def test_unet_forward_backward(golden_dir):
    fx = _load(golden_dir, "unet_mini.pt")
    W = O.init_weights(O.unet_param_shapes(fx["cfg"]), seed=fx["weight_seed"])
    ctx = fx["context"].clone().requires_grad_(True)
    out = O.unet_forward(W, fx["cfg"], fx["x"], fx["t"], ctx)
    (dctx,) = torch.autograd.grad((out * fx["dout"]).sum(), [ctx])
    assert torch.allclose(out, fx["out"], atol=2e-5, rtol=1e-4)
    assert torch.allclose(dctx, fx["dcontext"], atol=2e-5, rtol=1e-4)


def test_vae_encode(golden_dir):
    fx = _load(golden_dir, "vae_mini.pt")
    W = O.init_weights(O.vae_encoder_param_shapes(fx["cfg"]), seed=fx["weight_seed"])
    with torch.no_grad():
        mom = O.vae_encode_moments(W, fx["cfg"], fx["x"])
        z = O.vae_sample(mom, fx["noise"])
    assert torch.allclose(mom, fx["moments"], atol=2e-5, rtol=1e-4)
    assert torch.allclose(z, fx["z_scaled"], atol=2e-5, rtol=1e-4)


def test_frnet(golden_dir):
    fx = _load(golden_dir, "frnet_mini.pt")
    W = O.init_weights(O.frnet_param_shapes(fx["cfg"]), seed=fx["weight_seed"])
    with torch.no_grad():
        out = O.frnet_forward(W, fx["cfg"], fx["x"])
    assert torch.allclose(out, fx["out"], atol=1e-4, rtol=1e-4)


def test_clip_text_layers(golden_dir):
    fx = _load(golden_dir, "clip_mini.pt")
    W = O.init_weights(O.clip_param_shapes(fx["cfg"], fx["weight_prefix"]), seed=fx["weight_seed"])
    emb = W[fx["weight_prefix"] + "embeddings.token_embedding.weight"][fx["tokens"]]
    with torch.no_grad():
        out = O.clip_encode(W, fx["cfg"], emb, fx["weight_prefix"])
    assert torch.allclose(out, fx["last_hidden_state"], atol=2e-5, rtol=1e-4)


def test_manager_forward_backward_and_ema(golden_dir):
    """EmbeddingManagerId.forward (training) = warp/resize -> FR-net -> normalize -> MLP head -> basis -> EMA -> row surgery"""
    fx = _load(golden_dir, "manager_mini.pt")
    es, inner = fx["es"], fx["inner"]
    P = "embedding_manager.meta_id_net."
    W = O.init_weights(O.frnet_param_shapes(fx["fr_cfg"], P + "id_model."), seed=fx["fr_weight_seed"])
    Wm = fx["Wm"].clone().requires_grad_(True); bm = fx["bm"].clone().requires_grad_(True)
    W[P + "stylegan_mlp.net.0.weight"] = Wm; W[P + "stylegan_mlp.net.0.bias"] = bm
    W["celeb_embeddings"] = fx["basis"]
    assert fx["n_trainable"] == Wm.numel() + bm.numel()
    faces = torch.cat(fx["faces"].chunk(2, -1), 0)
    with torch.no_grad():
        _, small = O.face_warp_resize(faces, fx["fr_cfg"]["in_size"])
        v = F.normalize(O.frnet_forward(W, fx["fr_cfg"], small, P + "id_model."), dim=-1)
    coef, z = O.celeb_head(W, v, es, inner)
    B = fx["tokens"].shape[0]
    table = O.formula_table(fx["vocab"], 768)
    emb_in = table[fx["tokens"]]
    out = torch.stack([O.shift_rows(fx["tokens"][b], emb_in[b], z[b], 48136, es) for b in range(B)])
    assert torch.allclose(out, fx["out"], atol=1e-5, rtol=1e-4)
    dout = O.formula_table(out.shape[0] * out.shape[1], out.shape[2], 0.9).reshape(out.shape) * 50.0
    gW, gb = torch.autograd.grad((out * dout).sum(), [Wm, bm])
    assert torch.allclose(gW, fx["gW"], atol=1e-6, rtol=2e-3)
    assert torch.allclose(gb, fx["gb"], atol=1e-6, rtol=2e-3)
    id_coef, id_emb = fx["id_coef0"].clone(), fx["id_emb0"].clone()
    for b in range(B):
        i = int(fx["ids"][b, 0])
        id_emb[i] = 0.99 * id_emb[i] + 0.01 * z[b].detach().reshape(-1)
        id_coef[i] = 0.99 * id_coef[i] + 0.01 * coef[b].detach().reshape(-1)
    assert torch.allclose(id_coef, fx["id_coef1"], atol=1e-6)
    assert torch.allclose(id_emb, fx["id_emb1"], atol=1e-6)


def test_adamw_matches_torch():
    g = torch.Generator().manual_seed(0)
    p = torch.randn(64, 8, generator=g)
    ref = torch.nn.Parameter(p.clone())
    opt = torch.optim.AdamW([ref], lr=1e-2)
    m = torch.zeros_like(p); v = torch.zeros_like(p)
    for step in range(1, 6):
        grad = torch.randn(64, 8, generator=g)
        ref.grad = grad.clone()
        opt.step()
        p, m, v = O.adamw_step(p, grad, m, v, step, 1e-2)
        assert torch.allclose(p, ref.detach(), atol=1e-6, rtol=1e-5)


def test_config1_cpu_plumbing_ten_steps():
    """BASELINE config 1 (plumbing, no GPU): 10 personalization steps of the CPU restatement on a single identity at reduced
    width; the loss stays finite, the EMA coefficient block moves towards the (unit-norm) prediction and only the 2 x inner
    coefficients of the trained identity change."""
    unet = dict(in_channels=4, out_channels=4, model_channels=32, num_res_blocks=1, channel_mult=(1, 2), attention_resolutions=(1, 2),
                num_heads=2, context_dim=64)
    vae = dict(ch=32, ch_mult=(1, 2), num_res_blocks=1, in_channels=3, z_channels=4)
    clip = dict(vocab=100, width=64, layers=2, heads=2, ctx_len=77, mlp=128)
    fr = dict(layers=(1, 1, 1, 1), width=16, in_size=32, feat=512)
    W = O.init_weights(O.trainer_param_shapes(unet, vae, clip, fr, es=2, inner=16, max_ids=3), seed=1)
    st = O.TrainState(W)
    coef0 = st.id_coef.clone()
    g = torch.Generator().manual_seed(5)
    losses = []
    for step in range(10):
        tokens = torch.randint(0, 90, (1, 77), generator=g); tokens[:, 5] = 99
        batch = dict(image=torch.rand(1, 32, 32, 3, generator=g) * 2 - 1, faces=torch.rand(1, 32, 32, 6, generator=g) * 2 - 1,
                     ids=torch.tensor([[1, 1]]), tokens=tokens, t=torch.randint(0, 1000, (1,), generator=g),
                     noise=torch.randn(1, 4, 16, 16, generator=g), post_noise=torch.randn(1, 4, 16, 16, generator=g))
        losses.append(float(O.train_step(W, st, batch, 5e-3, unet, vae, clip, fr, es=2, inner=16, placeholder=99)))
    assert all(torch.isfinite(torch.tensor(losses))) and st.step == 10
    assert torch.equal(st.id_coef[0], coef0[0]) and torch.equal(st.id_coef[2], coef0[2])      # other identities untouched
    assert not torch.equal(st.id_coef[1], coef0[1])
    # EMA with momentum 0.99 of a unit-norm prediction: the norm of each (e) block contracts by 0.99^10 plus the drift
    blk = st.id_coef[1].reshape(2, 16)
    assert (blk.norm(dim=1) < coef0[1].reshape(2, 16).norm(dim=1) + 0.2).all()


def test_vae_decoder_vs_reference(golden_dir):
    """sampling path: post_quant_conv + Decoder, decode_first_stage scaling, [0,1] image map"""
    fx = _load(golden_dir, "vae_dec_mini.pt")
    with open(os.path.join(golden_dir, "state_dict_digests.json")) as f:
        assert keys_digest(O.vae_decoder_param_shapes(O.VAE_SD14)) == json.load(f)["vae_decoder_sd14"]
    W = O.init_weights(O.vae_decoder_param_shapes(fx["cfg"]), seed=fx["weight_seed"])
    with torch.no_grad():
        img = O.decode_first_stage(W, fx["cfg"], fx["z"], fx["scale_factor"])
    assert torch.allclose(img, fx["image"], atol=2e-5, rtol=1e-4)
    assert torch.allclose(torch.clamp((img + 1.0) / 2.0, 0.0, 1.0), fx["image01"], atol=2e-5, rtol=0)


def test_ddim_schedule_and_sampling_vs_reference(golden_dir):
    """DDIM tables bit-exact; one p_sample_ddim update and short guided runs (eta 0 and eta 0.5) over the mini UNet"""
    fx = _load(golden_dir, "ddim_mini.pt")
    ac = O.make_schedule()["alphas_cumprod"]
    for S, eta in ((50, 0.0), (5, 0.0), (4, 0.5)):
        ref, got = fx[f"sched_{S}_{eta}"], O.ddim_schedule(ac, S, eta)
        assert torch.equal(torch.as_tensor(got["timesteps"]), ref["timesteps"])
        # (the square root is one ulp host dependent: bit-exact on the fixture's host, 1e-7 relative elsewhere)
        assert torch.equal(got["alphas"], ref["alphas"]) and torch.allclose(got["sqrt_one_minus_alphas"], ref["sqrt_one_minus_alphas"], atol=0, rtol=2e-7)
        assert torch.equal(got["alphas_prev"], ref["alphas_prev"].float()) and torch.equal(got["sigmas"], ref["sigmas"].float())
    assert fx["sched_50_0.0"]["timesteps"][:3].tolist() == [1, 21, 41]           # range(0,1000,20)+1, SURVEY §8 A10
    assert torch.equal(torch.as_tensor(O.ddim_schedule(ac, 10, 0.0, method="quad")["timesteps"]), fx["quad_10"])
    cfg = fx["cfg"]
    W = O.init_weights(O.unet_param_shapes(cfg), seed=fx["weight_seed"])
    with torch.no_grad():
        s5 = O.ddim_schedule(ac, 5, 0.0)
        t = torch.full((2,), int(s5["timesteps"][2]), dtype=torch.long)
        e_u, e_c = O.unet_forward(W, cfg, torch.cat([fx["x_T"]] * 2), torch.cat([t] * 2), torch.cat([fx["uncond"], fx["cond"]])).chunk(2)
        x1, p0 = O.ddim_step(fx["x_T"], e_u + 7.5 * (e_c - e_u), s5["alphas"][2], s5["alphas_prev"][2], s5["sigmas"][2],
                             s5["sqrt_one_minus_alphas"][2])
        assert torch.allclose(x1, fx["step_index2"]["x_prev"], atol=3e-5, rtol=1e-4)
        assert torch.allclose(p0, fx["step_index2"]["pred_x0"], atol=3e-5, rtol=1e-4)
        x, pred = O.ddim_sample(W, cfg, fx["x_T"], fx["cond"], fx["uncond"], s5, scale=7.5)
        assert torch.allclose(x, fx["run_5_eta0_cfg7.5"]["samples"], atol=2e-4, rtol=1e-3)
        assert torch.allclose(pred, fx["run_5_eta0_cfg7.5"]["pred_x0"], atol=2e-4, rtol=1e-3)
        x, _ = O.ddim_sample(W, cfg, fx["x_T"], fx["cond"], None, s5, scale=1.0)
        assert torch.allclose(x, fx["run_5_eta0_nocfg"]["samples"], atol=1e-4, rtol=1e-3)
        r = fx["run_4_eta0.5_cfg3"]
        x, _ = O.ddim_sample(W, cfg, fx["x_T"], fx["cond"], fx["uncond"], O.ddim_schedule(ac, 4, 0.5), scale=3.0, noise=r["noise"])
        assert torch.allclose(x, r["samples"], atol=2e-4, rtol=1e-3)


def test_host_ddim_schedule_matches_reference(golden_dir):
    """the host mirror's make_schedule buffers (celebbasis_amd.ldm.models.diffusion.ddim) against the reference's tables"""
    from celebbasis_amd.ldm.models.diffusion.ddim import DDIMSampler, ddim_timesteps
    fx = _load(golden_dir, "ddim_mini.pt")
    sch = O.make_schedule()

    class M:
        num_timesteps = 1000
        betas, alphas_cumprod = sch["betas"], sch["alphas_cumprod"]
        alphas_cumprod_prev = torch.cat([torch.ones(1), sch["alphas_cumprod"][:-1]])

    s = DDIMSampler(M())
    for S, eta in ((50, 0.0), (5, 0.0), (4, 0.5)):
        s.make_schedule(S, ddim_eta=eta, verbose=False)
        ref = fx[f"sched_{S}_{eta}"]
        assert torch.equal(torch.as_tensor(s.ddim_timesteps), ref["timesteps"])
        a, ap, sg, s1m = s._tables()
        assert torch.equal(torch.as_tensor(a), ref["alphas"]) and torch.equal(torch.as_tensor(ap), ref["alphas_prev"].float())
        assert torch.equal(torch.as_tensor(sg), ref["sigmas"].float()) and torch.allclose(torch.as_tensor(s1m), ref["sqrt_one_minus_alphas"], atol=0, rtol=2e-7)
        assert torch.equal(torch.as_tensor(s.ddim_sigmas), ref["sigmas"])            # float64, as the reference keeps it
    assert torch.equal(torch.as_tensor(ddim_timesteps("quad", 10, 1000)), fx["quad_10"])
    import pytest
    with pytest.raises(IndexError):          # 1000 // 3 = 333 -> steps [1, 334, 667, 1000]: out of range, as in the reference
        s.make_schedule(3, verbose=False)


def test_celeb_basis_builder_vs_reference(golden_dir):
    """celebbasis_amd.basis against the reference's own _get_celeb_embeddings (run on its name list and recorded BPE ids):
    mean row bit-exact; the singular vectors agree as a matrix when LAPACK matches, and at least per-row up to sign"""
    from celebbasis_amd import basis as B
    fx = _load(golden_dir, "basis_mini.pt")
    table = torch.randn(49408, fx["width"], generator=torch.Generator().manual_seed(fx["table_seed"])) * 0.02
    names = sorted(set(fx["names"]))                                   # rm_repeats + sort, modules.py:477-486
    assert len(names) == 580 and names[0] == "Adam Sandler"
    rows = [[49406] + fx["token_table"][n] + [49407] * (76 - len(fx["token_table"][n])) for n in names]
    for key, flatten in (("basis", False), ("basis_flatten", True)):
        ref = fx[key]
        got = B.build_celeb_basis(table, rows, n_components=fx["n_components"], num_embeds_per_token=2, use_flatten=flatten)
        assert got.shape == ref.shape == (2, 1 + fx["n_components"], fx["width"])
        assert torch.allclose(got[:, 0], ref[:, 0], atol=1e-7, rtol=0)                       # the mean embedding
        cos = (got[:, 1:] * ref[:, 1:]).sum(-1)                                              # rows are unit vectors
        assert torch.all(cos.abs() > 1 - 1e-4), float(cos.abs().min())
        assert torch.allclose(got, ref, atol=1e-4, rtol=0), "LAPACK sign convention differs from the fixture's"      # 1e-5 on the fixture's host; other LAPACK builds move the 5th digit
    # ragged rows (just the name's own tokens) give the same columns as padded rows
    cols_a = B.name_token_columns(rows)
    cols_b = B.name_token_columns([[49406] + fx["token_table"][n] for n in names])
    assert cols_a == cols_b and len(cols_a[0]) == 580 and len(cols_a[1]) == 576            # 4 single-token names
    # the documented quirk: the reference never drops repeated tokens; the intended filter would
    assert len(B.name_token_columns(rows, dedupe_tokens=True)[0]) < 580
    pm = B.build_celeb_basis(table, rows, n_components=8, sign="positive_max")
    lead = pm[:, 1:].gather(2, pm[:, 1:].abs().argmax(dim=2, keepdim=True))
    assert torch.all(lead > 0)


def _step_fixture_state(fx, tag):
    """weights (re-drawn from the recorded seeds) + the run's initial trainable / EMA state, under the oracle's names"""
    rec, max_ids = fx[tag], fx[tag]["id_coef0"].shape[0]
    sd = fx["seeds"]
    W = {}
    W.update(O.init_weights(O.unet_param_shapes(fx["unet"], "model.diffusion_model."), seed=sd["unet"]))
    W.update(O.init_weights(O.vae_encoder_param_shapes(fx["vae"], "first_stage_model."), seed=sd["vae"]))
    W.update(O.init_weights(O.clip_param_shapes(fx["clip"], "cond_stage_model.transformer.text_model."), seed=sd["clip"]))
    W["cond_stage_model.transformer.text_model.embeddings.token_embedding.weight"] = O.formula_table(fx["clip"]["vocab"], fx["clip"]["width"])
    W.update(O.init_weights(O.frnet_param_shapes(fx["fr"], "embedding_manager.meta_id_net.id_model."), seed=sd["fr"]))
    W["celeb_embeddings"] = O.init_weights({"celeb_embeddings": (fx["es"], 1 + fx["inner"], fx["clip"]["width"])}, seed=sd["basis"])["celeb_embeddings"]
    W["embedding_manager.meta_id_net.stylegan_mlp.net.0.weight"] = rec["Wm"].clone()
    W["embedding_manager.meta_id_net.stylegan_mlp.net.0.bias"] = rec["bm"].clone()
    W["id_coefficients"] = rec["id_coef0"].clone(); W["id_embeddings"] = rec["id_emb0"].clone()
    if rec["coef0"] is not None:
        W["embedding_manager.meta_id_net.coef"] = rec["coef0"].clone()
    return W, rec, max_ids


def step_fixture_batch(st):
    """one replay-stream element of step_mini.pt (the images are re-drawn from the recorded seed)"""
    gi = torch.Generator().manual_seed(st["image_seed"])
    image = torch.rand(2, 64, 64, 3, generator=gi) * 2 - 1; faces = torch.rand(2, 64, 64, 6, generator=gi) * 2 - 1
    return dict(image=image, faces=faces, ids=st["ids"], tokens=st["tokens"], t=st["t"], noise=st["noise"], post_noise=st["post_noise"])


def test_composed_training_step_vs_reference_run(golden_dir):
    """SURVEY C-8 item 6 / the composition of the step is no longer "the oracle's own": oracle.train_step on the replay stream
    recorded from the REFERENCE's LatentDiffusion.forward / p_losses / q_sample / get_loss + torch.optim.AdamW run
    (tests/golden/step_mini.pt, oracle/gen_golden.py:gen_step) — per-step loss and latent, first-step gradient, final MLP weights
    and EMA dictionaries; default MLP path and `use_rm_mlp` (with the reference's flatten(ids) pairing across a batch of two ids)."""
    fx = _load(golden_dir, "step_mini.pt")
    for tag, rm in (("mlp", False), ("rm_mlp", True)):
        W, rec, max_ids = _step_fixture_state(fx, tag)
        st = O.TrainState(W)
        for i, s in enumerate(rec["steps"]):
            loss, aux = O.train_step(W, st, step_fixture_batch(s), fx["lr"], fx["unet"], fx["vae"], fx["clip"], fx["fr"], es=fx["es"],
                                     inner=fx["inner"], scale=fx["scale"], return_aux=True, use_rm_mlp=rm)
            assert torch.allclose(aux["z0"], s["z0"], atol=2e-6, rtol=1e-5), (tag, i)
            assert abs(float(loss) - float(s["loss"])) <= 2e-6 * abs(float(s["loss"])), (tag, i, float(loss), float(s["loss"]))
            assert abs(float(s["loss"]) - float(s["loss_simple"])) < 1e-7                  # logvar == 0, elbo weight 0, neg loss 0
            if s["grad"] is not None:
                g = aux["gW"].reshape(s["grad"].shape)
                assert torch.allclose(g, s["grad"], atol=2e-6 * float(s["grad"].abs().max()), rtol=1e-4), (tag, i)
        if rm:
            assert torch.allclose(st.coef, rec["coef_final"], atol=1e-6, rtol=1e-5)
        else:
            # AdamW divides by sqrt(v): an element whose gradient is ~0 turns fp32 round-off into an O(lr) difference, so compare
            # against the size of the UPDATE: at most 1e-2 of the largest weight change, and all but a handful of elements exact
            d = (st.Wm - rec["W_final"]).abs()
            upd = float((rec["W_final"] - rec["Wm"]).abs().max())
            assert float(d.max()) <= 1e-2 * upd and int((d > 1e-5).sum()) <= d.numel() // 1000, (float(d.max()), upd, int((d > 1e-5).sum()))
            assert torch.allclose(st.bm, rec["b_final"], atol=1e-6, rtol=1e-5)
        assert (st.id_coef - rec["id_coef_final"]).norm(dim=1).max().item() <= 1e-6
        assert torch.allclose(st.id_emb, rec["id_emb_final"], atol=1e-5, rtol=1e-5)


# ------------------------------------------------------------------ BASELINE config 2's horizon (800 optimiser steps) run by the reference
def ref800_state(fx):
    """weights (re-drawn from the recorded seeds) + initial trainable / EMA state of tests/golden/horizon_mini_ref800.pt under the oracle's names"""
    c, sd = fx["cfg"], fx["cfg"]["seeds"]
    W = {}
    W.update(O.init_weights(O.unet_param_shapes(c["unet"], "model.diffusion_model."), seed=sd["unet"]))
    W.update(O.init_weights(O.vae_encoder_param_shapes(c["vae"], "first_stage_model."), seed=sd["vae"]))
    W.update(O.init_weights(O.clip_param_shapes(c["clip"], "cond_stage_model.transformer.text_model."), seed=sd["clip"]))
    W["cond_stage_model.transformer.text_model.embeddings.token_embedding.weight"] = O.formula_table(c["clip"]["vocab"], c["clip"]["width"])
    W.update(O.init_weights(O.frnet_param_shapes(c["fr"], "embedding_manager.meta_id_net.id_model."), seed=sd["fr"]))
    W["celeb_embeddings"] = O.init_weights({"celeb_embeddings": (c["es"], 1 + c["inner"], c["clip"]["width"])}, seed=sd["basis"])["celeb_embeddings"]
    W["embedding_manager.meta_id_net.stylegan_mlp.net.0.weight"] = fx["init"]["Wm"].clone()
    W["embedding_manager.meta_id_net.stylegan_mlp.net.0.bias"] = fx["init"]["bm"].clone()
    W["id_coefficients"] = fx["init"]["id_coef0"].clone(); W["id_embeddings"] = fx["init"]["id_emb0"].clone()
    return W


def learned_l2(a, b):
    """north_star's quantity: L2 distance of the learned coefficients (the EMA dictionary rows, embedding_manager.py:484-489), worst identity"""
    return (a - b).norm(dim=-1).max(dim=-1).values


def test_oracle_tracks_the_reference_for_800_steps(golden_dir):
    """tests/golden/horizon_mini_ref800.pt (oracle/gen_horizon_ref.py): 800 optimiser steps — aigc_id.yaml:135 max_steps — of the composed
    training step run by the REFERENCE's own code (the harness step_mini.pt pins for 3 steps), plus the same reference code with seeded
    noise on the FR-net features and the oracle on the same stream. (a) the oracle recomputed live agrees with the reference on the
    first 40 steps and with its own stored trajectory; (b) over all 800 steps the oracle's learned coefficients stay within 2e-6 L2
    of the reference's (measured 6e-7) — north_star's bar is 1e-3; (c) the reference perturbed by 3e-6 / 2.5e-5 / 2e-4 relative feature
    noise stays within 1e-5 / 5e-5 / 5e-4: at this width the trajectory is not chaotic, a distance above those numbers is an error."""
    from oracle.replay import ref_stream_batch
    fx = _load(golden_dir, "horizon_mini_ref800.pt")
    c, ref = fx["cfg"], fx["ref"]
    assert fx["steps"] == 800 and ref["id_coef"].shape[0] == 800
    W = ref800_state(fx)
    st = O.TrainState(W)
    lat = 64 // (2 ** (len(c["vae"]["ch_mult"]) - 1))
    for s in range(40):
        loss = O.train_step(W, st, ref_stream_batch(s, lat), c["lr"], c["unet"], c["vae"], c["clip"], c["fr"], es=c["es"], inner=c["inner"], scale=c["scale"])
        assert abs(float(loss) - float(ref["loss"][s])) <= 5e-6 * abs(float(ref["loss"][s])), (s, float(loss), float(ref["loss"][s]))
        assert float(learned_l2(st.id_coef, ref["id_coef"][s])) <= 2e-6, s
        assert float(learned_l2(st.id_coef, fx["oracle"]["id_coef"][s])) <= 5e-6, s           # the stored oracle trajectory is this oracle's (<= 1e-6 on its own host, 1.04e-6 measured on another)
    d = learned_l2(fx["oracle"]["id_coef"], ref["id_coef"])
    assert float(d.max()) <= 2e-6 and float(d[-1]) <= 2e-6, (float(d.max()), float(d[-1]))
    assert float(((fx["oracle"]["loss"] - ref["loss"]).abs() / ref["loss"].abs()).max()) <= 2e-5
    for tag, bar in (("ref_fr3e-6", 1e-5), ("ref_fr2.5e-5", 5e-5), ("ref_fr2e-4", 5e-4)):
        assert float(learned_l2(fx[tag]["id_coef"], ref["id_coef"]).max()) <= bar, tag


def test_fixture_host_warp_grid_digest_and_distance_to_this_host():
    """the SD-size oracle trajectories were generated on a host whose torch evaluates F.affine_grid(trans_matrix, 512 x 512) to the table
    oracle/replay.py::fixture_host_warp_grid restates (digest-pinned; the `-m gpu` horizon tests hand the library THAT table through
    cb_ctx_set_warp_grid). On any host it must be within one fp32 ulp of the host's own affine_grid — torch's K = 3 bmm rounds host
    dependently (tools/probes/probe_affine_grid_host.py) — and the oracle's face_warp_resize must use the reference's call."""
    import hashlib
    import torch.nn.functional as F
    from oracle import cb_oracle as O
    from oracle.replay import FIXTURE_HOST_GRID_SHA1, fixture_host_warp_grid
    g = fixture_host_warp_grid(512)
    assert g.shape == (512, 512, 2) and g.dtype == torch.float32
    assert hashlib.sha1(g.numpy().tobytes()).hexdigest()[:12] == FIXTURE_HOST_GRID_SHA1[512]
    here = F.affine_grid(torch.tensor([O.TRANS_MATRIX], dtype=torch.float32), size=[1, 3, 512, 512], align_corners=True)[0]
    assert (g - here).abs().max().item() <= 1.2e-7 + 1e-12, "the fixture host's grid is more than one ulp from this host's"
    # the corners: align_corners=True maps the output corners through the matrix exactly (up to rounding)
    t = torch.tensor(O.TRANS_MATRIX)
    for (y, x, u, v) in ((0, 0, -1.0, -1.0), (511, 511, 1.0, 1.0), (0, 511, 1.0, -1.0)):
        want = t @ torch.tensor([u, v, 1.0])
        assert torch.allclose(g[y, x], want, atol=3e-7), (y, x, g[y, x], want)
