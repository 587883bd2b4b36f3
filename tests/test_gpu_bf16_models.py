"""GPU parity tests of the bf16 path (BASELINE config 3: bf16 UNet / VAE / CLIP, fp32 coefficient head + gradient + AdamW),
graph level, through the C ABI on a bf16 context, against the fp32 CPU oracle on the same seeded weights and replay stream.

Tolerances. bf16 has 8 significant bits (ulp 2^-8 = 3.9e-3): every stored activation is rounded once, so an L-layer network
accumulates ~sqrt(L) * 2^-9 relative noise. Stated per check as max|hip - ref| <= tol * max|ref|:
  * network outputs (eps, conditioning, latents): 4e-2;
  * loss: 2e-2 relative;
  * the MLP gradient dW (what config 3 keeps in fp32 downstream of the bf16 networks): relative L2 error <= 6e-2 and cosine
    similarity >= 0.995 — it passes through the whole UNet + CLIP backward in bf16;
  * learned coefficients: north_star states its 1e-3 L2 bar for config 2 (fp32, "match reference coeffs"); for config 3 the
    bar at MINI width is COEF_BAR_BF16 = 4e-3 over 40 optimiser steps, and the drift is also reported relative to the distance the
    coefficients travelled (<= 2 %); at the REAL width the bars are the measured envelope of
    test_train_20_steps_sd14_bf16_bs4_horizon (learned coefficients 2.8e-3 .. 3.9e-3 after 20 steps at lr 2e-2). north_star's 1e-3 is
    REPORTED per sample by test_train_40_steps_mini_bf16_summation_order_samples (no expected-failure marker). Why not 1e-3: AdamW
    divides by sqrt(v), so an element's update is ~lr whatever the gradient's size, and the bf16 rounding noise of dW (relative L2
    1.5e-2, cosine 0.9999 at SD size) moves each of the 1024 coefficients by a few 1e-5 per step.
    The 40-step distance is a random variable of the summation order, not a constant: SEVEN samples (every GEMM / conv of the step pinned
    to another tiling, profiles/r05_bf16_mini40_summation_order_samples.json) measure 1.9e-3 .. 2.9e-3, mean 2.25e-3, none within 1e-3.
    The bar is 4e-3 again (round 4 had raised it to 8e-3 on a 4.8e-3 sample; that sample's excess came from the fp32 identity branch —
    the warp / resize arithmetic fixed in round 5 — not from bf16: the same plan now measures 2.2e-3)."""
import json
import os

import pytest
import torch

from test_gpu_models import MINI, cached_weights, close, dv, load, make_batch, oracle_first_step, run_horizon, trainer_cfg

pytestmark = pytest.mark.gpu
REPORT = {}
DEV = "cuda:0"
COEF_BAR_BF16 = 4e-3


@pytest.fixture(scope="module")
def cbf(built_lib):
    from celebbasis_amd import lib
    ctx = lib.Context(dtype="bf16")
    yield lib, ctx
    ctx.sync()
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/bf16_models_report.json", "w") as f:
        json.dump(REPORT, f, indent=1, sort_keys=True)


def rel_l2(a, b):
    a = a.detach().float().cpu().reshape(-1); b = b.detach().float().cpu().reshape(-1)
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def cosine(a, b):
    a = a.detach().float().cpu().reshape(-1); b = b.detach().float().cpu().reshape(-1)
    return (a @ b / (a.norm() * b.norm()).clamp_min(1e-30)).item()


def run_steps_bf16(lib, ctx, m, B, steps, lr, seed, taps, per_step=True, bar=None):
    from oracle import cb_oracle as O
    shapes = O.trainer_param_shapes(m["unet"], m["vae"], m["clip"], m["fr"], es=m["es"], inner=m["inner"], max_ids=m["max_ids"])
    W = cached_weights(O, shapes, seed) if m["image"] == 512 else O.init_weights(shapes, seed=seed)
    tr = lib.Trainer(ctx, lib.WeightTable(W), trainer_cfg(lib, m))
    st = O.TrainState(W)
    coef0 = st.id_coef.clone()
    lat = m["image"] // (2 ** (len(m["vae"]["ch_mult"]) - 1))
    tag = f"bf16.step[{m['image']}px,B{B}]"
    for s in range(steps):
        batch = make_batch(B, m["image"], lat, m["clip"]["vocab"], m["placeholder"], 100 + s, m["max_ids"])
        ctx.set_debug_taps(taps and s == 0)
        dev = {k: dv(v) for k, v in batch.items()}
        tr.step(dev["image"], dev["faces"], dev["ids"], dev["tokens"], dev["t"], dev["noise"], dev["post_noise"], lr)
        if steps == 1 and m["image"] == 512:       # the same oracle step as test_gpu_models' fp32 test of this batch size (computed once)
            loss_ref, aux, st = oracle_first_step(O, W, m, B, seed, lr, batch)
        else:
            loss_ref, aux = O.train_step(W, st, batch, lr, m["unet"], m["vae"], m["clip"], m["fr"], es=m["es"], inner=m["inner"],
                                         placeholder=m["placeholder"], return_aux=True)
        ctx.sync()
        if taps and s == 0:
            nhwc = lambda t: t.permute(0, 2, 3, 1).reshape(-1)
            close(tag + ".tap.z0", ctx.tap("z0"), nhwc(aux["z0"]), 4e-2)
            # the FR-net and the coefficient head stay fp32 in config 3: fp32-class agreement
            close(tag + ".tap.frnet.features", ctx.tap("frnet.features"), aux["feat"].reshape(-1), 2e-4)
            close(tag + ".tap.coef", ctx.tap("coef"), aux["coef"].reshape(-1), 1e-4)
            close(tag + ".tap.clip.cond", ctx.tap("clip.last_hidden_state"), aux["cond"].reshape(-1), 4e-2)
            close(tag + ".tap.unet.out", ctx.tap("unet.out"), nhwc(aux["eps_hat"]), 4e-2)
        if per_step:
            close(tag + f".s{s}.loss", tr.get_state("loss"), loss_ref.reshape(1), 2e-2)
            gW = tr.get_state("mlp.grad_weight")
            e, c = rel_l2(gW, aux["gW"]), cosine(gW, aux["gW"])
            REPORT[tag + f".s{s}.grad_weight"] = dict(rel_l2=e, cosine=c)
            assert e <= 6e-2 and c >= 0.995, f"{tag}.s{s}: dW rel L2 {e:.3e} / cosine {c:.5f} vs the fp32 oracle"
    coef_hip = tr.get_state("id_coefficients").reshape(m["max_ids"], -1)
    l2 = (coef_hip - st.id_coef).norm(dim=1).max().item()
    travelled = (st.id_coef - coef0).norm(dim=1).max().item()
    REPORT[tag + f".id_coefficients.L2[{steps} steps]"] = dict(l2=l2, travelled=travelled, relative=l2 / max(travelled, 1e-30))
    bar = COEF_BAR_BF16 if bar is None else bar
    assert l2 <= bar, f"{tag}: coefficient L2 distance {l2:.3e} exceeds the bar {bar:.0e}"
    assert steps < 10 or l2 <= 2e-2 * travelled, f"{tag}: coefficient drift {l2:.3e} vs travelled {travelled:.3e}"
    close(tag + ".id_embeddings", tr.get_state("id_embeddings"), st.id_emb.reshape(-1), 1e-3)
    assert int(tr.get_state("step").item()) == steps
    assert torch.isfinite(tr.get_state("mlp.weight")).all()
    tr.close()


def test_unet_mini_bf16_vs_oracle(cbf):
    from oracle import cb_oracle as O
    lib, ctx = cbf
    cfg = MINI["unet"]
    W = O.init_weights(O.unet_param_shapes(cfg), seed=3)
    u = lib.UNet(ctx, lib.WeightTable(W), lib.unet_cfg(**cfg))
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 4, 16, 16, generator=g); t = torch.tensor([417, 33]); c = torch.randn(2, 77, cfg["context_dim"], generator=g)
    dout = torch.randn(2, 4, 16, 16, generator=g)
    out = u.forward(dv(x), dv(t), dv(c), need_grad=True)        # public tensors stay fp32 on a bf16 context
    dctx = u.backward_context(dv(dout))
    ctx.sync()
    cr = c.clone().requires_grad_(True)
    ref = O.unet_forward(W, cfg, x, t, cr)
    (dref,) = torch.autograd.grad((ref * dout).sum(), [cr])
    close("bf16.unet_mini.out", out, ref, 4e-2)
    close("bf16.unet_mini.dcontext", dctx, dref, 6e-2)
    u.close()


def test_train_step_mini_bf16_vs_oracle(cbf):
    lib, ctx = cbf
    run_steps_bf16(lib, ctx, MINI, B=2, steps=3, lr=1e-2, seed=1, taps=True)


def test_train_40_steps_mini_bf16_coefficient_bar(cbf):
    """config 3's parity bar over a horizon: 40 optimiser steps in bf16 keep the learned coefficients within COEF_BAR_BF16 of
    the fp32 oracle on the same replay stream (and within 2 % of the distance they travelled); see the module docstring"""
    lib, ctx = cbf
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    run_steps_bf16(lib, ctx, MINI, B=2, steps=40, lr=5e-3, seed=7, taps=False, per_step=False)


SUMMATION_ORDER_PLANS = [(0, 0, 0), (64, 64, 0), (128, 128, 0), (128, 64, 0), (64, 128, 0), (64, 64, 2), (32, 80, 0)]


def test_train_40_steps_mini_bf16_summation_order_samples(cbf):
    """what COEF_BAR_BF16 rests on. The 40-step distance between the bf16 path's learned coefficients and the fp32 oracle's is a random
    variable of the summation order (AdamW turns the sign of a round-off-sized gradient element into a full +-lr step): the same run is
    repeated with every GEMM / conv of the step pinned to another tiling (cb_debug_force_plan: other k-split / accumulation orders in
    the bf16 UNet / VAE AND in the fp32 identity branch — all of them valid evaluations of the same arithmetic), 7 samples against ONE
    oracle trajectory. Every sample must stay inside COEF_BAR_BF16; the samples, their spread and which of them would meet north_star's
    fp32 bar of 1e-3 are written to gpurun_out/bf16_mini40_summation_order_samples.json (committed under profiles/). north_star states
    1e-3 for config 2 (fp32, "match reference coeffs"); a bf16 UNet does not meet it and this test REPORTS that instead of hiding it
    behind an expected-failure marker."""
    from oracle import cb_oracle as O
    lib, ctx = cbf
    m, B, steps, lr, seed = MINI, 2, 40, 5e-3, 7
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    shapes = O.trainer_param_shapes(m["unet"], m["vae"], m["clip"], m["fr"], es=m["es"], inner=m["inner"], max_ids=m["max_ids"])
    W = O.init_weights(shapes, seed=seed)
    lat = m["image"] // (2 ** (len(m["vae"]["ch_mult"]) - 1))
    batches = [make_batch(B, m["image"], lat, m["clip"]["vocab"], m["placeholder"], 100 + s, m["max_ids"]) for s in range(steps)]
    st = O.TrainState(W)
    coef0 = st.id_coef.clone()
    for b in batches:
        O.train_step(W, st, b, lr, m["unet"], m["vae"], m["clip"], m["fr"], es=m["es"], inner=m["inner"], placeholder=m["placeholder"])
    travelled = (st.id_coef - coef0).norm(dim=1).max().item()
    samples = []
    try:
        for plan in SUMMATION_ORDER_PLANS:
            lib.check(lib.load().cb_debug_force_plan(*plan))
            tr = lib.Trainer(ctx, lib.WeightTable(W), trainer_cfg(lib, m))
            for b in batches:
                dev = {k: dv(v) for k, v in b.items()}
                tr.step(dev["image"], dev["faces"], dev["ids"], dev["tokens"], dev["t"], dev["noise"], dev["post_noise"], lr)
            l2 = (tr.get_state("id_coefficients").reshape(m["max_ids"], -1) - st.id_coef).norm(dim=1).max().item()
            samples.append(dict(plan=list(plan), learned_coefficient_l2=l2, relative_to_travelled=l2 / travelled, within_north_star_1e_3=l2 <= 1e-3))
            tr.close()
    finally:
        lib.check(lib.load().cb_debug_force_plan(0, 0, 0))
    vals = [x["learned_coefficient_l2"] for x in samples]
    rec = dict(steps=steps, lr=lr, travelled=travelled, bar=COEF_BAR_BF16, samples=samples, max=max(vals), min=min(vals), mean=sum(vals) / len(vals),
               samples_within_north_star_1e_3=sum(x["within_north_star_1e_3"] for x in samples))
    REPORT["bf16.mini40.summation_order_samples"] = rec
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/bf16_mini40_summation_order_samples.json", "w") as f:
        json.dump(rec, f, indent=1)
    assert len(samples) >= 5 and max(vals) <= COEF_BAR_BF16, rec
    assert max(vals) <= 2e-2 * travelled, rec


def test_train_step_sd14_bf16_bs2_vs_oracle(cbf):
    """BASELINE config 3's arithmetic at SD v1-4 size against the LIVE oracle with every intermediate tap: 512x512, bf16 UNet / VAE +
    fp32 coefficient gradient, one full step at bs=2 (bs=4 through round 5: the live fp32 oracle costs 7 s per sample; config 3's own
    bs=4 shapes run in test_train_20_steps_sd14_bf16_bs4_horizon against the stored oracle trajectory, gradient rows included)"""
    from oracle import cb_oracle as O
    lib, ctx = cbf
    full = dict(unet=O.UNET_SD14, vae=O.VAE_SD14, clip=O.CLIP_L14, fr=O.FR_R100, es=2, inner=512, max_ids=10, image=512,
                placeholder=O.PLACEHOLDER_SKS)
    torch.set_num_threads(min(64, os.cpu_count() or 1))
    run_steps_bf16(lib, ctx, full, B=2, steps=1, lr=1e-2, seed=6, taps=True)      # seed / rate of test_train_step_sd14_bs2_vs_oracle: one oracle step for both


def test_train_100_steps_sd14_bf16_bs4_horizon(cbf, golden_dir):
    """config 3 beyond 20 steps (round-5 verdict: "config 3 has no horizon beyond 20"): the same stream continued to 100 optimiser steps
    against tests/golden/horizon_sd14_b4_100.pt (oracle/gen_horizon.py bf16long: fp32 oracle, bs=4, lr 2e-2; its first 20 steps are the
    20-step fixture's). The table (gpurun_out/horizon_sd14_b4_100_bf16.json) shows how the bf16 path's distance to the fp32 oracle
    grows with the step count; the asserted envelope is the measured one with margin, north_star's 1e-3 is reported, not met.
    A fixture still being generated is compared over the steps it holds (>= 40)."""
    lib, ctx = cbf
    path = os.path.join(golden_dir, "horizon_sd14_b4_100.pt")
    if not os.path.exists(path):
        pytest.skip("tests/golden/horizon_sd14_b4_100.pt not generated (python oracle/gen_horizon.py bf16long)")
    fx = dict(load(golden_dir, "horizon_sd14_b4_100.pt"))
    n = int(fx["steps"])
    if n < 40:
        pytest.skip(f"horizon_sd14_b4_100.pt holds {n} steps so far")
    fx20 = load(golden_dir, "horizon_sd14_b4.pt")
    assert torch.equal(fx["coef"][:20], fx20["coef"][:20]), "the 100-step trajectory must start with the 20-step fixture"
    fx["cfg"] = dict(fx20["cfg"], **fx["cfg"])
    # measured (round 6, 80 of the 100 steps): 2.2e-3 @ 17, 6.8e-3 @ 41, 1.85e-2 @ 80 — linear in the step count, as two trajectories that
    # differ by a fixed perturbation (the bf16 rounding of the frozen weights: tools/probes/probe_bf16_weight_floor.py) part; losses within 6e-4
    s = run_horizon(lib, ctx, fx, n, "horizon_sd14_b4_100_bf16", REPORT, dict(id_coef_l2_max=4e-2, loss_rel_max=5e-3))
    travelled = (fx["id_coef"][n - 1] - fx["id_coef"][0]).norm(dim=-1).max().item()          # learned rows, step 1 -> step n (20.2 at n = 100)
    REPORT["bf16.sd14.horizon100.relative_to_travelled"] = s["id_coef_l2_max"] / travelled
    assert s["id_coef_l2_max"] <= 3e-3 * travelled, (s, travelled)
    tab = REPORT["horizon_sd14_b4_100_bf16.table"]
    REPORT["bf16.sd14.horizon100.learned_coefficient_l2_at"] = {str(k): tab[k - 1]["id_coef_l2"] for k in (1, 10, 20, 40, 60, 80, 100) if k <= n}
    REPORT["bf16.sd14.horizon100.within_north_star_1e-3"] = s["id_coef_l2_max"] <= 1e-3


def test_train_20_steps_sd14_bf16_bs4_horizon(cbf, golden_dir):
    """BASELINE config 3 over a horizon AT FULL SIZE: 20 optimiser steps of SD v1-4 512x512 bs=4 (lr 2e-2) with the bf16 UNet / VAE
    against the fp32 oracle trajectory committed in tests/golden/horizon_sd14_b4.pt. The drift table (gpurun_out/horizon_sd14_b4_bf16.json,
    committed copy profiles/r03_sd14_bf16_horizon20.json) is the real-width evidence for config 3's bars.
    Measured (two runs, with and without the fp32 K/V-projection lever CB_BF16_KV_F32): dW of a step is within 0.7 - 2 % relative L2 of
    the oracle's; the FIRST AdamW step already turns that into a 7e-3 coefficient difference (every weight moves by lr * sign(g): the
    ~0.5 % of elements whose sign differs move the other way), the step coefficients are 2.4e-2 .. 4.5e-2 apart after 20 steps and
    the learned (EMA dictionary) coefficients 2.8e-3 .. 3.9e-3 (bar 5e-3); losses agree to 5e-4. north_star's 1e-3 bar is therefore NOT met by
    the bf16 path at real width (REPORT key below; the mini-width samples are in test_train_40_steps_mini_bf16_summation_order_samples);
    this test holds the path to its measured envelope with 2x margin so that a regression of the bf16 kernels is caught."""
    lib, ctx = cbf
    fx = load(golden_dir, "horizon_sd14_b4.pt")
    s = run_horizon(lib, ctx, fx, 20, "horizon_sd14_b4_bf16", REPORT, dict(coef_l2_max=9e-2, id_coef_l2_last=5e-3, loss_rel_max=5e-3))
    REPORT["bf16.sd14.learned_coefficients_within_north_star_1e-3"] = s["id_coef_l2_last"] <= 1e-3
    tab = REPORT["horizon_sd14_b4_bf16.table"]
    g1 = tab[0]["grad_weight_rel_l2_sampled_rows"]
    assert g1 <= 3e-2, f"first-step dW relative L2 {g1:.3e} (bf16 UNet backward) — measured 7e-3 .. 8e-3"


def test_trainer_step_b16_largest_batch_is_finite_and_deterministic(cbf):
    """the largest batch the trainer accepts (kMaxBatch = 16: 32 faces, a 16384-row head GEMM instead of the GEMV, 48 GiB of step arena in
    bf16) at SD size: two steps from the same state give the same loss bit for bit and finite gradients (no oracle at this size)"""
    from celebbasis_amd import synthetic
    lib, ctx = cbf
    state = synthetic.synthetic_state(synthetic.trainer_shapes(), seed=0)
    tr = lib.Trainer(ctx, lib.WeightTable(state))
    b = synthetic.synthetic_batch(16, step=0, device=DEV, identity=1)
    losses = []
    for _ in range(2):
        tr.set_state("mlp.weight", state["embedding_manager.meta_id_net.stylegan_mlp.net.0.weight"])
        tr.set_state("mlp.bias", state["embedding_manager.meta_id_net.stylegan_mlp.net.0.bias"])
        tr.set_state("step", torch.zeros(1))
        for n_, k in (("adam.m_weight", 1024 * 512), ("adam.v_weight", 1024 * 512), ("adam.m_bias", 1024), ("adam.v_bias", 1024)):
            tr.set_state(n_, torch.zeros(k))
        tr.step(b["image"], b["faces"], b["ids"], b["tokens"], b["t"], b["noise"], b["post_noise"], 5e-3 * 16)
        ctx.sync()
        losses.append(float(tr.get_state("loss")[0]))
        assert torch.isfinite(tr.get_state("mlp.weight")).all() and torch.isfinite(tr.get_state("mlp.grad_weight")).all()
    REPORT["bf16.step[512px,B16].loss"] = losses
    assert losses[0] == losses[1] and 0.5 < losses[0] < 3.0
    with pytest.raises(lib.CBError):
        b17 = synthetic.synthetic_batch(17, step=0, device=DEV, identity=1)
        tr.step(b17["image"], b17["faces"], b17["ids"], b17["tokens"], b17["t"], b17["noise"], b17["post_noise"], 5e-3)
    tr.close()


def test_sampler_mini_bf16_vs_oracle(cbf):
    """DDIM with guidance on a bf16 context (the reference samples under fp16 autocast, stable_txt2img.py:320-322): the sampler
    state stays fp32, the UNet runs in bf16; 4 steps against the fp32 oracle"""
    from oracle import cb_oracle as O
    lib, ctx = cbf
    cfg = MINI["unet"]
    W = O.init_weights(O.unet_param_shapes(cfg), seed=11)
    u = lib.UNet(ctx, lib.WeightTable(W), lib.unet_cfg(**cfg))
    g = torch.Generator().manual_seed(12)
    n = 2
    xT = torch.randn(n, 4, 16, 16, generator=g)
    cond = torch.randn(n, 77, cfg["context_dim"], generator=g); uncond = torch.randn(n, 77, cfg["context_dim"], generator=g)
    s = O.ddim_schedule(O.make_schedule()["alphas_cumprod"], 4, 0.0)
    x, _ = u.ddim_sample(dv(xT), dv(cond), dv(uncond), s["timesteps"], s["alphas"], s["alphas_prev"], s["sigmas"],
                         s["sqrt_one_minus_alphas"], scale=5.0)
    ctx.sync()
    ref = O.ddim_sample(W, cfg, xT, cond, uncond, s, scale=5.0)
    ref = ref[0] if isinstance(ref, (tuple, list)) else ref
    close("bf16.ddim_mini.x0", x, ref, 6e-2)
    u.close()
