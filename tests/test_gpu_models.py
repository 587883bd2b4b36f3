"""GPU parity tests, graph level, through the C ABI:
  * mini networks against the golden vectors written by the REFERENCE's own modules (tests/golden, oracle/gen_golden.py)
  * mini + SD-v1-4-size training steps against the CPU oracle on the same seeded weights and replay stream.
Tolerances are written per check as max|hip - ref| <= tol * max|ref| (fp32; summation order is the only difference).
The north_star bar for the learned coefficients is 1e-3 L2; these per-tensor bounds are far tighter."""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
REPORT = {}
DEV = "cuda:0"


@pytest.fixture(scope="module")
def cb(built_lib):
    from celebbasis_amd import lib
    ctx = lib.Context()
    yield lib, ctx
    ctx.sync()
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/models_report.json", "w") as f:
        json.dump(REPORT, f, indent=1, sort_keys=True)


def close(name, got, ref, tol):
    got = got.detach().float().cpu(); ref = ref.detach().float().cpu()
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    assert torch.isfinite(got).all(), f"{name}: non-finite output"
    err = (got - ref).abs().max().item()
    scale = max(ref.abs().max().item(), 1e-30)
    REPORT[name] = dict(max_abs_err=err, ref_max=scale, rel=err / scale, tol=tol)
    assert err <= tol * scale, f"{name}: max abs err {err:.3e} vs ref max {scale:.3e} (rel {err / scale:.3e} > {tol})"


def dv(t):
    """device copy on the context's stream. Temporaries may be dropped right after a native call: every entry point orders its
    reads on the context stream (cb_trainer_prefetch is the exception, handled by record_stream in lib.Trainer.prefetch)"""
    return t.contiguous().to(DEV)


def load(golden_dir, name):
    return torch.load(os.path.join(golden_dir, name), weights_only=False)


_W_CACHE = {}


_W_AHEAD = {}


def _w_key(shapes, seed):
    return (len(shapes), sum(int(torch.Size(v).numel()) for v in shapes.values()), seed)


def weights_ahead(O, shapes, seed):
    """start O.init_weights(shapes, seed) on a host thread: the 800-step runs are GPU-bound, and the next draw's frozen weights (another
    seed: 7 s of host time on the GPU box, 40 s here) can be generated under them. cached_weights picks the result up."""
    from concurrent.futures import ThreadPoolExecutor
    key = _w_key(shapes, seed)
    if key not in _W_CACHE and key not in _W_AHEAD:
        ex = ThreadPoolExecutor(max_workers=1)
        _W_AHEAD[key] = (ex, ex.submit(O.init_weights, shapes, seed=seed))


def cached_weights(O, shapes, seed):
    """O.init_weights(shapes, seed) — 40 s of host time for the 1.08e9 SD-size parameters — kept for the two most recent (size, seed) pairs:
    several SD-size tests share seed 8. The tensors are read-only for every user (the trainers copy them to the device)."""
    key = _w_key(shapes, seed)
    if key not in _W_CACHE:
        while len(_W_CACHE) >= 2:
            _W_CACHE.pop(next(iter(_W_CACHE)))
        if key in _W_AHEAD:
            ex, fut = _W_AHEAD.pop(key)
            _W_CACHE[key] = fut.result()
            ex.shutdown(wait=False)
        else:
            _W_CACHE[key] = O.init_weights(shapes, seed=seed)
    else:
        _W_CACHE[key] = _W_CACHE.pop(key)          # most recently used last
    return _W_CACHE[key]


def test_unet_mini_vs_reference_golden(cb, golden_dir):
    from oracle import cb_oracle as O
    lib, ctx = cb
    fx = load(golden_dir, "unet_mini.pt")
    cfg = fx["cfg"]
    W = O.init_weights(O.unet_param_shapes(cfg), seed=fx["weight_seed"])
    wt = lib.WeightTable(W)
    u = lib.UNet(ctx, wt, lib.unet_cfg(**cfg))
    out = u.forward(dv(fx["x"]), dv(fx["t"]), dv(fx["context"]), need_grad=True)
    dctx = u.backward_context(dv(fx["dout"]))
    ctx.sync()
    close("unet_mini.out", out, fx["out"], 5e-5)
    close("unet_mini.dcontext", dctx, fx["dcontext"], 1e-4)


def test_vae_mini_vs_reference_golden(cb, golden_dir):
    from oracle import cb_oracle as O
    lib, ctx = cb
    fx = load(golden_dir, "vae_mini.pt")
    W = O.init_weights(O.vae_encoder_param_shapes(fx["cfg"]), seed=fx["weight_seed"])
    v = lib.VAE(ctx, lib.WeightTable(W), lib.vae_cfg(**fx["cfg"]))
    img = dv(fx["x"].permute(0, 2, 3, 1))
    mom, z = v.encode(img, dv(fx["noise"]), 0.18215)
    ctx.sync()
    close("vae_mini.moments", mom.permute(0, 3, 1, 2), fx["moments"], 5e-5)
    close("vae_mini.z", z, fx["z_scaled"], 5e-5)


def test_frnet_mini_vs_reference_golden(cb, golden_dir):
    from oracle import cb_oracle as O
    lib, ctx = cb
    fx = load(golden_dir, "frnet_mini.pt")
    c = fx["cfg"]
    W = O.init_weights(O.frnet_param_shapes(c), seed=fx["weight_seed"])
    f = lib.FrNet(ctx, lib.WeightTable(W), layers=c["layers"], width=c["width"], in_size=c["in_size"], feat=c["feat"])
    out = f.forward(dv(fx["x"].permute(0, 2, 3, 1)))
    ctx.sync()
    close("frnet_mini.features", out, fx["out"], 1e-4)


@pytest.mark.parametrize("inlaunch", [1, 0])
@pytest.mark.parametrize("plan", [(16, 80, 2), (32, 80, 3), (64, 80, 1), (64, 64, 2), (64, 64, 4), (128, 128, 1)])
def test_frnet_mini_fused_epilogues_under_pinned_plans(cb, golden_dir, plan, inlaunch):
    """the FR backbone's fused conv epilogues (PReLU on the output; the next block's BatchNorm as a second output) on every route a plan can
    send them through: the 80-column tiles with in-block k-groups, a cross-block split-K whose last-arriving block (in-launch reduction,
    the default) or whose REDUCE kernel (cb_debug_splitk_inlaunch(0)) then applies the epilogue, the 64x64 and 128x128 tiles — pinned with
    cb_debug_force_plan (a layer the pinned tile is not eligible for keeps the planner's choice) and held to the reference's golden
    features (round-4 advisor finding: the epilogue had no test under forced split-K / 80-column tiles)."""
    from oracle import cb_oracle as O
    lib, ctx = cb
    if not inlaunch and plan[2] == 1:
        pytest.skip("no split-K in this plan: one route")
    fx = load(golden_dir, "frnet_mini.pt")
    c = fx["cfg"]
    W = O.init_weights(O.frnet_param_shapes(c), seed=fx["weight_seed"])
    f = lib.FrNet(ctx, lib.WeightTable(W), layers=c["layers"], width=c["width"], in_size=c["in_size"], feat=c["feat"])
    lib.check(lib.load().cb_debug_force_plan(*plan))
    lib.check(lib.load().cb_debug_splitk_inlaunch(inlaunch))
    ctx.profile_begin()
    try:
        out = f.forward(dv(fx["x"].permute(0, 2, 3, 1)))
        ctx.sync()
    finally:
        lib.check(lib.load().cb_debug_force_plan(0, 0, 0))
        lib.check(lib.load().cb_debug_splitk_inlaunch(1))
        prof = ctx.profile_end()
        ran = [r["kernel"] for r in prof]
    close(f"frnet_mini.features[plan {plan}, inlaunch {inlaunch}]", out, fx["out"], 1e-4)
    n_reduce = sum(r["launches"] for r in prof if r["kernel"].startswith("splitk_reduce"))
    n_regstaged = sum(r["launches"] for r in prof if r["kernel"].startswith("gemm_kernel<"))     # the 3-channel stem conv: its split-K keeps the reduce launch
    if plan[2] > 1 and not inlaunch:
        assert n_reduce > n_regstaged, prof
    else:
        assert n_reduce <= n_regstaged, prof
    REPORT[f"frnet_mini.pinned_plan{plan}.inlaunch{inlaunch}.kernels"] = sorted(set(ran))


def test_frnet_mini_without_the_lds_dma_kernels(golden_dir):
    """CB_F32_GLDS=0 (the documented A/B switch, read once per process) sends every conv the register-staged kernel can run to it; that
    kernel has no PReLU / second-output epilogue, so the FR backbone's fused epilogues must run DE-FUSED there (plain conv + the
    per-channel passes) instead of failing — and give the same features as the golden (round-4 advisor finding). Own process: the
    switch is latched at first use."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, torch; sys.path.insert(0, %r)\n"
        "from celebbasis_amd import lib\n"
        "from oracle import cb_oracle as O\n"
        "fx = torch.load(%r, weights_only=False); c = fx['cfg']\n"
        "W = O.init_weights(O.frnet_param_shapes(c), seed=fx['weight_seed'])\n"
        "ctx = lib.Context(); ctx.reset_stats()\n"
        "f = lib.FrNet(ctx, lib.WeightTable(W), layers=c['layers'], width=c['width'], in_size=c['in_size'], feat=c['feat'])\n"
        "out = f.forward(fx['x'].permute(0, 2, 3, 1).contiguous().cuda()); ctx.sync()\n"
        "err = (out.cpu() - fx['out']).abs().max().item() / fx['out'].abs().max().item()\n"
        "print('FRNET_NOGLDS', err, ctx.stats()['launches'])\n" % (root, os.path.join(golden_dir, "frnet_mini.pt")))
    outs = {}
    for mode in ("0", "1"):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=root, env=dict(os.environ, CB_F32_GLDS=mode))
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        line = [l for l in r.stdout.splitlines() if l.startswith("FRNET_NOGLDS")][-1].split()
        outs[mode] = (float(line[1]), int(line[2]))
        assert outs[mode][0] <= 1e-4, (mode, outs)
    assert outs["0"][1] > outs["1"][1], f"CB_F32_GLDS=0 did not change the launch count: the switch is not exercised ({outs})"
    REPORT["frnet_mini.without_lds_dma"] = dict(rel_err=outs["0"][0], launches=outs["0"][1], launches_fused=outs["1"][1])


def test_clip_mini_vs_hf_golden_and_backward(cb, golden_dir):
    from oracle import cb_oracle as O
    lib, ctx = cb
    fx = load(golden_dir, "clip_mini.pt")
    cfg, pre = fx["cfg"], fx["weight_prefix"]
    W = O.init_weights(O.clip_param_shapes(cfg, pre), seed=fx["weight_seed"])
    k = lib.Clip(ctx, lib.WeightTable(W), lib.clip_cfg(**cfg), prefix=pre)
    tokens = dv(fx["tokens"])
    emb = k.embed_tokens(tokens)
    close("clip_mini.token_embedding", emb, W[pre + "embeddings.token_embedding.weight"][fx["tokens"]], 0.0)
    hidden = k.forward(emb, need_grad=True)
    dh = torch.randn(hidden.shape, generator=torch.Generator().manual_seed(3))
    de = k.backward_embeds(dv(dh))
    ctx.sync()
    close("clip_mini.last_hidden_state", hidden, fx["last_hidden_state"], 5e-5)
    e = emb.cpu().clone().requires_grad_(True)
    ref = O.clip_encode(W, cfg, e, pre)
    (de_ref,) = torch.autograd.grad((ref * dh).sum(), [e])
    close("clip_mini.dembedded", de, de_ref, 1e-4)


MINI = dict(
    unet=dict(in_channels=4, out_channels=4, model_channels=32, num_res_blocks=1, channel_mult=(1, 2, 2), attention_resolutions=(1, 2),
              num_heads=2, context_dim=64),
    vae=dict(ch=32, ch_mult=(1, 2, 2), num_res_blocks=1, in_channels=3, z_channels=4),
    clip=dict(vocab=120, width=64, layers=2, heads=4, ctx_len=77, mlp=128),
    fr=dict(layers=(1, 1, 1, 1), width=16, in_size=32, feat=512),
    es=2, inner=16, max_ids=3, image=64, placeholder=99)


from oracle.replay import make_batch  # noqa: E402  (the seeded replay stream, shared with oracle/gen_horizon.py)


def trainer_cfg(lib, m):
    c = lib.trainer_cfg_default()
    c.unet = lib.unet_cfg(**m["unet"]); c.vae = lib.vae_cfg(**m["vae"]); c.clip = lib.clip_cfg(**m["clip"])
    for i, v in enumerate(m["fr"]["layers"]):
        c.fr_layers[i] = v
    c.fr_width, c.fr_in = m["fr"]["width"], m["fr"]["in_size"]
    c.image_size, c.es, c.inner_dim, c.max_ids = m["image"], m["es"], m["inner"], m["max_ids"]
    c.placeholder_token = m["placeholder"]
    return c


_ORACLE_STEP1 = {}


def oracle_first_step(O, W, m, B, seed, lr, batch):
    """the CPU oracle's first optimiser step from the seeded weights at SD size, computed once per (batch size, seed, rate): the fp32 and the
    bf16 one-step tests at bs=2 are held against the SAME oracle step (14 s of host time each through round 5). Returns (loss, aux,
    a private copy of the state after the step)."""
    import copy
    key = (m["image"], B, seed, lr)
    if key not in _ORACLE_STEP1:
        st = O.TrainState(W)
        loss_ref, aux = O.train_step(W, st, batch, lr, m["unet"], m["vae"], m["clip"], m["fr"], es=m["es"], inner=m["inner"],
                                     placeholder=m["placeholder"], return_aux=True)
        _ORACLE_STEP1[key] = (loss_ref, aux, st)
    loss_ref, aux, st = _ORACLE_STEP1[key]
    return loss_ref, aux, copy.deepcopy(st)


def run_steps(lib, ctx, m, B, steps, lr, seed, check_aux, per_step=True, drift=None):
    from oracle import cb_oracle as O
    shapes = O.trainer_param_shapes(m["unet"], m["vae"], m["clip"], m["fr"], es=m["es"], inner=m["inner"], max_ids=m["max_ids"])
    W = cached_weights(O, shapes, seed)
    tr = lib.Trainer(ctx, lib.WeightTable(W), trainer_cfg(lib, m))
    st = O.TrainState(W)
    lat = m["image"] // (2 ** (len(m["vae"]["ch_mult"]) - 1))
    tag = f"step[{m['image']}px,B{B}]"
    for s in range(steps):
        batch = make_batch(B, m["image"], lat, m["clip"]["vocab"], m["placeholder"], 100 + s, m["max_ids"])
        ctx.set_debug_taps(check_aux and s == 0)
        dev = {k: dv(v) for k, v in batch.items()}
        tr.step(dev["image"], dev["faces"], dev["ids"], dev["tokens"], dev["t"], dev["noise"], dev["post_noise"], lr)
        if steps == 1 and m["image"] == 512:
            loss_ref, aux, st = oracle_first_step(O, W, m, B, seed, lr, batch)
        else:
            loss_ref, aux = O.train_step(W, st, batch, lr, m["unet"], m["vae"], m["clip"], m["fr"], es=m["es"], inner=m["inner"],
                                         placeholder=m["placeholder"], return_aux=True)
        ctx.sync()
        if check_aux and s == 0:
            nhwc = lambda t: t.permute(0, 2, 3, 1).reshape(-1)
            close(tag + ".tap.vae.moments", ctx.tap("vae.moments"), nhwc(aux["moments"]), 1e-4)
            close(tag + ".tap.z0", ctx.tap("z0"), nhwc(aux["z0"]), 1e-4)
            close(tag + ".tap.frnet.features", ctx.tap("frnet.features"), aux["feat"].reshape(-1), 2e-4)
            close(tag + ".tap.coef", ctx.tap("coef"), aux["coef"].reshape(-1), 1e-4)
            close(tag + ".tap.clip.cond", ctx.tap("clip.last_hidden_state"), aux["cond"].reshape(-1), 1e-4)
            close(tag + ".tap.unet.out", ctx.tap("unet.out"), nhwc(aux["eps_hat"]), 2e-4)
        if per_step:          # elementwise relative error of the MLP gradient, worst step (see the weight check at the end)
            dlt = (tr.get_state("mlp.grad_weight").reshape(aux["gW"].shape) - aux["gW"]).abs() / aux["gW"].abs().clamp_min(1e-30)
            g_delta = dlt if s == 0 else torch.maximum(g_delta, dlt)
        if per_step:
            close(tag + f".s{s}.loss", tr.get_state("loss"), loss_ref.reshape(1), 1e-4)
            close(tag + f".s{s}.grad_weight", tr.get_state("mlp.grad_weight"), aux["gW"].reshape(-1), 1e-3)
            close(tag + f".s{s}.grad_bias", tr.get_state("mlp.grad_bias"), aux["gb"].reshape(-1), 1e-3)
        if drift is not None:                    # error as a function of the step count (one row per optimiser step)
            rel = lambda a, b: ((a - b).norm() / b.norm().clamp_min(1e-30)).item()
            W0_ = W["embedding_manager.meta_id_net.stylegan_mlp.net.0.weight"].reshape(-1)
            drift.append(dict(step=s + 1, loss_ref=float(loss_ref), loss_rel=abs(float(tr.get_state("loss")[0]) - float(loss_ref)) / abs(float(loss_ref)),
                              grad_weight_rel_l2=rel(tr.get_state("mlp.grad_weight"), aux["gW"].reshape(-1)),
                              weight_delta_rel_l2=rel(tr.get_state("mlp.weight") - W0_, st.Wm.reshape(-1) - W0_),
                              coef_l2=(tr.get_state("id_coefficients").reshape(m["max_ids"], -1) - st.id_coef).norm(dim=1).max().item()))
    # end state. AdamW normalises the gradient, so compare the parameter DELTA, and the north_star's L2 bar on coefficients.
    W0 = W["embedding_manager.meta_id_net.stylegan_mlp.net.0.weight"].reshape(-1)
    dW_hip = tr.get_state("mlp.weight") - W0
    dW_ref = st.Wm.reshape(-1) - W0
    if per_step:
        # relative L2: after one AdamW step an element's delta is lr * g / (|g| + eps), so the handful of elements whose gradient
        # is ~eps legitimately differ by a fraction of lr (max-abs over 524 288 elements is dominated by them)
        dl2 = ((dW_hip - dW_ref).norm() / dW_ref.norm().clamp_min(1e-30)).item()
        REPORT[tag + ".mlp.weight_delta"] = dict(rel_l2=dl2, max_abs_err=(dW_hip - dW_ref).abs().max().item(), ref_max=dW_ref.abs().max().item())
        assert dl2 <= 2e-2, f"{tag}: AdamW weight delta rel L2 {dl2:.3e}"
    if per_step:
        # a few steps: an element's update is lr * m / (sqrt(v) + eps) — bounded by ~lr per step whatever the gradient's size, and as
        # sensitive to the RELATIVE error of that element's gradient as 1 : 1. So every element is held to its own bound
        #     |w_hip - w_ref| <= 1e-4 * max|w|  +  lr * steps * min(2, 4 * delta),   delta = max over steps of |g_hip - g_ref| / |g_ref|
        # (fp32 bar on well-conditioned elements; an element whose gradient is itself round-off sized, delta ~ 1, may differ by a
        # sign, i.e. by at most 2 lr per step). The fraction of ill-conditioned elements (delta > 0.25) is reported.
        w_hip, w_ref = tr.get_state("mlp.weight"), st.Wm.reshape(-1)
        err = (w_hip - w_ref).abs()
        scale = w_ref.abs().max().item()
        bound = 1e-4 * scale + lr * steps * (4.0 * g_delta.reshape(-1)).clamp(max=2.0)
        worst = (err / bound).max().item()
        REPORT[tag + ".mlp.weight"] = dict(max_abs_err=err.max().item(), worst_err_over_bound=worst, ref_max=scale, lr=lr,
                                           ill_conditioned_fraction=(g_delta > 0.25).float().mean().item(),
                                           max_abs_err_well_conditioned=err[g_delta.reshape(-1) <= 1e-3].max().item())
        assert worst <= 1.0, f"{tag}.mlp.weight: an element is {worst:.2f}x over its conditioning-aware bound"
    else:
        close(tag + ".mlp.weight", tr.get_state("mlp.weight"), st.Wm.reshape(-1), 1e-3)
    close(tag + ".mlp.bias", tr.get_state("mlp.bias"), st.bm.reshape(-1), 1e-3 if per_step else 2e-2)
    coef_hip = tr.get_state("id_coefficients").reshape(m["max_ids"], -1)
    l2 = (coef_hip - st.id_coef).norm(dim=1).max().item()
    REPORT[tag + ".id_coefficients.L2"] = l2
    assert l2 <= 1e-3, f"{tag}: coefficient L2 distance {l2:.3e} exceeds the north_star bar 1e-3"
    close(tag + ".id_embeddings", tr.get_state("id_embeddings"), st.id_emb.reshape(-1), 1e-4 if per_step else 1e-3)
    assert int(tr.get_state("step").item()) == steps
    tr.close()


def test_train_step_mini_vs_oracle(cb):
    lib, ctx = cb
    run_steps(lib, ctx, MINI, B=2, steps=3, lr=1e-2, seed=1, check_aux=True)


def test_train_40_steps_mini_coefficient_bar(cb):
    """the north_star bar over a longer horizon: 40 optimiser steps (bs=2, replay stream) — coefficient dictionaries within
    1e-3 L2 of the CPU oracle, MLP weights within 1e-3 of their scale (AdamW amplifies round-off in tiny gradients, so the
    per-step gradient checks of the short runs are not repeated here)"""
    lib, ctx = cb
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    run_steps(lib, ctx, MINI, B=2, steps=40, lr=5e-3, seed=7, check_aux=False, per_step=False)


def test_train_step_mini_b1_vs_oracle(cb):
    lib, ctx = cb
    run_steps(lib, ctx, MINI, B=1, steps=2, lr=5e-3, seed=2, check_aux=False)


def test_unet_sd14_vs_oracle(cb):
    """SD v1-4 sized UNet (aigc_id.yaml:39-54), bs=1, 64x64 latent: eps and d/d context against the CPU oracle"""
    from oracle import cb_oracle as O
    lib, ctx = cb
    cfg = O.UNET_SD14
    W = O.init_weights(O.unet_param_shapes(cfg), seed=3)
    u = lib.UNet(ctx, lib.WeightTable(W), lib.unet_cfg(**cfg))
    g = torch.Generator().manual_seed(4)
    x = torch.randn(1, 4, 64, 64, generator=g); t = torch.tensor([417]); c = torch.randn(1, 77, 768, generator=g)
    dout = torch.randn(1, 4, 64, 64, generator=g)
    out = u.forward(dv(x), dv(t), dv(c), need_grad=True)
    dctx = u.backward_context(dv(dout))
    ctx.sync()
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    cr = c.clone().requires_grad_(True)
    ref = O.unet_forward(W, cfg, x, t, cr)
    (dref,) = torch.autograd.grad((ref * dout).sum(), [cr])
    close("unet_sd14.out", out, ref, 2e-4)
    close("unet_sd14.dcontext", dctx, dref, 5e-4)
    u.close()


def test_train_step_sd14_vs_oracle(cb):
    """BASELINE config 2 shapes: SD v1-4, 512x512, bs=1, fp32 — one full step against the oracle (loss, dW, db, EMA)"""
    from oracle import cb_oracle as O
    lib, ctx = cb
    full = dict(unet=O.UNET_SD14, vae=O.VAE_SD14, clip=O.CLIP_L14, fr=O.FR_R100, es=2, inner=512, max_ids=10, image=512,
                placeholder=O.PLACEHOLDER_SKS)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    run_steps(lib, ctx, full, B=1, steps=1, lr=5e-3, seed=5, check_aux=True)


def _graph_vs_eager(lib, ctx, m, seed, script):
    """run `script` = [(B, lr), ...] twice on fresh trainers — eager launches / HIP-graph replay — and return both end states + the graph counters"""
    from oracle import cb_oracle as O
    shapes = O.trainer_param_shapes(m["unet"], m["vae"], m["clip"], m["fr"], es=m["es"], inner=m["inner"], max_ids=m["max_ids"])
    W = cached_weights(O, shapes, seed)
    lat = m["image"] // (2 ** (len(m["vae"]["ch_mult"]) - 1))
    out = {}
    ctx.set_debug_taps(False)                    # an earlier test of the module may have left them on: a step that collects taps runs eagerly
    for graph in (False, True):
        tr = lib.Trainer(ctx, lib.WeightTable(W), trainer_cfg(lib, m))
        tr.set_graph_replay(graph)
        losses = []
        for s, (B, lr) in enumerate(script):
            dev = {k: dv(v) for k, v in make_batch(B, m["image"], lat, m["clip"]["vocab"], m["placeholder"], 100 + s, m["max_ids"]).items()}
            tr.step(dev["image"], dev["faces"], dev["ids"], dev["tokens"], dev["t"], dev["noise"], dev["post_noise"], lr)
            del dev                              # the batch may be freed right after the call: the graph reads its staging copy
            losses.append(float(tr.get_state("loss")[0]))
        out[graph] = dict(W=tr.get_state("mlp.weight").clone(), b=tr.get_state("mlp.bias").clone(), idc=tr.get_state("id_coefficients").clone(),
                          ide=tr.get_state("id_embeddings").clone(), gW=tr.get_state("mlp.grad_weight").clone(), losses=losses, stats=tr.graph_stats())
        tr.close()
    return out


def test_graph_replay_is_bitwise_the_eager_step_mini(cb):
    """SURVEY §8 B3 `cb_train_step (whole step, HIP-graph replay)`: the captured graph holds the same kernels in the same order on the same
    arena addresses, so a run that replays it must end in BITWISE the state of the eager run — across a change of batch size (first step
    at a size runs eagerly, the second captures) and of learning rate (one graph per (B, lr))."""
    lib, ctx = cb
    script = [(1, 5e-3)] * 6 + [(2, 1e-2)] * 4 + [(1, 5e-3)] * 2 + [(1, 2.5e-3)] * 3
    out = _graph_vs_eager(lib, ctx, MINI, 3, script)
    e, g = out[False], out[True]
    assert e["stats"] == dict(captures=0, replays=0), e["stats"]
    # eager: the first step at B=1, the first at B=2 and the first back at B=1 (each re-plans the step arena; the B=2 arena is larger, so
    # the B=1 graph captured before it is dropped and re-captured afterwards). One graph per batch size — the learning rate is read from
    # a device scalar (round 6; one graph per (B, lr) before) —: captures B=1, B=2 and, if the arena moved, B=1 again; a capture step
    # launches its graph too (at mini size the arena's slack covers B=2 and nothing is re-allocated: 2 captures)
    assert g["stats"]["captures"] in (2, 3) and g["stats"]["replays"] == len(script) - 3, g["stats"]
    assert g["losses"] == e["losses"], (g["losses"], e["losses"])
    for k in ("W", "b", "idc", "ide", "gW"):
        assert torch.equal(g[k], e[k]), f"graph replay changed {k}: max diff {(g[k] - e[k]).abs().max().item():.3e}"
    REPORT["graph_replay.mini"] = g["stats"]


def test_graph_replay_drops_graphs_captured_under_another_warp_table_or_plan(cb):
    """round-5 advisor finding: a captured step holds the context's warp-grid device pointer and the GEMM plans in force as baked-in
    kernel arguments / nodes. cb_ctx_set_warp_grid and the cb_debug_* switches bump generation stamps; a graph captured under other
    stamps is dropped and re-captured at its next use — the replaying trainer stays BITWISE on the eager one across both changes."""
    from oracle import cb_oracle as O
    lib, ctx = cb
    m = MINI
    shapes = O.trainer_param_shapes(m["unet"], m["vae"], m["clip"], m["fr"], es=m["es"], inner=m["inner"], max_ids=m["max_ids"])
    W = cached_weights(O, shapes, 5)
    lat = m["image"] // (2 ** (len(m["vae"]["ch_mult"]) - 1))
    ctx.set_debug_taps(False)
    base = ctx.warp_grid_host.clone() if getattr(ctx, "warp_grid_host", None) is not None and ctx.warp_grid_host.shape[0] == m["image"] else None
    out = {}
    try:
        for graph in (False, True):
            tr = lib.Trainer(ctx, lib.WeightTable(W), trainer_cfg(lib, m))
            grid0 = ctx.warp_grid_host.clone()
            tr.set_graph_replay(graph)
            for s in range(9):
                if s == 3:
                    ctx.set_warp_grid(grid0 + 1e-3)                      # another table (a shifted crop): frees / rewrites the device buffer
                if s == 6:
                    lib.check(lib.load().cb_debug_force_plan(64, 64, 1))  # other kernels for every GEMM the pin binds
                dev = {k: dv(v) for k, v in make_batch(1, m["image"], lat, m["clip"]["vocab"], m["placeholder"], 500 + s, m["max_ids"]).items()}
                tr.step(dev["image"], dev["faces"], dev["ids"], dev["tokens"], dev["t"], dev["noise"], dev["post_noise"], 5e-3 * (1 + s % 2))
                del dev
            out[graph] = dict(W=tr.get_state("mlp.weight").clone(), idc=tr.get_state("id_coefficients").clone(), loss=float(tr.get_state("loss")[0]),
                              stats=tr.graph_stats())
            lib.check(lib.load().cb_debug_force_plan(0, 0, 0))
            ctx.set_warp_grid(grid0)
            tr.close()
    finally:
        lib.check(lib.load().cb_debug_force_plan(0, 0, 0))
        if base is not None:
            ctx.set_warp_grid(base)
    e, g = out[False], out[True]
    assert g["stats"]["captures"] == 3 and g["stats"]["replays"] == 8, g["stats"]        # one capture per stamp set; two learning rates, one graph
    assert g["loss"] == e["loss"] and torch.equal(g["W"], e["W"]) and torch.equal(g["idc"], e["idc"])


def test_graph_replay_is_bitwise_the_eager_step_sd14(cb):
    """the same at SD v1-4 size, bs=1: 1 eager + 1 capture + 4 replays of the ~1 650-launch, two-stream step"""
    from oracle import cb_oracle as O
    lib, ctx = cb
    out = _graph_vs_eager(lib, ctx, SD14(O), 8, [(1, 5e-3)] * 6)
    e, g = out[False], out[True]
    assert g["stats"] == dict(captures=1, replays=5), g["stats"]
    assert g["losses"] == e["losses"], (g["losses"], e["losses"])
    for k in ("W", "b", "idc", "ide", "gW"):
        assert torch.equal(g[k], e[k]), f"graph replay changed {k}: max diff {(g[k] - e[k]).abs().max().item():.3e}"
    REPORT["graph_replay.sd14"] = g["stats"]


def _tail_run(lib, ctx, m, seed, script, overlap):
    """`script` = [(B, lr), ...] on a fresh trainer with the step's tail deferred to the side stream (`overlap`) or joined; nothing is read
    back between the steps (the loss goes to a device tensor in stream order), so consecutive steps really overlap"""
    from oracle import cb_oracle as O
    shapes = O.trainer_param_shapes(m["unet"], m["vae"], m["clip"], m["fr"], es=m["es"], inner=m["inner"], max_ids=m["max_ids"])
    W = cached_weights(O, shapes, seed)
    lat = m["image"] // (2 ** (len(m["vae"]["ch_mult"]) - 1))
    ctx.set_debug_taps(False)
    tr = lib.Trainer(ctx, lib.WeightTable(W), trainer_cfg(lib, m))
    tr.set_tail_overlap(overlap)
    losses = []
    for s, (B, lr) in enumerate(script):
        dev = {k: dv(v) for k, v in make_batch(B, m["image"], lat, m["clip"]["vocab"], m["placeholder"], 300 + s, m["max_ids"]).items()}
        tr.step(dev["image"], dev["faces"], dev["ids"], dev["tokens"], dev["t"], dev["noise"], dev["post_noise"], lr)
        del dev                                  # freed in context-stream order right after the call: the tail reads no caller memory
        losses.append(tr.loss_tensor(torch.device("cuda:0")))
    out = dict(W=tr.get_state("mlp.weight").clone(), b=tr.get_state("mlp.bias").clone(), mW=tr.get_state("adam.m_weight").clone(),
               vW=tr.get_state("adam.v_weight").clone(), idc=tr.get_state("id_coefficients").clone(), ide=tr.get_state("id_embeddings").clone(),
               gW=tr.get_state("mlp.grad_weight").clone(), step=float(tr.get_state("step")[0]), losses=[float(l) for l in losses])
    tr.close()
    return out


def test_deferred_tail_is_bitwise_the_joined_step_mini(cb):
    """cb_trainer_set_tail_overlap (default on): the CLIP input-gradient chain, the head's backward and AdamW of step i run on the side
    stream beside step i+1's VAE encode and are joined lazily. Same kernels, same order, disjoint arena regions: the end state after
    back-to-back steps — across a change of batch size, which re-plans the arena and must join first — is BITWISE the joined form's."""
    lib, ctx = cb
    script = [(1, 5e-3)] * 6 + [(2, 1e-2)] * 3 + [(1, 5e-3)] * 3
    a = _tail_run(lib, ctx, MINI, 4, script, True)
    b = _tail_run(lib, ctx, MINI, 4, script, False)
    assert a["step"] == b["step"] == len(script)
    assert a["losses"] == b["losses"], (a["losses"], b["losses"])
    for k in ("W", "b", "mW", "vW", "idc", "ide", "gW"):
        assert torch.equal(a[k], b[k]), f"deferred tail changed {k}: max diff {(a[k] - b[k]).abs().max().item():.3e}"


def test_deferred_tail_joins_before_other_users_of_the_context(cb):
    """Whoever else touches the context while a step's tail is still in flight on the library's third stream must wait for it first
    (Ctx::join_tail in Ctx::launch): here a UNet forward through the trainer's shared handle right after every step — it rewinds the
    step arena and writes over the very activations the CLIP gradient chain of that tail still reads (plan_and_run: the join has to
    survive its dry sizing pass, which a first version swallowed). Five such steps must end BITWISE where five steps with a full
    device synchronisation before each forward end."""
    from oracle import cb_oracle as O
    lib, ctx = cb
    m = MINI
    shapes = O.trainer_param_shapes(m["unet"], m["vae"], m["clip"], m["fr"], es=m["es"], inner=m["inner"], max_ids=m["max_ids"])
    W = cached_weights(O, shapes, 6)
    lat = m["image"] // (2 ** (len(m["vae"]["ch_mult"]) - 1))
    ctx.set_debug_taps(False)
    g = torch.Generator().manual_seed(11)
    xs = torch.randn(1, m["unet"]["in_channels"], lat, lat, generator=g); cs = torch.randn(1, 77, m["unet"]["context_dim"], generator=g)
    ts = torch.tensor([17], dtype=torch.int64)
    ends = []
    for synced in (False, True):
        tr = lib.Trainer(ctx, lib.WeightTable(W), trainer_cfg(lib, m))
        u = tr.unet()
        outs = []
        for s in range(5):
            dev = {k: dv(v) for k, v in make_batch(1, m["image"], lat, m["clip"]["vocab"], m["placeholder"], 700 + s, m["max_ids"]).items()}
            tr.step(dev["image"], dev["faces"], dev["ids"], dev["tokens"], dev["t"], dev["noise"], dev["post_noise"], 5e-3)
            del dev
            if synced:
                torch.cuda.synchronize()
            outs.append(u.forward(dv(xs), dv(ts), dv(cs)).clone())
        ends.append(dict(W=tr.get_state("mlp.weight").clone(), gW=tr.get_state("mlp.grad_weight").clone(), idc=tr.get_state("id_coefficients").clone(),
                         outs=[o.cpu() for o in outs]))
        u.close(); tr.close()
    a, b = ends
    for k in ("W", "gW", "idc"):
        assert torch.equal(a[k], b[k]), f"{k}: a user of the context overtook the deferred tail (max diff {(a[k] - b[k]).abs().max().item():.3e})"
    assert all(torch.equal(x, y) for x, y in zip(a["outs"], b["outs"]))


def test_deferred_tail_is_bitwise_the_joined_step_sd14(cb):
    """the same at SD v1-4 size, bs=1: 6 back-to-back steps of the ~1 400-launch two-stream step"""
    from oracle import cb_oracle as O
    lib, ctx = cb
    a = _tail_run(lib, ctx, SD14(O), 8, [(1, 5e-3)] * 6, True)
    b = _tail_run(lib, ctx, SD14(O), 8, [(1, 5e-3)] * 6, False)
    assert a["losses"] == b["losses"], (a["losses"], b["losses"])
    for k in ("W", "b", "mW", "vW", "idc", "ide", "gW"):
        assert torch.equal(a[k], b[k]), f"deferred tail changed {k}: max diff {(a[k] - b[k]).abs().max().item():.3e}"


SD14 = lambda O: dict(unet=O.UNET_SD14, vae=O.VAE_SD14, clip=O.CLIP_L14, fr=O.FR_R100, es=2, inner=512, max_ids=10, image=512,
                      placeholder=O.PLACEHOLDER_SKS)


def test_train_step_sd14_bs2_vs_oracle(cb):
    """BASELINE config 5's training half at full size: SD v1-4, 512x512, bs=2 (two identities in the batch, lr = bs * base_lr)
    — one full step against the oracle"""
    from oracle import cb_oracle as O
    lib, ctx = cb
    torch.set_num_threads(min(64, os.cpu_count() or 1))
    run_steps(lib, ctx, SD14(O), B=2, steps=1, lr=1e-2, seed=6, check_aux=True)


def test_train_2_steps_sd14_live_oracle(cb, golden_dir):
    """BASELINE config 2 AT FULL SIZE against the LIVE oracle on this box (the product's default warp table on both sides): 2 optimiser
    steps, SD v1-4 512x512 bs=1, on the replay stream. Every step checks loss / dW / db; the end state meets the north_star bar
    (coefficients within 1e-3 L2), and the live oracle reproduces the losses of the committed trajectories, which ties those fixtures
    to it. (10 steps through round 5, 4 in the first suites of round 6; the horizon on the default table is
    test_train_120_steps_sd14_default_table — 120 steps against a stored oracle run — and the live oracle's 8 s per step went with it:
    the second step is what shows the AdamW state carried over.)"""
    from oracle import cb_oracle as O
    lib, ctx = cb
    torch.set_num_threads(min(64, os.cpu_count() or 1))
    drift = []
    run_steps(lib, ctx, SD14(O), B=1, steps=2, lr=5e-3, seed=8, check_aux=False, drift=drift)
    REPORT["step[512px,B1].drift"] = drift
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/sd14_drift.json", "w") as f:
        json.dump(drift, f, indent=1)
    assert max(d["coef_l2"] for d in drift) <= 1e-3
    assert max(d["loss_rel"] for d in drift) <= 1e-4
    # the committed 100-step oracle trajectory (tests/golden/horizon_sd14_b1.pt, generated on other host cores) starts with these
    # very steps: the oracle recomputed live here must reproduce its losses
    fx = load(golden_dir, "horizon_sd14_b1.pt")
    for d, ref in zip(drift, fx["loss"][:4].tolist()):
        assert abs(d["loss_ref"] - ref) <= 2e-5 * abs(ref), (d["step"], d["loss_ref"], ref)
    # AdamW's update has size lr whatever the gradient's size, so the weight delta's relative error stays bounded instead of compounding
    assert drift[-1]["weight_delta_rel_l2"] <= 2e-2


def run_horizon(lib, ctx, fx, steps, tag, report, bars, save_traj=None, grid="fixture_host"):
    """HIP trainer over the replay stream of a committed oracle trajectory (oracle/gen_horizon.py -> tests/golden/horizon_*.pt):
    per-step table of |loss - ref| / ref, max-over-faces L2 of the step's coefficients (meta_net.py:289: the quantity north_star's
    bar is stated on), L2 of the EMA dictionary rows (embedding_manager.py:484-489), and the MLP gradient on the sampled rows.
    grid = "fixture_host": the warp table of the host that generated the trajectory is handed in through the ABI; "default": the
    product's own table (lib.Trainer builds it with this host's F.affine_grid) — for trajectories generated ON this host's table."""
    try:
        return _run_horizon(lib, ctx, fx, steps, tag, report, bars, save_traj, grid)
    finally:                                               # whatever happened: no debug taps, this host's own table (round-5 advisor finding)
        ctx.set_debug_taps(False)
        ctx._warp_S = None; ctx.ensure_warp_grid(512)


def _run_horizon(lib, ctx, fx, steps, tag, report, bars, save_traj, grid):
    from oracle import cb_oracle as O
    from oracle.replay import make_batch
    c = fx["cfg"]
    m = SD14(O)
    shapes = O.trainer_param_shapes(m["unet"], m["vae"], m["clip"], m["fr"], es=m["es"], inner=m["inner"], max_ids=m["max_ids"])
    W = cached_weights(O, shapes, c["weight_seed"])
    if c.get("next_weight_seed") is not None:              # the frozen weights of the draw that runs after this one, generated under this run
        weights_ahead(O, shapes, c["next_weight_seed"])
    ctx._warp_S = None                                     # the Trainer below installs this host's own table ...
    tr = lib.Trainer(ctx, lib.WeightTable(W), trainer_cfg(lib, m))
    if grid == "fixture_host":
        # ... replaced by the warp's constant sampling grid as the host that generated the trajectory evaluated it (this box's torch
        # rounds the K = 3 bmm of F.affine_grid differently in 14 % of the elements: oracle/replay.py::fixture_host_warp_grid)
        from oracle.replay import fixture_host_warp_grid
        ctx.set_warp_grid(fixture_host_warp_grid(512))
    B, ids, rs = c["B"], c["ids"], c["row_stride"]
    cst = int(c.get("id_coef_stride", 1))          # slim multi-seed fixtures keep the learned rows of every cst-th step
    steps = min(steps, int(fx["steps"]))
    table = []
    traj = dict(coef=[], id_coef=[], gW_rows=[], loss=[])
    ncoef = int(fx["coef"].shape[0])                 # slim long-horizon fixtures keep the step coefficients of their first steps only (or none)
    # Host reads are made only where the fixture has something to compare: the per-step coefficients are a debug tap (an eager step with a
    # sync), the learned rows a get_state (sync); the loss goes to a device scalar in stream order and is read once at the end. On the
    # slim three-draw fixtures (learned rows every 4th step, no step coefficients) three steps in four are enqueued back to back, as in
    # production — deferred tail and all (round 6: 39 -> 30 s per 800-step run on the GPU box).
    ctx.set_debug_taps(ncoef > 0)
    # the replay stream is generated on host cores (the photo-like faces: two float64 512^3 products per channel and step): a few worker
    # threads build the batches ahead of the GPU, same bytes in the same order
    from concurrent.futures import ThreadPoolExecutor
    gen = lambda s_: make_batch(B, 512, 64, m["clip"]["vocab"], m["placeholder"], c["batch_seed0"] + s_, m["max_ids"], photo=bool(c.get("photo")))
    first = gen(0)                                   # fills the low-pass operator cache before the threads start
    pool = ThreadPoolExecutor(max_workers=8)
    ahead = {0: None}
    for s_ in range(1, min(steps, 24)):
        ahead[s_] = pool.submit(gen, s_)
    losses_dev = []
    for s in range(steps):
        fut = ahead.pop(s)
        batch = first if s == 0 else fut.result()
        if s + 24 < steps:
            ahead[s + 24] = pool.submit(gen, s + 24)
        if s == ncoef and ncoef > 0:
            ctx.set_debug_taps(False)
        dev = {k: dv(v) for k, v in batch.items()}
        tr.step(dev["image"], dev["faces"], dev["ids"], dev["tokens"], dev["t"], dev["noise"], dev["post_noise"], c["lr"])
        losses_dev.append(tr.loss_tensor(torch.device("cuda:0")))
        row = dict(step=s + 1)
        idc = coef = None
        if (s + 1) % cst == 0:
            idc = tr.get_state("id_coefficients").reshape(m["max_ids"], -1)[ids]
            row["id_coef_l2"] = (idc - fx["id_coef"][(s + 1) // cst - 1]).norm(dim=1).max().item()
        if s < ncoef:
            coef = ctx.tap("coef").reshape(2 * B, -1)
            row["coef_l2"] = (coef - fx["coef"][s]).norm(dim=1).max().item()
        if s + 1 in fx.get("gW_rows", {}):
            g = tr.get_state("mlp.grad_weight").reshape(-1, 512)[::rs]
            ref = fx["gW_rows"][s + 1]
            row["grad_weight_rel_l2_sampled_rows"] = ((g - ref).norm() / ref.norm()).item()
            row["grad_weight_cosine_sampled_rows"] = (g.reshape(-1) @ ref.reshape(-1) / (g.norm() * ref.norm())).item()
        table.append(row)
        if save_traj and coef is not None and idc is not None:
            traj["coef"].append(coef.clone()); traj["id_coef"].append(idc.clone())
    losses = torch.stack(losses_dev).cpu().tolist()
    for row, loss in zip(table, losses):
        row["loss_rel"] = abs(loss - float(fx["loss"][row["step"] - 1])) / abs(float(fx["loss"][row["step"] - 1]))
    traj["loss"] = losses
    if save_traj:
        torch.save({k: (torch.stack(v) if k != "loss" else torch.tensor(v, dtype=torch.float64)) for k, v in traj.items() if len(v)}, save_traj)
    pool.shutdown(wait=True)
    if steps == int(fx["steps"]) and "W0_rows" in fx:
        w = tr.get_state("mlp.weight").reshape(-1, 512)[::rs]
        d_hip, d_ref = w - fx["W0_rows"], fx["Wm_rows"] - fx["W0_rows"]
        report[tag + ".weight_delta_rel_l2_sampled_rows"] = ((d_hip - d_ref).norm() / d_ref.norm()).item()
    assert int(tr.get_state("step").item()) == steps
    tr.close()
    report[tag + ".table"] = table
    n = len(table)
    ctab = [r for r in table if "coef_l2" in r]
    itab = [r for r in table if "id_coef_l2" in r]
    summary = dict(steps=n, id_coef_l2_last=itab[-1]["id_coef_l2"], id_coef_l2_max=max(r["id_coef_l2"] for r in itab),
                   loss_rel_max=max(r["loss_rel"] for r in table))
    if ctab:
        summary.update(coef_l2_max=max(r["coef_l2"] for r in ctab), coef_l2_last=ctab[-1]["coef_l2"], coef_l2_steps=len(ctab))
    report[tag + ".summary"] = summary
    os.makedirs("gpurun_out", exist_ok=True)
    with open(f"gpurun_out/{tag}.json", "w") as f:
        json.dump(dict(summary=summary, fixture_cfg={k: (list(v) if isinstance(v, (list, tuple)) else v) for k, v in c.items()}, table=table), f, indent=1)
    for k, bar in bars.items():
        assert summary[k] <= bar, f"{tag}: {k} = {summary[k]:.3e} exceeds {bar:.1e}"
    return summary


def test_train_120_steps_sd14_default_table(cb, golden_dir):
    """The full-horizon tests above hand the library the FIXTURE host's warp table through the ABI. This one runs the PRODUCT'S DEFAULT:
    lib.Trainer builds the table with this host's own F.affine_grid (lib.Context.ensure_warp_grid) and nothing is handed in. The oracle
    trajectory tests/golden/horizon_sd14_b1_boxgrid.pt (oracle/gen_horizon.py fp32boxgrid: config 2's default stream, 120 steps) was
    generated on exactly that table — the GPU box's EPYC evaluation, captured there (tests/golden/warp_grid_gpubox_512.npz,
    tools/probes/capture_host_affine_grid.py; 14 % of the elements one ulp off the fixture host's). A host whose F.affine_grid gives
    yet another table cannot run the comparison and says so. Bars: north_star's 1e-3 on the learned coefficients at every step and the
    implementation's own invariant — per-step coefficients within 2e-5 — over all 120 steps."""
    import hashlib
    import numpy as np
    lib, ctx = cb
    path = os.path.join(golden_dir, "horizon_sd14_b1_boxgrid.pt")
    if not os.path.exists(path):
        pytest.skip("tests/golden/horizon_sd14_b1_boxgrid.pt not generated (python oracle/gen_horizon.py fp32boxgrid)")
    z = np.load(os.path.join(golden_dir, "warp_grid_gpubox_512.npz"))
    M = torch.tensor([lib.WARP_TRANS_MATRIX], dtype=torch.float32)
    mine = torch.nn.functional.affine_grid(M, size=[1, 3, 512, 512], align_corners=True)[0].contiguous().numpy()
    if hashlib.sha1(mine.tobytes()).hexdigest()[:12] != str(z["sha1_host"]):
        pytest.skip(f"this host's F.affine_grid table ({hashlib.sha1(mine.tobytes()).hexdigest()[:12]}) is not the one the trajectory was generated on "
                    f"({z['sha1_host']}, {z['cpu']})")
    fx = load(golden_dir, "horizon_sd14_b1_boxgrid.pt")
    assert fx["cfg"].get("grid_npz") == "warp_grid_gpubox_512.npz" and int(fx["steps"]) >= 100
    s = run_horizon(lib, ctx, fx, int(fx["steps"]), "horizon_sd14_b1_default_table", REPORT,
                    dict(id_coef_l2_max=1e-3, coef_l2_max=2e-5, loss_rel_max=2e-4), grid="default")
    assert s["coef_l2_steps"] == s["steps"] >= 100


def test_train_800_steps_sd14_horizon(cb, golden_dir):
    """BASELINE config 2 at FULL SIZE over its own horizon: 800 optimiser steps (aigc_id.yaml:135) of SD v1-4 512x512 bs=1 on the HIP
    path against the oracle trajectory committed in tests/golden/horizon_sd14_b1_800.pt (oracle/gen_horizon.py fp32long, 3.6 h of host
    time; its first 100 steps are horizon_sd14_b1.pt, which also holds the sampled gradient rows, and its first four are the ones
    test_train_2_steps_sd14_live_oracle recomputes live) — the DEFAULT replay stream, whose face crops are white noise.
    Table: gpurun_out/horizon_sd14_b1_800.json (committed: profiles/r05_sd14_horizon800.json).

    north_star's bar — learned coefficients (EMA dictionary, embedding_manager.py:484-489) within 1e-3 L2 of the CPU path at fixed seed —
    is asserted at EVERY ONE of the 800 steps, with every loss to 2e-4 relative.

    History. Through round 4 this stream left the bar at step 76 and ended 4.8e-2 apart: the HIP warp re-derived the sampling
    grid (meta_net.py:253-257) in its own evaluation order, two fp32 ulps of a 512-pixel coordinate = 3e-5 on a white-noise face, and —
    found in round 5 — the 512 -> 112 resize computed its source index as an UNROUNDED product (the compiler contracted `scale * y - y0`
    into one fma). That 4e-5 step-coefficient difference met a LeakyReLU(0.2) kink of the head (meta_net.py:41-48) at step 16 and the
    trajectories separated. Now the warp consumes the reference's grid as a constant table (cb_ctx_set_warp_grid) and both kernels follow
    torch's operation order bit for bit (tests/test_gpu_ops.py: bit-exact at 512 x 512); what is left is the summation order inside the
    FR backbone / UNet (step coefficients ~5e-6), and sign disagreements of a coefficient happen only within that distance of zero."""
    lib, ctx = cb
    fx = dict(load(golden_dir, "horizon_sd14_b1_800.pt"))
    fx100 = load(golden_dir, "horizon_sd14_b1.pt")
    assert int(fx["steps"]) == 800 and torch.equal(fx["coef"][:100], fx100["coef"][:100]), "the 800-step trajectory must start with the 100-step fixture"
    fx["cfg"] = dict(fx100["cfg"], **fx["cfg"]); fx["gW_rows"] = fx100["gW_rows"]
    os.makedirs("gpurun_out", exist_ok=True)
    traj = "gpurun_out/horizon_sd14_b1_800_traj.pt"
    run_horizon(lib, ctx, fx, 800, "horizon_sd14_b1_800", REPORT, dict(loss_rel_max=2e-4), save_traj=traj)
    tab = REPORT["horizon_sd14_b1_800.table"]
    hip = torch.load(traj, weights_only=False)["coef"]
    ref = fx["coef"]                                   # the fixture keeps the per-step coefficients of the first 200 steps
    seen, events, first = set(), [], None
    for s_ in range(min(hip.shape[0], ref.shape[0])):
        for face, el in (torch.sign(hip[s_]) != torch.sign(ref[s_])).nonzero().tolist():
            if (face, el) not in seen:
                seen.add((face, el))
                events.append(dict(step=s_ + 1, face=face, element=el, oracle=float(ref[s_][face, el]), hip=float(hip[s_][face, el])))
                first = first or (s_ + 1)
    at = {str(k): dict(learned_coefficient_l2=tab[k - 1]["id_coef_l2"], step_coefficient_l2=tab[k - 1].get("coef_l2"), loss_rel=tab[k - 1]["loss_rel"])
          for k in (1, 10, 50, 75, 100, 200, 400, 800)}
    REPORT["horizon_sd14_b1_800.at_steps"] = at
    REPORT["horizon_sd14_b1_800.sign_disagreements_in_200_steps"] = events[:20]
    with open("gpurun_out/horizon_sd14_b1_800.json") as f:
        rec = json.load(f)
    rec.update(at_steps=at, first_sign_disagreement_step=first, sign_disagreements_in_200_steps=events[:20], coefficients_with_a_sign_disagreement=len(seen))
    rec["table"] = [r for r in rec["table"] if r["step"] <= 20 or r["step"] % 10 == 0]
    with open("gpurun_out/horizon_sd14_b1_800.json", "w") as f:
        json.dump(rec, f, indent=1)
    for e in events:                     # a sign can only differ where the value is round-off sized in BOTH paths
        assert abs(e["oracle"]) <= 2e-5 and abs(e["hip"]) <= 2e-5, f"sign disagreement away from zero: {e}"
    assert max(r["coef_l2"] for r in tab if "coef_l2" in r) <= 2e-5, "step coefficients (meta_net.py:289) beyond round-off in the first 200 steps (measured 4.7e-6)"
    worst = max(r["id_coef_l2"] for r in tab)
    assert worst <= 1e-3, f"learned coefficients left north_star's 1e-3 bar inside the 800-step horizon on the default (white-noise) stream: {worst:.3e}"
    assert worst <= 5e-5, f"regression guard: measured 2.0e-6 in round 5 (step coefficients 4.7e-6), now {worst:.3e}"


def test_two_gemm_tilings_stay_together_sd14(cb):
    """the control behind the horizon tests' reading: the HIP path's OWN noise is far below what separates it from the oracle. Two HIP
    runs of 40 SD-size steps on the white-noise stream that differ only in their GEMM tilings (the measured plan table against every
    problem pinned to 64x64 tiles with the planner's split: other fp32 summation orders in every conv / Linear of the step) must keep
    their learned coefficients within 1e-5 L2 of each other (measured 1.2e-6 in round 4): summation order alone does not separate two fp32
    evaluations at this horizon (what did, through round 4, was the warp's coordinate arithmetic: test_train_800_steps_sd14_horizon)."""
    from oracle import cb_oracle as O
    lib, ctx = cb
    m = SD14(O)
    shapes = O.trainer_param_shapes(m["unet"], m["vae"], m["clip"], m["fr"], es=m["es"], inner=m["inner"], max_ids=m["max_ids"])
    W = cached_weights(O, shapes, 8)
    out = []
    try:
        for plan in ((0, 0, 0), (64, 64, 0)):
            lib.check(lib.load().cb_debug_force_plan(*plan))
            tr = lib.Trainer(ctx, lib.WeightTable(W), trainer_cfg(lib, m))
            for s in range(40):
                dev = {k: dv(v) for k, v in make_batch(1, 512, 64, m["clip"]["vocab"], m["placeholder"], 100 + s, m["max_ids"]).items()}
                tr.step(dev["image"], dev["faces"], dev["ids"], dev["tokens"], dev["t"], dev["noise"], dev["post_noise"], 5e-3)
            out.append((tr.get_state("id_coefficients").reshape(m["max_ids"], -1).clone(), float(tr.get_state("loss")[0])))
            tr.close()
    finally:
        lib.check(lib.load().cb_debug_force_plan(0, 0, 0))
    d = (out[0][0] - out[1][0]).norm(dim=1).max().item()
    REPORT["sd14.two_gemm_tilings.learned_coefficient_l2_after_40_steps"] = d
    assert d <= 1e-5 and abs(out[0][1] - out[1][1]) <= 1e-5 * abs(out[0][1]), (d, out[0][1], out[1][1])


def test_train_800_steps_sd14_photo_faces_horizon(cb, golden_dir):
    """BASELINE config 2's full horizon at SD size on the replay stream with LOW-PASS ("photo-like") face crops (oracle/replay.py::low_pass;
    oracle trajectory tests/golden/horizon_sd14_b1_photo.pt, 800 steps, 2.3 h of host time). The default stream's faces are white noise, the
    worst case for the one place where the two paths' fp32 arithmetic differs by more than summation order: the warp's sampling coordinate
    (an ulp of 6e-5 pixel at 512 px against O(1) differences between neighbouring pixels, tools/probes/probe_warp_rounding.py). On smooth
    faces the step coefficients agree to ~6e-6 instead of 4e-5, far fewer pre-activations land within the difference of the head's
    LeakyReLU kink, and north_star's bar — learned coefficients within 1e-3 L2 of the CPU path at fixed seed — is asserted over the WHOLE
    800-step horizon, with every loss to 2e-4. Table: gpurun_out/horizon_sd14_b1_photo.json (committed: profiles/r04_sd14_horizon800_photo.json)."""
    lib, ctx = cb
    fx = dict(load(golden_dir, "horizon_sd14_b1_photo.pt"))
    assert fx["cfg"]["photo"] and int(fx["steps"]) == 800
    fx["cfg"] = dict(fx["cfg"], next_weight_seed=9)        # the first of test_train_800_steps_sd14_more_draws' frozen-weight draws
    os.makedirs("gpurun_out", exist_ok=True)
    traj = "gpurun_out/horizon_sd14_b1_photo_traj.pt"
    run_horizon(lib, ctx, fx, 800, "horizon_sd14_b1_photo", REPORT, dict(loss_rel_max=2e-4), save_traj=traj)
    tab = REPORT["horizon_sd14_b1_photo.table"]
    hip, ref = torch.load(traj, weights_only=False)["coef"], fx["coef"]          # the fixture keeps the step coefficients of the first 200 steps
    n = min(hip.shape[0], ref.shape[0])
    first = next((s_ + 1 for s_ in range(n) if bool((torch.sign(hip[s_]) != torch.sign(ref[s_])).any())), None)
    at = {str(k): dict(learned_coefficient_l2=tab[k - 1]["id_coef_l2"], step_coefficient_l2=tab[k - 1].get("coef_l2")) for k in (1, 10, 100, 200, 400, 800)}
    with open("gpurun_out/horizon_sd14_b1_photo.json") as f:
        rec = json.load(f)
    rec.update(at_steps=at, first_sign_disagreement_in_200_steps=first)
    rec["table"] = [r for r in rec["table"] if r["step"] <= 20 or r["step"] % 10 == 0]
    with open("gpurun_out/horizon_sd14_b1_photo.json", "w") as f:
        json.dump(rec, f, indent=1)
    REPORT["horizon_sd14_b1_photo.at_steps"] = at; REPORT["horizon_sd14_b1_photo.first_sign_disagreement_in_200_steps"] = first
    smooth = [r for r in tab[:(first - 1) if first else n] if "coef_l2" in r]
    assert len(smooth) >= 5 and max(r["coef_l2"] for r in smooth) <= 1.5e-5, (first, max(r["coef_l2"] for r in smooth) if smooth else None)
    worst = max(r["id_coef_l2"] for r in tab)
    assert worst <= 1e-3, f"learned coefficients left north_star's 1e-3 bar inside the 800-step horizon on photo-like faces: {worst:.3e}"


@pytest.mark.parametrize("name", ["horizon_sd14_b1_photo2", "horizon_sd14_b1_photo3", "horizon_sd14_b1_noise2"])
def test_train_800_steps_sd14_more_draws(cb, golden_dir, name):
    """more draws of the 800-step claim (round-4 verdict: one stream and one seed is one sample of a random event): other frozen
    weights (init seeds 9 / 10 / 11) AND other batch streams (seeds 3000+ / 5000+ / 7000+) — two with photo-like faces, one of the default
    white-noise kind — oracle trajectories from oracle/gen_horizon.py fp32photo2 / fp32photo3 / fp32noise2 (2.5 – 4 h of host time each;
    slim fixtures: loss per step, learned rows every 4th step; a fixture still being generated is compared over the steps it holds).
    north_star's 1e-3 on the learned coefficients is asserted over the whole horizon, every loss to 2e-4."""
    lib, ctx = cb
    path = os.path.join(golden_dir, name + ".pt")
    if not os.path.exists(path):
        pytest.skip(f"{name}.pt not generated yet (oracle/gen_horizon.py)")
    fx = dict(load(golden_dir, name + ".pt"))
    n = int(fx["steps"])
    if n < 100:
        pytest.skip(f"{name}.pt holds {n} steps so far (the generator writes it every 10 steps; at least 100 are needed for a horizon claim)")
    fx["cfg"] = dict(fx["cfg"], next_weight_seed={"horizon_sd14_b1_photo2": 10, "horizon_sd14_b1_photo3": 11}.get(name))
    # the slim files hold no per-step coefficients; <name>_head.pt (oracle/gen_horizon.py fp32*head, round 6) is the SAME oracle trajectory's
    # first 120 steps with them — its losses and learned rows must be the slim file's, bit for bit — so that the implementation's own
    # invariant, per-step coefficient agreement (asserted at 2e-5), is held on this draw too and not only on the default / photo streams
    head_steps = 0
    if os.path.exists(os.path.join(golden_dir, name + "_head.pt")):
        hd = load(golden_dir, name + "_head.pt")
        k = min(int(hd["steps"]), n)
        stride = int(fx["cfg"].get("id_coef_stride", 1))
        assert torch.equal(hd["loss"][:k], fx["loss"][:k]) and torch.equal(hd["id_coef"][:k // stride], fx["id_coef"][:k // stride]), \
            f"{name}_head.pt is not the first {k} steps of {name}.pt"
        fx["coef"] = hd["coef"][:k]
        head_steps = k
    s = run_horizon(lib, ctx, fx, n, name, REPORT, dict(loss_rel_max=2e-4))
    if head_steps:
        REPORT[name + ".per_step_coefficient_l2_max"] = dict(steps=head_steps, value=s["coef_l2_max"])
        assert s["coef_l2_steps"] == head_steps and s["coef_l2_max"] <= 2e-5, (name, s)
    tab = [r for r in REPORT[name + ".table"] if "id_coef_l2" in r]
    with open(f"gpurun_out/{name}.json") as f:
        rec = json.load(f)
    rec["table"] = [r for r in rec["table"] if r["step"] % 20 == 0]
    rec["at_steps"] = {str(r["step"]): r["id_coef_l2"] for r in tab if r["step"] in (4, 100, 200, 400, 800)}
    with open(f"gpurun_out/{name}.json", "w") as f:
        json.dump(rec, f, indent=1)
    REPORT[name + ".steps_compared"] = n
    # a draw whose trajectories meet a LeakyReLU kink inside the horizon shows it as a jump of 1 - 2 decades between two table rows:
    # recorded (photo2: round-off agreement 2.4e-6 through step 680, 3.5e-5 at 700, 4.8e-4 at 760, 4.3e-4 at 800), and still held to the bar
    jumps = [r["step"] for r in tab if r["id_coef_l2"] > 1e-5]              # round-off agreement is 2 - 3e-6 on every draw
    rec["first_step_beyond_1e-5"] = jumps[0] if jumps else None
    with open(f"gpurun_out/{name}.json", "w") as f:
        json.dump(rec, f, indent=1)
    REPORT[name + ".first_step_beyond_1e-5"] = rec["first_step_beyond_1e-5"]
    assert s["id_coef_l2_max"] <= 1e-3, f"{name}: learned coefficients left north_star's 1e-3 bar inside {n} steps: {s['id_coef_l2_max']:.3e}"


def test_train_800_steps_mini_vs_reference_run(cb, golden_dir):
    """BASELINE config 2's full horizon — 800 optimiser steps, aigc_id.yaml:135 — against a run of the REFERENCE ITSELF:
    tests/golden/horizon_mini_ref800.pt holds the trajectory of the reference's composed step (imported LatentDiffusion.forward /
    p_losses / q_sample / get_loss, UNetModel, Encoder, EmbeddingManagerId + MetaIdNet, torch.optim.AdamW; oracle/gen_horizon_ref.py)
    at mini size, the same reference code with seeded noise on the FR features, and the oracle's trajectory on the same stream
    (held to the reference on the CPU side: tests/test_oracle_golden.py::test_oracle_tracks_the_reference_for_800_steps).
    The HIP trainer runs the 800 steps here. north_star's bar — learned coefficients within 1e-3 L2 — is asserted at EVERY step
    against the reference (not the oracle), every loss to 1e-4, and the distance is put next to the perturbed reference runs:
    the HIP path must sit below the reference perturbed by 2.5e-5 (what its feature difference measures at SD size on white-noise
    faces). Table: gpurun_out/horizon_mini_ref800.json (committed copy under profiles/)."""
    from oracle.replay import ref_stream_batch
    from test_oracle_golden import learned_l2, ref800_state
    lib, ctx = cb
    fx = load(golden_dir, "horizon_mini_ref800.pt")
    c, ref = fx["cfg"], fx["ref"]
    m = dict(unet=c["unet"], vae=c["vae"], clip=c["clip"], fr=c["fr"], es=c["es"], inner=c["inner"], max_ids=3, image=64, placeholder=48136)
    tr = lib.Trainer(ctx, lib.WeightTable(ref800_state(fx)), trainer_cfg(lib, m))
    lat = 64 // (2 ** (len(c["vae"]["ch_mult"]) - 1))
    K = int(fx["steps"])
    coef, loss = [], []
    for s in range(K):
        dev = {k: dv(v) for k, v in ref_stream_batch(s, lat).items()}
        tr.step(dev["image"], dev["faces"], dev["ids"], dev["tokens"], dev["t"], dev["noise"], dev["post_noise"], c["lr"])
        coef.append(tr.get_state("id_coefficients").reshape(3, -1).clone()); loss.append(float(tr.get_state("loss")[0]))
    assert int(tr.get_state("step").item()) == K
    w_hip = tr.get_state("mlp.weight").reshape(ref["W_final"].shape)
    tr.close()
    coef = torch.stack(coef); loss = torch.tensor(loss, dtype=torch.float64)
    d_ref = learned_l2(coef, ref["id_coef"])
    at = (1, 10, 100, 200, 400, 800)
    row = lambda d: {str(k): float(d[k - 1]) for k in at}
    table = dict(hip_vs_reference=row(d_ref), hip_vs_oracle=row(learned_l2(coef, fx["oracle"]["id_coef"])),
                 oracle_vs_reference=row(learned_l2(fx["oracle"]["id_coef"], ref["id_coef"])))
    for tag in ("ref_fr3e-6", "ref_fr2.5e-5", "ref_fr2e-4"):
        table[tag + "_vs_reference"] = row(learned_l2(fx[tag]["id_coef"], ref["id_coef"]))
    summary = dict(steps=K, hip_vs_reference_max=float(d_ref.max()), hip_vs_reference_at_800=float(d_ref[-1]),
                   loss_rel_max=float(((loss - ref["loss"]).abs() / ref["loss"].abs()).max()),
                   mlp_weight_final_rel_l2=float((w_hip - ref["W_final"]).norm() / (ref["W_final"] - fx["init"]["Wm"]).norm()),
                   perturbed_reference_2p5e5_max=float(learned_l2(fx["ref_fr2.5e-5"]["id_coef"], ref["id_coef"]).max()))
    REPORT["horizon_mini_ref800"] = dict(summary=summary, learned_coefficient_l2_at_steps=table)
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/horizon_mini_ref800.json", "w") as f:
        json.dump(REPORT["horizon_mini_ref800"], f, indent=1)
    assert summary["loss_rel_max"] <= 1e-4, summary
    assert summary["hip_vs_reference_max"] <= 1e-3, f"learned coefficients left north_star's 1e-3 bar against the reference run: {summary}"
    assert summary["hip_vs_reference_max"] <= max(2.0 * summary["perturbed_reference_2p5e5_max"], 2e-5), summary


# ------------------------------------------------------------------ sampling path: DDIM + VAE decoder
def test_vae_decoder_mini_vs_reference_golden(cb, golden_dir):
    from oracle import cb_oracle as O
    lib, ctx = cb
    fx = load(golden_dir, "vae_dec_mini.pt")
    W = O.init_weights(O.vae_decoder_param_shapes(fx["cfg"]), seed=fx["weight_seed"])
    d = lib.VAEDecoder(ctx, lib.WeightTable(W), lib.vae_cfg(**fx["cfg"]))
    img = d.decode(dv(fx["z"]), fx["scale_factor"])
    img01 = d.decode(dv(fx["z"]), fx["scale_factor"], to01=True)
    ctx.sync()
    close("vae_dec_mini.image", img, fx["image"], 5e-5)
    close("vae_dec_mini.image01", img01, fx["image01"], 5e-5)
    d.close()


def test_ddim_mini_vs_reference_golden(cb, golden_dir):
    """the reference's DDIMSampler over its own mini UNetModel: one guided update, then whole-loop runs (cfg / no cfg / eta>0)"""
    from oracle import cb_oracle as O
    lib, ctx = cb
    fx = load(golden_dir, "ddim_mini.pt")
    cfg = fx["cfg"]
    u = lib.UNet(ctx, lib.WeightTable(O.init_weights(O.unet_param_shapes(cfg), seed=fx["weight_seed"])), lib.unet_cfg(**cfg))
    x_T, cond, uncond = dv(fx["x_T"]), dv(fx["cond"]), dv(fx["uncond"])
    ac = O.make_schedule()["alphas_cumprod"]
    s5 = O.ddim_schedule(ac, 5, 0.0)
    tabs = lambda s: (s["timesteps"], s["alphas"], s["alphas_prev"], s["sigmas"], s["sqrt_one_minus_alphas"])
    # one p_sample_ddim update at index 2 with guidance 7.5: batch-2n UNet pass + cb_ddim_update
    t = torch.full((4,), int(s5["timesteps"][2]), dtype=torch.int64)
    eps = u.forward(torch.cat([x_T, x_T]).contiguous(), dv(t), torch.cat([uncond, cond]).contiguous())
    x1, p0 = lib.ddim_update(ctx, x_T.clone(), eps, float(s5["alphas"][2]), float(s5["alphas_prev"][2]), float(s5["sigmas"][2]),
                             float(s5["sqrt_one_minus_alphas"][2]), scale=7.5)
    ctx.sync()
    close("ddim_mini.step.x_prev", x1, fx["step_index2"]["x_prev"], 1e-4)
    close("ddim_mini.step.pred_x0", p0, fx["step_index2"]["pred_x0"], 1e-4)
    # whole loops on the device
    x, p0 = u.ddim_sample(x_T, cond, uncond, *tabs(s5), scale=7.5, want_pred_x0=True)
    ctx.sync()
    close("ddim_mini.run_cfg.samples", x, fx["run_5_eta0_cfg7.5"]["samples"], 5e-4)
    close("ddim_mini.run_cfg.pred_x0", p0, fx["run_5_eta0_cfg7.5"]["pred_x0"], 5e-4)
    x, _ = u.ddim_sample(x_T, cond, None, *tabs(s5), scale=1.0)
    ctx.sync()
    close("ddim_mini.run_nocfg.samples", x, fx["run_5_eta0_nocfg"]["samples"], 5e-4)
    r = fx["run_4_eta0.5_cfg3"]
    x, _ = u.ddim_sample(x_T, cond, uncond, *tabs(O.ddim_schedule(ac, 4, 0.5)), scale=3.0, noise=dv(r["noise"]))
    ctx.sync()
    close("ddim_mini.run_eta.samples", x, r["samples"], 5e-4)
    # eta != 0 without a noise tensor is an argument error, not a silent eta = 0
    with pytest.raises(lib.CBError):
        u.ddim_sample(x_T, cond, uncond, *tabs(O.ddim_schedule(ac, 4, 0.5)), scale=3.0, noise=None)
    assert torch.equal(x_T.cpu(), fx["x_T"]), "x_T must not be modified by the sampler"
    u.close()


def test_host_sampler_mini_vs_oracle(cb):
    """the reference-facing classes (LatentDiffusion + DDIMSampler + AutoencoderKL.decode, same call sequence as
    scripts/stable_txt2img.py:320-343) at mini sizes against the oracle: fused loop and the hook (step-by-step) loop"""
    import celebbasis_amd
    from celebbasis_amd.ldm.models.diffusion.ddim import DDIMSampler
    from celebbasis_amd.ldm.models.diffusion.ddpm import LatentDiffusion
    from oracle import cb_oracle as O
    lib, ctx = cb
    ucfg = dict(in_channels=4, out_channels=4, model_channels=32, num_res_blocks=1, channel_mult=(1, 2, 2), attention_resolutions=(1, 2),
                num_heads=2, context_dim=48)
    vcfg = dict(ch=32, ch_mult=(1, 2, 2), num_res_blocks=1, in_channels=3, z_channels=4)
    T = lambda t, p: dict(target=t, params=p)
    m = LatentDiffusion(
        first_stage_config=T("celebbasis_amd.ldm.models.autoencoder.AutoencoderKL",
                             dict(embed_dim=4, ddconfig=dict(double_z=True, z_channels=4, resolution=64, in_channels=3, out_ch=3, ch=32,
                                                             ch_mult=[1, 2, 2], num_res_blocks=1, attn_resolutions=[], dropout=0.0))),
        cond_stage_config=T("celebbasis_amd.ldm.modules.encoders.modules.FrozenCLIPEmbedder", dict(vocab=120, width=48, layers=1, heads=4, mlp=96)),
        unet_config=T("celebbasis_amd.ldm.modules.diffusionmodules.openaimodel.UNetModel",
                      dict(image_size=16, in_channels=4, out_channels=4, model_channels=32, attention_resolutions=[1, 2], num_res_blocks=1,
                           channel_mult=[1, 2, 2], num_heads=2, use_spatial_transformer=True, transformer_depth=1, context_dim=48)),
        linear_start=0.00085, linear_end=0.012, timesteps=1000, scale_factor=0.18215, image_size=16, channels=4)
    Wu = O.init_weights(O.unet_param_shapes(ucfg), seed=11)
    Wd = O.init_weights(O.vae_decoder_param_shapes(vcfg), seed=12)
    m.model.diffusion_model.load_state_dict(Wu)
    m.first_stage_model.load_state_dict(Wd, strict=False)
    g = torch.Generator().manual_seed(13)
    n = 2
    x_T = torch.randn(n, 4, 16, 16, generator=g); c = torch.randn(n, 77, 48, generator=g); uc = torch.randn(n, 77, 48, generator=g)
    sampler = DDIMSampler(m)
    samples, inter = sampler.sample(S=4, conditioning=dv(c), batch_size=n, shape=[4, 16, 16], verbose=False, unconditional_guidance_scale=5.0,
                                    unconditional_conditioning=dv(uc), eta=0.0, x_T=dv(x_T))
    img = torch.clamp((m.decode_first_stage(samples) + 1.0) / 2.0, min=0.0, max=1.0)
    seen = []
    samples2, _ = sampler.sample(S=4, conditioning=dv(c), batch_size=n, shape=[4, 16, 16], verbose=False, unconditional_guidance_scale=5.0,
                                 unconditional_conditioning=dv(uc), eta=0.0, x_T=dv(x_T), callback=seen.append)
    ctx.sync()
    with torch.no_grad():
        ref, ref_p0 = O.ddim_sample(Wu, ucfg, x_T, c, uc, O.ddim_schedule(O.make_schedule()["alphas_cumprod"], 4, 0.0), scale=5.0)
        ref_img = torch.clamp((O.decode_first_stage(Wd, vcfg, ref, 0.18215) + 1.0) / 2.0, 0.0, 1.0)
    close("host_sampler.samples", samples, ref, 5e-4)
    close("host_sampler.pred_x0", inter["pred_x0"][-1], ref_p0, 5e-4)
    close("host_sampler.samples_stepwise", samples2, ref, 5e-4)
    close("host_sampler.image", img, ref_img, 1e-3)
    assert seen == [0, 1, 2, 3]


@pytest.mark.parametrize("n", [1, 8])
def test_sampler_sd14_vs_oracle(cb, n):
    """SD v1-4 sizes: a guided DDIM step (a batch-2n UNet pass at 64x64) + decode to 512x512 against the
    CPU oracle; n = 8 is the reference's n_samples (stable_txt2img.py:108, config 5): a batch-16 UNet pass and the chunked 8-image decode"""
    from oracle import cb_oracle as O
    lib, ctx = cb
    cfg = O.UNET_SD14
    W = O.init_weights(O.unet_param_shapes(cfg), seed=3)
    u = lib.UNet(ctx, lib.WeightTable(W), lib.unet_cfg(**cfg))
    g = torch.Generator().manual_seed(6)
    x_T = torch.randn(n, 4, 64, 64, generator=g); c = torch.randn(n, 77, 768, generator=g); uc = torch.randn(n, 77, 768, generator=g)
    s = O.ddim_schedule(O.make_schedule()["alphas_cumprod"], 1, 0.0)     # one step keeps the CPU oracle's share of the suite short (the multi-step loop: ddim_mini)
    x, _ = u.ddim_sample(dv(x_T), dv(c), dv(uc), s["timesteps"], s["alphas"], s["alphas_prev"], s["sigmas"], s["sqrt_one_minus_alphas"], scale=10.0)
    ctx.sync()
    u.close()
    Wd = O.init_weights(O.vae_decoder_param_shapes(O.VAE_SD14), seed=8)
    d = lib.VAEDecoder(ctx, lib.WeightTable(Wd), lib.vae_cfg(**O.VAE_SD14))
    torch.set_num_threads(min(64, os.cpu_count() or 1))
    pick = sorted({0, n - 1})                    # the oracle samples / decodes a subset: the first and the last image of the batch (samples of a
    with torch.no_grad():                        # batch do not interact: a batch-2|pick| oracle pass instead of batch-2n, 30 s of host time at n = 8)
        ref_pick, _ = O.ddim_sample(W, cfg, x_T[pick], c[pick], uc[pick], s, scale=10.0)
        ref_img = O.decode_first_stage(Wd, O.VAE_SD14, ref_pick, 0.18215)
    ref = x.detach().cpu().clone()               # the latents handed to the decoders: the oracle's where it has them (so the decode check is
    ref[pick] = ref_pick                         # independent of the sampler check), the product's own for the rest of the chunked n-image decode
    img = d.decode(dv(ref), 0.18215)[pick]
    ctx.sync()
    close(f"sampler_sd14[n={n}].samples", x[pick], ref_pick, 1e-3)
    close(f"sampler_sd14[n={n}].image", img, ref_img, 5e-4)
    d.close()
    # the same sampler + decoder on a bf16 context (the reference samples under fp16 autocast, stable_txt2img.py:320-322; config 5's
    # bf16 numbers in DESIGN are quoted for this path) against the SAME fp32 oracle outputs. bf16 bars (8 significant bits, a
    # guidance scale of 10 multiplies the eps error): latents 8e-2 of scale, decoded image 4e-2 of scale
    ctxb = lib.Context(dtype="bf16")
    ub = lib.UNet(ctxb, lib.WeightTable(W), lib.unet_cfg(**cfg))
    xb, _ = ub.ddim_sample(dv(x_T), dv(c), dv(uc), s["timesteps"], s["alphas"], s["alphas_prev"], s["sigmas"], s["sqrt_one_minus_alphas"], scale=10.0)
    ctxb.sync()
    ub.close()
    db = lib.VAEDecoder(ctxb, lib.WeightTable(Wd), lib.vae_cfg(**O.VAE_SD14))
    imgb = db.decode(dv(ref), 0.18215)[pick]
    img01 = db.decode(dv(ref), 0.18215, to01=True)[pick]
    ctxb.sync()
    close(f"bf16.sampler_sd14[n={n}].samples", xb[pick], ref_pick, 8e-2)
    close(f"bf16.sampler_sd14[n={n}].image", imgb, ref_img, 4e-2)
    close(f"bf16.sampler_sd14[n={n}].image01", img01, ((ref_img + 1) / 2).clamp(0, 1), 4e-2)
    db.close(); ctxb.close()


def test_extract_pt_formats(cb, tmp_path):
    """scripts/extract_pt.py:96-139 interchange files: names, shapes, dtypes and values (einsum with the basis + mean)"""
    from celebbasis_amd.scripts.extract_pt import extract
    g = torch.Generator().manual_seed(17)
    es, h, K, W, max_ids = 2, 1, 512, 768, 10
    basis = torch.randn(es, 1 + K, W, generator=g) * 0.05
    coefs = [torch.randn(es, h, K, generator=g) for _ in range(max_ids)]
    for fp16 in (False, True):
        folder = tmp_path / ("ti16" if fp16 else "ti32")
        extract(coefs, basis, str(tmp_path / "celeb_basis.pt"), str(folder), save_fp16=fp16, device=DEV)
        assert torch.equal(torch.load(tmp_path / "celeb_basis.pt"), basis)
        tag = "_fp16" if fp16 else ""
        assert sorted(os.listdir(folder)) == sorted([f"id_embedding{tag}_{i:03d}.pt" for i in range(max_ids)] +
                                                    [f"id_coefficient{tag}_{i:03d}.pt" for i in range(max_ids)])
        for i in (0, 9):
            z = torch.load(folder / f"id_embedding{tag}_{i:03d}.pt"); c = torch.load(folder / f"id_coefficient{tag}_{i:03d}.pt")
            assert z.dtype == (torch.float16 if fp16 else torch.float32) and tuple(z.shape) == (es * h, W) and tuple(c.shape) == (es, h, K)
            ref = (torch.einsum("ehk,ekc->ehc", coefs[i], basis[:, 1:]) + basis[:, 0].unsqueeze(1)).reshape(es * h, W)
            close(f"extract_pt.z{tag}.{i}", z.float(), ref, 2e-3 if fp16 else 1e-5)
            assert torch.equal(c.float(), coefs[i].half().float() if fp16 else coefs[i])


def test_txt2img_script_mini(cb, tmp_path):
    """the stable_txt2img counterpart end to end at mini sizes: YAML with the reference's target strings -> .ckpt ->
    embeddings_gs-*.pt -> prompts -> DDIM with guidance -> decoded .jpg files (orchestration; numerics are covered above)"""
    import yaml
    from PIL import Image
    from celebbasis_amd.scripts import stable_txt2img
    from celebbasis_amd.ldm.modules.encoders.modules import FrozenCLIPEmbedder
    from oracle import cb_oracle as O
    ucfg = dict(in_channels=4, out_channels=4, model_channels=32, num_res_blocks=1, channel_mult=(1, 2, 2), attention_resolutions=(1, 2),
                num_heads=2, context_dim=48)
    vcfg = dict(ch=32, ch_mult=(1, 2, 2), num_res_blocks=1, in_channels=3, z_channels=4)
    ccfg = dict(vocab=49408, width=48, layers=1, heads=4, ctx_len=77, mlp=96)
    K = 16
    cfg = dict(model=dict(target="ldm.models.diffusion.ddpm.LatentDiffusion", params=dict(
        linear_start=0.00085, linear_end=0.012, timesteps=1000, scale_factor=0.18215, image_size=16, channels=4,
        first_stage_key="image", cond_stage_key="caption", conditioning_key="crossattn", cond_stage_trainable=False,
        personalization_config=dict(target="ldm.modules.embedding_manager.EmbeddingManagerId", params=dict(
            placeholder_strings=["sks", "ks"], initializer_words=["face"], max_ids=10, num_embeds_per_token=2, meta_mlp_depth=1,
            loss_type="none", meta_inner_dim=K, test_mode="coefficient", momentum=0.99, save_fp16=False)),
        unet_config=dict(target="ldm.modules.diffusionmodules.openaimodel.UNetModel", params=dict(
            image_size=16, in_channels=4, out_channels=4, model_channels=32, attention_resolutions=[1, 2], num_res_blocks=1,
            channel_mult=[1, 2, 2], num_heads=2, use_spatial_transformer=True, transformer_depth=1, context_dim=48)),
        first_stage_config=dict(target="ldm.models.autoencoder.AutoencoderKL", params=dict(embed_dim=4, ddconfig=dict(
            double_z=True, z_channels=4, resolution=64, in_channels=3, out_ch=3, ch=32, ch_mult=[1, 2, 2], num_res_blocks=1,
            attn_resolutions=[], dropout=0.0))),
        cond_stage_config=dict(target="ldm.modules.encoders.modules.FrozenCLIPEmbedder", params=dict(
            vocab=49408, width=48, layers=1, heads=4, mlp=96, num_embeds_per_token=2, n_components=K)))))
    (tmp_path / "mini.yaml").write_text(yaml.safe_dump(cfg))
    sd = {"model.diffusion_model." + k: v for k, v in O.init_weights(O.unet_param_shapes(ucfg), seed=21).items()}
    sd.update({"first_stage_model." + k: v for k, v in O.init_weights(O.vae_decoder_param_shapes(vcfg), seed=22).items()})
    clip_shapes = {"cond_stage_model." + FrozenCLIPEmbedder.PREFIX + k: v for k, v in FrozenCLIPEmbedder.param_shapes(ccfg).items()}
    sd.update(O.init_weights(clip_shapes, seed=23))
    g = torch.Generator().manual_seed(24)
    sd["celeb_embeddings"] = torch.randn(2, 1 + K, 48, generator=g) * 0.05
    torch.save({"state_dict": sd}, tmp_path / "mini.ckpt")
    torch.save({"id_coefficients": [torch.randn(2, 1, K, generator=g) for _ in range(10)]}, tmp_path / "embeddings_gs-800.pt")
    (tmp_path / "prompts.txt").write_text("a photo of sks person\nsks person and ks person on the beach\n")
    out = tmp_path / "out"
    stable_txt2img.main(["--config", str(tmp_path / "mini.yaml"), "--ckpt", str(tmp_path / "mini.ckpt"),
                         "--embedding_path", str(tmp_path / "embeddings_gs-800.pt"), "--from-file", str(tmp_path / "prompts.txt"),
                         "--n_samples", "2", "--ddim_steps", "4", "--H", "64", "--W", "64", "--f", "4", "--scale", "5", "--eval_id1", "3",
                         "--eval_id2", "6", "--outdir", str(out), "--fixed_code", "--allow_offline_tokens"])
    files = sorted(os.listdir(out / "samples"))
    assert files == ["00000.jpg", "00001.jpg", "00002.jpg", "00003.jpg"]
    assert Image.open(out / "samples" / "00000.jpg").size == (64, 64)
    grids = [f for f in os.listdir(out) if f.endswith(".jpg")]
    assert len(grids) == 2 and Image.open(out / grids[0]).size == (128, 64)


def test_txt2img_script_image_mode(cb, tmp_path):
    """`test_mode: image` end to end through the script (stable_txt2img.py:290-308): face FILES -> faces tensor -> FR-net (real
    iresnet100 shapes) + trained MLP head -> embeddings -> prompts with both placeholders -> DDIM -> .jpg"""
    import numpy as np
    import yaml
    from PIL import Image
    from celebbasis_amd.scripts import stable_txt2img
    from celebbasis_amd.ldm.modules.encoders.modules import FrozenCLIPEmbedder
    from oracle import cb_oracle as O
    ucfg = dict(in_channels=4, out_channels=4, model_channels=32, num_res_blocks=1, channel_mult=(1, 2, 2), attention_resolutions=(1, 2),
                num_heads=2, context_dim=48)
    vcfg = dict(ch=32, ch_mult=(1, 2, 2), num_res_blocks=1, in_channels=3, z_channels=4)
    ccfg = dict(vocab=49408, width=48, layers=1, heads=4, ctx_len=77, mlp=96)
    K = 16
    cfg = dict(model=dict(target="ldm.models.diffusion.ddpm.LatentDiffusion", params=dict(
        linear_start=0.00085, linear_end=0.012, timesteps=1000, scale_factor=0.18215, image_size=16, channels=4,
        first_stage_key="image", cond_stage_key="caption", conditioning_key="crossattn", cond_stage_trainable=False,
        personalization_config=dict(target="ldm.modules.embedding_manager.EmbeddingManagerId", params=dict(
            placeholder_strings=["sks", "ks"], max_ids=10, num_embeds_per_token=2, meta_mlp_depth=1,
            loss_type="none", meta_inner_dim=K, test_mode="image", momentum=0.99, save_fp16=False)),
        unet_config=dict(target="ldm.modules.diffusionmodules.openaimodel.UNetModel", params=dict(
            image_size=16, in_channels=4, out_channels=4, model_channels=32, attention_resolutions=[1, 2], num_res_blocks=1,
            channel_mult=[1, 2, 2], num_heads=2, use_spatial_transformer=True, transformer_depth=1, context_dim=48)),
        first_stage_config=dict(target="ldm.models.autoencoder.AutoencoderKL", params=dict(embed_dim=4, ddconfig=dict(
            double_z=True, z_channels=4, resolution=64, in_channels=3, out_ch=3, ch=32, ch_mult=[1, 2, 2], num_res_blocks=1,
            attn_resolutions=[], dropout=0.0))),
        cond_stage_config=dict(target="ldm.modules.encoders.modules.FrozenCLIPEmbedder", params=dict(
            vocab=49408, width=48, layers=1, heads=4, mlp=96, num_embeds_per_token=2, n_components=K)))))
    (tmp_path / "mini.yaml").write_text(yaml.safe_dump(cfg))
    sd = {"model.diffusion_model." + k: v for k, v in O.init_weights(O.unet_param_shapes(ucfg), seed=21).items()}
    sd.update({"first_stage_model." + k: v for k, v in O.init_weights(O.vae_decoder_param_shapes(vcfg), seed=22).items()})
    clip_shapes = {"cond_stage_model." + FrozenCLIPEmbedder.PREFIX + k: v for k, v in FrozenCLIPEmbedder.param_shapes(ccfg).items()}
    sd.update(O.init_weights(clip_shapes, seed=23))
    sd.update(O.init_weights(O.frnet_param_shapes(O.FR_R100, "embedding_manager.meta_id_net.id_model."), seed=25))
    g = torch.Generator().manual_seed(24)
    sd["celeb_embeddings"] = torch.randn(2, 1 + K, 48, generator=g) * 0.05
    torch.save({"state_dict": sd}, tmp_path / "mini.ckpt")
    torch.save({"meta_id_net": {"stylegan_mlp.net.0.weight": torch.randn(2 * K, 512, generator=g),
                                "stylegan_mlp.net.0.bias": torch.zeros(2 * K)}}, tmp_path / "embeddings_gs-800.pt")
    rs = np.random.RandomState(3)
    for n in ("a.png", "b.png"):
        Image.fromarray(rs.randint(0, 255, (96, 96, 3), dtype=np.uint8)).save(tmp_path / n)
    (tmp_path / "prompts.txt").write_text("sks person and ks person shaking hands\n")
    out = tmp_path / "out"
    stable_txt2img.main(["--config", str(tmp_path / "mini.yaml"), "--ckpt", str(tmp_path / "mini.ckpt"),
                         "--embedding_path", str(tmp_path / "embeddings_gs-800.pt"), "--from-file", str(tmp_path / "prompts.txt"),
                         "--n_samples", "2", "--ddim_steps", "4", "--H", "64", "--W", "64", "--f", "4", "--scale", "5",
                         "--face_img1", str(tmp_path / "a.png"), "--face_img2", str(tmp_path / "b.png"),
                         "--outdir", str(out), "--fixed_code", "--allow_offline_tokens"])
    assert sorted(os.listdir(out / "samples")) == ["00000.jpg", "00001.jpg"]
    with pytest.raises(SystemExit):        # image mode without a face image is an error, not a silent coefficient lookup
        stable_txt2img.main(["--config", str(tmp_path / "mini.yaml"), "--ckpt", str(tmp_path / "mini.ckpt"),
                             "--embedding_path", str(tmp_path / "embeddings_gs-800.pt"), "--outdir", str(out), "--allow_offline_tokens"])


def test_manager_image_mode_mini_vs_oracle(cb):
    """eval-mode EmbeddingManagerId with `test_mode: image` (embedding_manager.py:296-300): faces -> warp/resize -> FR-net ->
    MLP head -> basis -> row surgery, all native, against the oracle composition; two persons in one prompt"""
    from celebbasis_amd.ldm.modules.embedding_manager import EmbeddingManagerId
    from celebbasis_amd.ldm.modules.id_embedding.iresnet import IResNet
    from oracle import cb_oracle as O
    lib, ctx = cb
    fr = MINI["fr"]; es, K, Wd = MINI["es"], MINI["inner"], MINI["clip"]["width"]

    class Embedder:                    # the two things the manager asks of the CLIP wrapper
        cfg = dict(width=Wd)

        def tokenize(self, s):
            ids = torch.full((1, 77), 49407, dtype=torch.int64); ids[0, 0] = 49406
            ids[0, 1] = {"sks": 99, "ks": 98}[s]
            return ids

    em = EmbeddingManagerId(Embedder(), placeholder_strings=["sks", "ks"], max_ids=3, num_embeds_per_token=es, meta_mlp_depth=1,
                            loss_type="none", meta_inner_dim=K, test_mode="image", save_fp16=False).eval()
    Wfr = O.init_weights(O.frnet_param_shapes(fr), seed=31)
    em.meta_id_net.id_model = IResNet(fr["layers"], fr["feat"], fr["width"], fr["in_size"])
    em.meta_id_net.id_model.load_state_dict(Wfr)
    g = torch.Generator().manual_seed(32)
    em.meta_id_net.weight.data = torch.randn(es * K, 512, generator=g); em.meta_id_net.bias.data = torch.randn(es * K, generator=g) * 0.1
    basis = torch.randn(es, 1 + K, Wd, generator=g) * 0.05
    B, S = 2, 64
    faces = torch.rand(B, S, S, 6, generator=g) * 2 - 1
    tokens = torch.randint(0, 90, (B, 77), generator=g)
    tokens[0, 4] = 99; tokens[0, 9] = 98; tokens[1, 2] = 98; tokens[1, 30] = 99
    emb = torch.randn(B, 77, Wd, generator=g)
    ori = dict(faces=faces, ids=torch.tensor([[1, 2], [2, 1]]), num_ids=torch.tensor([2, 2]))
    emb_dev = dv(emb)
    out = em(dv(tokens), emb_dev, None, ori, dv(basis))
    ctx.sync()
    assert out is emb_dev, "the manager mutates embedded_text in place and returns it (embedding_manager.py:347-394)"
    Wh = {"embedding_manager.meta_id_net.stylegan_mlp.net.0.weight": em.meta_id_net.weight.data,
          "embedding_manager.meta_id_net.stylegan_mlp.net.0.bias": em.meta_id_net.bias.data, "celeb_embeddings": basis}
    with torch.no_grad():
        stack = torch.cat([faces[..., :3], faces[..., 3:]], 0)                           # chunk(num, -1) -> cat(0), meta_net.py:337-338
        _, small = O.face_warp_resize(stack, fr["in_size"])
        v = torch.nn.functional.normalize(O.frnet_forward(Wfr, fr, small), dim=-1, p=2)
        _, z = O.celeb_head(Wh, v, es, K)                                                # [2B, es, Wd]
        ref = torch.stack([O.shift_rows_multi(tokens[b], emb[b], [z[b], z[B + b]], [99, 98], es) for b in range(B)])
    close("manager.image_mode", out, ref, 2e-4)


def test_prefetch_pipeline_is_bitwise_identical(cb):
    """cb_trainer_prefetch (first stage of the next batch on its own stream) must not change a single bit: 4 steps with the
    pipeline (incl. one batch that was never prefetched and one prefetched batch that is skipped) against 4 plain steps"""
    from oracle import cb_oracle as O
    lib, ctx = cb
    m = MINI
    shapes = O.trainer_param_shapes(m["unet"], m["vae"], m["clip"], m["fr"], es=m["es"], inner=m["inner"], max_ids=m["max_ids"])
    W = O.init_weights(shapes, seed=9)
    lat = m["image"] // (2 ** (len(m["vae"]["ch_mult"]) - 1))
    batches = [{k: dv(v) for k, v in make_batch(2, m["image"], lat, m["clip"]["vocab"], m["placeholder"], 300 + s, m["max_ids"]).items()}
               for s in range(5)]
    step = lambda tr, b: tr.step(b["image"], b["faces"], b["ids"], b["tokens"], b["t"], b["noise"], b["post_noise"], 5e-3)
    pre = lambda tr, b: tr.prefetch(b["image"], b["faces"], b["post_noise"])
    results = []
    for pipelined in (False, True):
        tr = lib.Trainer(ctx, lib.WeightTable(W), trainer_cfg(lib, m))
        if pipelined:
            pre(tr, batches[0])
            pre(tr, batches[1]); step(tr, batches[0])
            pre(tr, batches[4]); step(tr, batches[1])          # batches[4] is prefetched but never stepped ...
            step(tr, batches[2])                               # ... and batches[2] is stepped without a prefetch
            pre(tr, batches[3]); step(tr, batches[3])          # prefetch immediately followed by its own step
        else:
            for i in range(4):
                step(tr, batches[i])
        ctx.sync()
        results.append({k: tr.get_state(k).clone() for k in ("loss", "mlp.weight", "mlp.bias", "id_coefficients", "id_embeddings")})
        tr.close()
    for k in results[0]:
        assert torch.equal(results[0][k], results[1][k]), f"pipelined run differs in {k}"


def mini_latent_diffusion(seed):
    """the reference-facing LatentDiffusion at mini sizes with oracle-seeded weights -> (model, sizes, weights); shared with the
    one-rank RCCL merge test (tests/test_gpu_dist.py)"""
    from celebbasis_amd.ldm.models.diffusion.ddpm import LatentDiffusion
    from celebbasis_amd.ldm.modules.id_embedding.iresnet import IResNet
    from oracle import cb_oracle as O
    m = dict(MINI, clip=dict(vocab=49408, width=64, layers=1, heads=4, ctx_len=77, mlp=128), placeholder=48136, max_ids=10)
    T = lambda t, p: dict(target=t, params=p)
    model = LatentDiffusion(
        first_stage_config=T("celebbasis_amd.ldm.models.autoencoder.AutoencoderKL",
                             dict(embed_dim=4, ddconfig=dict(double_z=True, z_channels=4, resolution=64, in_channels=3, out_ch=3, ch=32,
                                                             ch_mult=[1, 2, 2], num_res_blocks=1, attn_resolutions=[], dropout=0.0))),
        cond_stage_config=T("celebbasis_amd.ldm.modules.encoders.modules.FrozenCLIPEmbedder",
                            dict(vocab=49408, width=64, layers=1, heads=4, mlp=128, num_embeds_per_token=2, n_components=16)),
        unet_config=T("celebbasis_amd.ldm.modules.diffusionmodules.openaimodel.UNetModel",
                      dict(image_size=16, in_channels=4, out_channels=4, model_channels=32, attention_resolutions=[1, 2], num_res_blocks=1,
                           channel_mult=[1, 2, 2], num_heads=2, use_spatial_transformer=True, transformer_depth=1, context_dim=64)),
        personalization_config=T("celebbasis_amd.ldm.modules.embedding_manager.EmbeddingManagerId",
                                 dict(placeholder_strings=["sks"], max_ids=10, num_embeds_per_token=2, meta_mlp_depth=1, loss_type="none",
                                      meta_inner_dim=16, test_mode="coefficient", momentum=0.99, save_fp16=False)),
        linear_start=0.00085, linear_end=0.012, timesteps=1000, scale_factor=0.18215, image_size=16, channels=4)
    fr = m["fr"]
    model.embedding_manager.meta_id_net.id_model = IResNet(fr["layers"], fr["feat"], fr["width"], fr["in_size"])
    shapes = O.trainer_param_shapes(m["unet"], m["vae"], m["clip"], fr, es=2, inner=16, max_ids=10)
    W = O.init_weights(shapes, seed=seed)
    model.load_state_dict(W)
    em = model.embedding_manager
    em.meta_id_net.weight.data = W["embedding_manager.meta_id_net.stylegan_mlp.net.0.weight"].clone()
    em.meta_id_net.bias.data = W["embedding_manager.meta_id_net.stylegan_mlp.net.0.bias"].clone()
    em.id_coefficients = [W["id_coefficients"][i].reshape(2, 1, 16).clone() for i in range(10)]
    em.id_embeddings = [W["id_embeddings"][i].reshape(2, 64).clone() for i in range(10)]
    return model, m, W


def test_host_training_step_mini_vs_oracle(cb, tmp_path):
    """the reference-facing LatentDiffusion.training_step (+ prefetch_first_stage, on_save_checkpoint) at mini sizes against the
    oracle on a pinned replay stream: loss, MLP weights, coefficient dictionary, and the embeddings_gs-*.pt written at the end"""
    from oracle import cb_oracle as O
    lib, ctx = cb
    model, m, W = mini_latent_diffusion(seed=41)
    em, fr = model.embedding_manager, m["fr"]
    params = em.trainable_parameters()
    model.learning_rate = 5e-3
    model.train()
    st = O.TrainState(W)
    raw = [make_batch(2, 64, 16, 40000, 48136, 500 + s, 10) for s in range(3)]
    dev_batches = []
    for b in raw:
        dev_batches.append(dict(image=dv(b["image"]), tokens=b["tokens"], t=b["t"], noise=b["noise"], post_noise=b["post_noise"],
                                image_ori=dict(faces=b["faces"], ids=b["ids"], num_ids=torch.ones(2, dtype=torch.int64))))
    for i, b in enumerate(dev_batches):
        if i + 1 < len(dev_batches):
            model.prefetch_first_stage(dev_batches[i + 1])
        loss = model.training_step(b, i)
        loss_ref = O.train_step(W, st, raw[i], 5e-3, m["unet"], m["vae"], m["clip"], fr, es=2, inner=16, placeholder=48136)
        assert isinstance(loss, torch.Tensor) and loss.dim() == 0 and loss.is_cuda      # training_step returns the loss (ddpm.py:357-370)
        close(f"host_step.s{i}.loss", loss.reshape(1), loss_ref.reshape(1), 1e-4)
        assert abs(float(loss) - model.last_loss()) == 0.0
    model.on_save_checkpoint({}, ckptdir=str(tmp_path))
    ck = torch.load(tmp_path / "embeddings_gs-3.pt", weights_only=False)
    coef = torch.stack([c.reshape(-1) for c in ck["id_coefficients"]])
    assert (coef - st.id_coef).norm(dim=1).max().item() <= 1e-3
    close("host_step.mlp.weight", em.meta_id_net.weight.reshape(-1), st.Wm.reshape(-1), 1e-4)
    # the Parameter objects handed out before training are the ones the sync refreshed
    assert params[0] is em.meta_id_net.weight and params[1] is em.meta_id_net.bias
    assert isinstance(model.configure_optimizers(), torch.optim.AdamW)
    # stand-alone q_sample (ddpm.py:289-292)
    g = torch.Generator().manual_seed(5)
    x0 = torch.randn(2, 4, 16, 16, generator=g); nz = torch.randn(2, 4, 16, 16, generator=g); tt = torch.tensor([10, 900])
    xt = model.q_sample(dv(x0), dv(tt), dv(nz))
    sa = model.sqrt_alphas_cumprod[tt].reshape(2, 1, 1, 1); sb = model.sqrt_one_minus_alphas_cumprod[tt].reshape(2, 1, 1, 1)
    close("host_step.q_sample", xt, sa * x0 + sb * nz, 1e-6)


def test_use_rm_mlp_mini_vs_oracle(cb):
    """the ablation switch `use_rm_mlp` (meta_net.py:173-175, 272-273) — the literal "optimise the 1024 coefficients" mode: the
    (max_ids, es, 1, inner) coefficients are the parameter, x = coef[id_idx] (no FR-net, no MLP, no L2-norm), AdamW on the whole
    tensor. Three steps at mini sizes against the oracle branch: loss, dense gradient, parameter, EMA dictionary."""
    from oracle import cb_oracle as O
    lib, ctx = cb
    m = MINI
    shapes = O.trainer_param_shapes(m["unet"], m["vae"], m["clip"], m["fr"], es=m["es"], inner=m["inner"], max_ids=m["max_ids"])
    shapes["embedding_manager.meta_id_net.coef"] = (m["max_ids"], m["es"], 1, m["inner"])
    W = O.init_weights(shapes, seed=17)
    W["embedding_manager.meta_id_net.coef"] = torch.randn(shapes["embedding_manager.meta_id_net.coef"], generator=torch.Generator().manual_seed(18))
    cfg = trainer_cfg(lib, m)
    cfg.use_rm_mlp = 1
    tr = lib.Trainer(ctx, lib.WeightTable(W), cfg)
    st = O.TrainState(W)
    lat = m["image"] // (2 ** (len(m["vae"]["ch_mult"]) - 1))
    for s in range(3):
        batch = make_batch(2, m["image"], lat, m["clip"]["vocab"], m["placeholder"], 700 + s, m["max_ids"])
        dev = {k: dv(v) for k, v in batch.items()}
        tr.step(dev["image"], dev["faces"], dev["ids"], dev["tokens"], dev["t"], dev["noise"], dev["post_noise"], 1e-2)
        loss_ref, aux = O.train_step(W, st, batch, 1e-2, m["unet"], m["vae"], m["clip"], m["fr"], es=m["es"], inner=m["inner"],
                                     placeholder=m["placeholder"], return_aux=True, use_rm_mlp=True)
        ctx.sync()
        close(f"rm_mlp.s{s}.loss", tr.get_state("loss"), loss_ref.reshape(1), 1e-4)
        close(f"rm_mlp.s{s}.grad_coef", tr.get_state("rm.grad_coef"), aux["gW"].reshape(-1), 1e-3)
    close("rm_mlp.coef", tr.get_state("rm.coef"), st.coef.reshape(-1), 1e-4)
    l2 = (tr.get_state("id_coefficients").reshape(m["max_ids"], -1) - st.id_coef).norm(dim=1).max().item()
    assert l2 <= 1e-3, l2
    with pytest.raises(lib.CBError):
        lib.Trainer(ctx, lib.WeightTable({k: v for k, v in W.items() if not k.endswith(".coef")}), cfg)      # the parameter must be in the table
    tr.close()


def test_two_person_prompt_conditioning_vs_oracle(cb):
    """stable_txt2img.py:310-318 always passes num_ids = 2 with ids = [eval_id1, eval_id2 | eval_id1]: 'sks' takes the first
    identity and 'ks' the second (README two-person usage); a prompt without 'ks' is unaffected by the second identity.
    Eval-mode manager (test_mode 'coefficient') + CLIP against the oracle composition."""
    from celebbasis_amd.ldm.modules.embedding_manager import EmbeddingManagerId
    from celebbasis_amd.ldm.modules.encoders.modules import FrozenCLIPEmbedder
    from oracle import cb_oracle as O
    lib, ctx = cb
    ccfg = dict(vocab=49408, width=64, layers=1, heads=4, ctx_len=77, mlp=128)
    es, K, Wd = 2, 16, 64
    cs = FrozenCLIPEmbedder(vocab=49408, width=Wd, layers=1, heads=4, mlp=128, num_embeds_per_token=es, n_components=K)
    Wc = O.init_weights({FrozenCLIPEmbedder.PREFIX + k: v for k, v in FrozenCLIPEmbedder.param_shapes(ccfg).items()}, seed=51)
    cs.load_state_dict(Wc)
    g = torch.Generator().manual_seed(52)
    basis = torch.randn(es, 1 + K, Wd, generator=g) * 0.05
    cs.load_celeb_basis(basis)
    em = EmbeddingManagerId(cs, placeholder_strings=["sks", "ks", "ata", "tre"], max_ids=10, num_embeds_per_token=es, meta_mlp_depth=1,
                            loss_type="none", meta_inner_dim=K, test_mode="coefficient", save_fp16=False).eval()
    assert em.string_to_token_dict["sks"] == 48136 and em.string_to_token_dict["ks"] == 662        # modules.py:198-262
    em.id_coefficients = [torch.randn(es, 1, K, generator=g) for _ in range(10)]
    B = 2
    tokens = torch.full((B, 77), 49407, dtype=torch.int64); tokens[:, 0] = 49406
    tokens[0, 1:8] = torch.tensor([320, 1125, 539, 48136, 2533, 537, 662])      # "... sks person and ks"
    tokens[1, 1:5] = torch.tensor([320, 1125, 539, 48136])                      # no 'ks' in this prompt
    id1, id2 = 3, 7
    ori = dict(faces=None, ids=torch.tensor([[id1, id2]] * B), num_ids=torch.ones(B, dtype=torch.int64) * 2)
    cond = cs.encode(dv(tokens), embedding_manager=em, image_ori=ori)
    ctx.sync()
    with torch.no_grad():
        zs = [torch.einsum("ehk,ekc->ehc", em.id_coefficients[i], basis[:, 1:]).reshape(es, Wd) + basis[:, 0] for i in (id1, id2)]
        table = Wc[FrozenCLIPEmbedder.PREFIX + "embeddings.token_embedding.weight"]
        emb = torch.stack([O.shift_rows_multi(tokens[b], table[tokens[b]], zs, [48136, 662], es) for b in range(B)])
        ref = O.clip_encode(Wc, ccfg, emb, FrozenCLIPEmbedder.PREFIX)
    close("two_person.cond", cond, ref, 2e-4)
    ori1 = dict(faces=None, ids=torch.tensor([[id1, id1]] * B), num_ids=torch.ones(B, dtype=torch.int64) * 2)
    cond1 = cs.encode(dv(tokens), embedding_manager=em, image_ori=ori1)
    ctx.sync()
    assert torch.equal(cond1[1], cond[1]) and not torch.equal(cond1[0], cond[0])


def test_run_config5_tool_first_identity_vs_oracle(cb, tmp_path):
    """tools/run_config5.py is the config-5 driver whose numbers DESIGN quotes: run it for real (SD v1-4 size, one identity, one
    bs=2 training step, a 2-step DDIM with 2 samples, decode) and replay the training of its first identity on the CPU oracle
    from the replay description it dumps — the learned coefficient row must be inside north_star's 1e-3 L2 bar, the images finite
    and non-degenerate, the JSON line complete"""
    import subprocess
    import sys
    from celebbasis_amd import synthetic
    from oracle import cb_oracle as O
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dump = tmp_path / "first.pt"
    # the tool's replay description for these flags is known before it runs (its defaults: weight seed 0, batch seed 23 + rank, slot 0,
    # lr 5e-3 * bs): the oracle replays it on the host cores WHILE the subprocess has the GPU, and the dump is then held to that description
    want = dict(steps=1, batch=2, lr=5e-3 * 2, weight_seed=0, batch_seed=23, slot=0)
    torch.set_num_threads(min(64, os.cpu_count() or 1))

    def replay():
        W = synthetic.synthetic_state(dict(synthetic.trainer_shapes()), seed=want["weight_seed"])
        st = O.TrainState(W)
        for s_ in range(want["steps"]):
            b = synthetic.synthetic_batch(want["batch"], step=s_, identity=0, seed=want["batch_seed"])
            b["ids"] = torch.full((want["batch"], 2), want["slot"], dtype=torch.int64)
            O.train_step(W, st, b, want["lr"])
        return st
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=1) as ex:
        fut = ex.submit(replay)
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "run_config5.py"), "--identities", "1", "--train-steps", "1", "--ddim-steps", "2",
                            "--n-samples", "2", "--dump-first", str(dump)], capture_output=True, text=True, timeout=900, cwd=root)
        st = fut.result()
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["merged_identities"] == 1 and line["finite"] is True and line["first_identity_dump"] == str(dump)
    fx = torch.load(dump, weights_only=False)
    assert fx["image_shape"] == [2, 3, 512, 512] and 0.0 < fx["image_mean"] < 1.0 and fx["image_std"] > 1e-3
    assert {k: fx[k] for k in want} == want and fx["n_batches"] >= want["steps"], {k: fx[k] for k in want}
    l2 = float((fx["coef"] - st.id_coef[fx["slot"]]).norm())
    REPORT["run_config5.first_identity.coef_l2"] = l2
    assert l2 <= 1e-3, f"tools/run_config5.py: first identity's coefficients are {l2:.3e} from the oracle"


def test_main_cli_trains_from_the_device_feeder_and_writes_the_reference_checkpoints(cb, tmp_path):
    """`python -m celebbasis_amd.main` itself (main_id_embed.py's flags, the shipped aigc_id.yaml values, SD v1-4 size, seeded synthetic
    weights / faces): a few steps from the default device feeder in fp32, then bf16 + `--pipeline` at bs=2 — the process exits 0, the
    wall-clock report is complete, and `embeddings.pt` / `embeddings_gs-N.pt` carry the reference's format (embedding_manager.py:396-410:
    `id_coefficients` = max_ids tensors (es, heads, K)) with the trained slots moved off their initial value and everything finite"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    runs = (("f32", [], 1), ("bf16", ["--precision", "bf16", "--pipeline"], 2))

    def launch(run):
        tag, extra, bs = run
        logdir, report = tmp_path / tag, tmp_path / (tag + ".json")
        cmd = [sys.executable, "-m", "celebbasis_amd.main", "--base", os.path.join(root, "tests", "golden", "aigc_id_params.json"), "-t", "--synthetic",
               "--max_steps", "6", "--report", str(report), "--report_warmup", "2", "-l", str(logdir), f"data.params.batch_size={bs}"] + extra
        return subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root)
    results = [launch(r_) for r_ in runs]          # (side by side they took 43 - 47 s against 34 s one after the other: both synthesise and upload 4 GB of weights)
    for (tag, extra, bs), r in zip(runs, results):
        logdir, report = tmp_path / tag, tmp_path / (tag + ".json")
        assert r.returncode == 0, r.stderr[-3000:]
        rec = json.loads(report.read_text())
        assert rec["feeder"] == "device" and rec["steps_timed"] == 4 and rec["batch_size"] == bs and rec["precision"] == tag
        assert rec["pipeline"] == ("--pipeline" in extra) and rec["steps_per_sec"] > 1.0
        ck = logdir / "celebbasis" / "checkpoints"
        last, numbered = torch.load(ck / "embeddings.pt", weights_only=False), torch.load(ck / "embeddings_gs-6.pt", weights_only=False)
        coefs = last["id_coefficients"]
        assert len(coefs) == 10 and all(tuple(c.shape) == (2, 1, 512) for c in coefs)                     # max_ids x (es, heads, K), aigc_id.yaml:83-91
        assert all(torch.equal(a, b) for a, b in zip(coefs, numbered["id_coefficients"]))
        assert all(bool(torch.isfinite(c.float()).all()) for c in coefs)
        # the dictionary starts as ONE random tensor repeated max_ids times (embedding_manager.py:229-230): slots nobody trained still
        # hold it, the slots of the two training identities (num_ids = 2; which of them six shuffled samples hit is seed-dependent) moved
        assert all(torch.equal(c, coefs[2]) for c in coefs[3:])
        assert any(not torch.equal(c, coefs[2]) for c in coefs[:2])
        assert "loss" in r.stdout and "nan" not in r.stdout.lower()


def test_config5_mini_personalise_then_sample_vs_oracle(cb):
    """BASELINE config 5 at mini sizes, the two halves of one run (01_start_train.sh + 02_start_test.sh) for several identities
    one after the other on ONE trainer: reset the coefficient head + optimiser, bs=2 training steps, then sample with the learned
    coefficients (eval-mode manager `test_mode: coefficient`, embedding_manager.py:464-475 -> CLIP -> guided DDIM n_samples=2 ->
    VAE decode) through the trainer's shared UNet / CLIP views. Against the oracle: per-identity coefficients and the images."""
    import ctypes as C
    from oracle import cb_oracle as O
    lib, ctx = cb
    m = MINI
    es, K, Wd = m["es"], m["inner"], m["clip"]["width"]
    shapes = O.trainer_param_shapes(m["unet"], m["vae"], m["clip"], m["fr"], es=es, inner=K, max_ids=m["max_ids"])
    W = O.init_weights(shapes, seed=61)
    Wdec = O.init_weights(O.vae_decoder_param_shapes(m["vae"]), seed=62)
    tr = lib.Trainer(ctx, lib.WeightTable(W), trainer_cfg(lib, m))
    unet, clip = tr.unet(), tr.clip()
    dec = lib.VAEDecoder(ctx, lib.WeightTable(Wdec), lib.vae_cfg(**m["vae"]))
    L = lib.load()
    lat = m["image"] // (2 ** (len(m["vae"]["ch_mult"]) - 1))
    n_samples, B, steps, lr = 2, 2, 3, 1e-2
    sched = O.ddim_schedule(O.make_schedule()["alphas_cumprod"], 4, 0.0)
    cp, up = "cond_stage_model.transformer.text_model.", "model.diffusion_model."
    table = W[cp + "embeddings.token_embedding.weight"]
    basis = W["celeb_embeddings"]
    g = torch.Generator().manual_seed(63)
    prompt = torch.randint(0, m["clip"]["vocab"] - 30, (77,), generator=g); prompt[4] = m["placeholder"]
    empty = torch.randint(0, m["clip"]["vocab"] - 30, (77,), generator=g)
    w0, b0 = W["embedding_manager.meta_id_net.stylegan_mlp.net.0.weight"], W["embedding_manager.meta_id_net.stylegan_mlp.net.0.bias"]
    for ident in range(3):
        slot = ident % m["max_ids"]
        tr.set_state("mlp.weight", w0); tr.set_state("mlp.bias", b0); tr.set_state("step", torch.zeros(1))
        for name in ("adam.m_weight", "adam.v_weight"):
            tr.set_state(name, torch.zeros(w0.numel()))
        for name in ("adam.m_bias", "adam.v_bias"):
            tr.set_state(name, torch.zeros(b0.numel()))
        st = O.TrainState(W)
        if ident:                                  # the EMA dictionaries carry over between identities (one manager per run)
            st.id_coef, st.id_emb = coef_prev.clone(), emb_prev.clone()
        for s in range(steps):
            batch = make_batch(B, m["image"], lat, m["clip"]["vocab"], m["placeholder"], 700 + 10 * ident + s, m["max_ids"])
            batch["ids"] = torch.full_like(batch["ids"], slot)
            d = {k: dv(v) for k, v in batch.items()}
            tr.step(d["image"], d["faces"], d["ids"], d["tokens"], d["t"], d["noise"], d["post_noise"], lr)
            O.train_step(W, st, batch, lr, m["unet"], m["vae"], m["clip"], m["fr"], es=es, inner=K, placeholder=m["placeholder"])
        ctx.sync()
        coef_prev, emb_prev = st.id_coef, st.id_emb
        coef = tr.get_state("id_coefficients").reshape(m["max_ids"], es * K)[slot]
        l2 = (coef - st.id_coef[slot]).norm().item()
        REPORT[f"config5_mini.id{ident}.coef_l2"] = l2
        assert l2 <= 1e-3
        # ---- sampling half, on the device
        c_dev = dv(coef.reshape(es, K)); basis_dev = dv(basis)
        z = torch.empty(es, Wd, device=DEV)
        for e in range(es):
            gd = lib.GemmDesc()
            gd.A, gd.B, gd.C = c_dev[e].data_ptr(), basis_dev[e, 1:].data_ptr(), z[e].data_ptr()
            gd.M, gd.N, gd.K = 1, Wd, K
            gd.lda, gd.ldb, gd.ldc = K, Wd, Wd
            gd.a_layout, gd.b_layout, gd.batch = 0, 1, 1
            gd.bias = basis_dev[e, 0].data_ptr(); gd.alpha, gd.beta = 1.0, 0.0
            lib.check(L.cb_gemm(ctx.h, C.byref(gd)))
        toks = dv(prompt.unsqueeze(0).repeat(n_samples, 1)); etoks = dv(empty.unsqueeze(0).repeat(n_samples, 1))
        emb = clip.embed_tokens(toks)
        out = torch.empty_like(emb); mp = torch.empty(n_samples * 77, dtype=torch.int32, device=DEV)
        zrep = z.unsqueeze(0).repeat(n_samples, 1, 1).contiguous()
        lib.check(L.cb_row_surgery(ctx.h, lib.ptr(emb), lib.ptr(zrep), lib.ptr(toks, torch.int64), lib.ptr(out), C.c_void_p(mp.data_ptr()),
                                   n_samples, 77, Wd, es, m["placeholder"]))
        cond = clip.forward(out)
        uncond = clip.forward(clip.embed_tokens(etoks))
        x_T = torch.randn(n_samples, 4, lat, lat, generator=g)
        x, _ = unet.ddim_sample(dv(x_T), cond, uncond, sched["timesteps"], sched["alphas"], sched["alphas_prev"], sched["sigmas"],
                                sched["sqrt_one_minus_alphas"], scale=10.0)
        img = dec.decode(x, 0.18215, to01=True)
        ctx.sync()
        with torch.no_grad():
            zr = torch.einsum("ek,ekc->ec", st.id_coef[slot].reshape(es, K), basis[:, 1:]) + basis[:, 0]
            er = O.shift_rows(prompt, table[prompt], zr, m["placeholder"], es)
            cr = O.clip_encode(W, m["clip"], er.unsqueeze(0).repeat(n_samples, 1, 1), cp)
            ur = O.clip_encode(W, m["clip"], table[empty].unsqueeze(0).repeat(n_samples, 1, 1), cp)
            xr, _ = O.ddim_sample(W, m["unet"], x_T, cr, ur, sched, scale=10.0, prefix=up)
            ir = torch.clamp((O.decode_first_stage(Wdec, m["vae"], xr, 0.18215) + 1.0) / 2.0, 0.0, 1.0)
        close(f"config5_mini.id{ident}.cond", cond, cr, 2e-4)
        close(f"config5_mini.id{ident}.latents", x, xr, 1e-3)
        close(f"config5_mini.id{ident}.image01", img, ir, 1e-3)
    dec.close(); tr.close()
