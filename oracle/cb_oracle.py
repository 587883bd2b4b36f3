"""CPU ORACLE — TEST INFRASTRUCTURE ONLY (never imported by the product path `celebbasis_amd/`).

A from-scratch, functional torch-CPU fp32 restatement of the CelebBasis personalization training step of
ygtxr1997/CelebBasis. Each function cites the reference file:line it follows. It is used only
  * by tests/ as the checker of the HIP path (same seeded inputs, small sizes),
  * by `__graft_entry__.smoke()` as the checker,
  * by bench.py's `cpu_baseline` leg (kind "port") as the thing timed on host cores.

PINNING. The restatement is pinned against the reference itself: oracle/gen_golden.py imports the reference's
own modules from /root/reference (UNetModel, Encoder, ResBlock, SpatialTransformer, iresnet, MetaIdNet /
EmbeddingManagerId, shift_tensor_dim0, DDPM schedule, ...) in the build container, feeds them seeded inputs and
writes the input/output vectors to tests/golden/*.pt; tests/test_oracle_golden.py replays them through this file.
The CLIP text layers live in the third-party dependency transformers==4.18.0 (environment.yaml:199), which is not
vendored and whose installed version (5.x) is API-incompatible with the reference's monkey-patch; they are
restated from the published 4.18 `modeling_clip` algorithm and cross-checked against the installed
`transformers.CLIPTextModel` arithmetic (same weights) in gen_golden.py.
The feeder's colour jitter lives in torchvision==0.11.3 (environment.yaml:57, absent here) on top of PIL (installed): the
restatement is pinned against PIL's ImageEnhance / HSV results to uint8 quantisation (tests/test_host_cpu.py::
test_oracle_jitter_vs_pil); no reference-produced fixture exists for it.

Weights are plain {state-dict key: tensor} dicts using the reference's key names.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------- configs (aigc_id.yaml)
UNET_SD14 = dict(in_channels=4, out_channels=4, model_channels=320, num_res_blocks=2, channel_mult=(1, 2, 4, 4),
                 attention_resolutions=(4, 2, 1), num_heads=8, context_dim=768)             # aigc_id.yaml:39-54
VAE_SD14 = dict(ch=128, ch_mult=(1, 2, 4, 4), num_res_blocks=2, in_channels=3, z_channels=4)  # aigc_id.yaml:56-77
CLIP_L14 = dict(vocab=49408, width=768, layers=12, heads=12, ctx_len=77, mlp=3072)
FR_R100 = dict(layers=(3, 13, 30, 3), width=64, in_size=112, feat=512)                       # iresnet.py:222-225
TRANS_MATRIX = [[1.07695457, -0.03625215, -1.56352194 / 512], [0.03625215, 1.07695457, -5.32134629 / 512]]  # meta_net.py:131-142
PLACEHOLDER_SKS = 48136                                                                      # modules.py:198-262


# ----------------------------------------------------------------------------- parameter shape tables
def unet_param_shapes(cfg, prefix=""):
    """state-dict keys/shapes of UNetModel (openaimodel.py:413-742) for use_spatial_transformer, legacy=False"""
    mc = cfg["model_channels"]; ted = 4 * mc; cd = cfg["context_dim"]
    S = {}

    def lin(p, i, o, bias=True):
        S[p + ".weight"] = (o, i)
        if bias:
            S[p + ".bias"] = (o,)

    def conv(p, i, o, k):
        S[p + ".weight"] = (o, i, k, k); S[p + ".bias"] = (o,)

    def norm(p, c):
        S[p + ".weight"] = (c,); S[p + ".bias"] = (c,)

    def res(p, ci, co):
        norm(p + ".in_layers.0", ci); conv(p + ".in_layers.2", ci, co, 3); lin(p + ".emb_layers.1", ted, co)
        norm(p + ".out_layers.0", co); conv(p + ".out_layers.3", co, co, 3)
        if ci != co:
            conv(p + ".skip_connection", ci, co, 1)

    def xf(p, ch):
        norm(p + ".norm", ch); conv(p + ".proj_in", ch, ch, 1)
        tb = p + ".transformer_blocks.0"
        for a, kd in (("attn1", ch), ("attn2", cd)):
            lin(f"{tb}.{a}.to_q", ch, ch, False); lin(f"{tb}.{a}.to_k", kd, ch, False); lin(f"{tb}.{a}.to_v", kd, ch, False)
            lin(f"{tb}.{a}.to_out.0", ch, ch)
        lin(tb + ".ff.net.0.proj", ch, 8 * ch); lin(tb + ".ff.net.2", 4 * ch, ch)
        for n in ("norm1", "norm2", "norm3"):
            norm(f"{tb}.{n}", ch)
        conv(p + ".proj_out", ch, ch, 1)

    lin(prefix + "time_embed.0", mc, ted); lin(prefix + "time_embed.2", ted, ted)
    conv(prefix + "input_blocks.0.0", cfg["in_channels"], mc, 3)
    chans = [mc]; ch = mc; ds = 1; idx = 1
    nlev = len(cfg["channel_mult"])
    for level, mult in enumerate(cfg["channel_mult"]):
        for _ in range(cfg["num_res_blocks"]):
            res(f"{prefix}input_blocks.{idx}.0", ch, mult * mc); ch = mult * mc
            if ds in cfg["attention_resolutions"]:
                xf(f"{prefix}input_blocks.{idx}.1", ch)
            chans.append(ch); idx += 1
        if level != nlev - 1:
            conv(f"{prefix}input_blocks.{idx}.0.op", ch, ch, 3); chans.append(ch); ds *= 2; idx += 1
    res(prefix + "middle_block.0", ch, ch); xf(prefix + "middle_block.1", ch); res(prefix + "middle_block.2", ch, ch)
    o = 0
    for level, mult in list(enumerate(cfg["channel_mult"]))[::-1]:
        for i in range(cfg["num_res_blocks"] + 1):
            ich = chans.pop()
            res(f"{prefix}output_blocks.{o}.0", ch + ich, mc * mult); ch = mc * mult
            sub = 1
            if ds in cfg["attention_resolutions"]:
                xf(f"{prefix}output_blocks.{o}.1", ch); sub = 2
            if level and i == cfg["num_res_blocks"]:
                conv(f"{prefix}output_blocks.{o}.{sub}.conv", ch, ch, 3); ds //= 2
            o += 1
    norm(prefix + "out.0", ch); conv(prefix + "out.2", mc, cfg["out_channels"], 3)
    return S


def vae_encoder_param_shapes(cfg, prefix=""):
    """AutoencoderKL encoder half (model.py:368-432, autoencoder.py:302)"""
    S = {}

    def conv(p, i, o, k):
        S[p + ".weight"] = (o, i, k, k); S[p + ".bias"] = (o,)

    def norm(p, c):
        S[p + ".weight"] = (c,); S[p + ".bias"] = (c,)

    def res(p, ci, co):
        norm(p + ".norm1", ci); conv(p + ".conv1", ci, co, 3); norm(p + ".norm2", co); conv(p + ".conv2", co, co, 3)
        if ci != co:
            conv(p + ".nin_shortcut", ci, co, 1)

    e = prefix + "encoder."
    conv(e + "conv_in", cfg["in_channels"], cfg["ch"], 3)
    bi = cfg["ch"]
    n = len(cfg["ch_mult"])
    for l, m in enumerate(cfg["ch_mult"]):
        bo = cfg["ch"] * m
        for b in range(cfg["num_res_blocks"]):
            res(f"{e}down.{l}.block.{b}", bi, bo); bi = bo
        if l != n - 1:
            conv(f"{e}down.{l}.downsample.conv", bi, bi, 3)
    res(e + "mid.block_1", bi, bi)
    norm(e + "mid.attn_1.norm", bi)
    for q in ("q", "k", "v", "proj_out"):
        conv(f"{e}mid.attn_1.{q}", bi, bi, 1)
    res(e + "mid.block_2", bi, bi)
    norm(e + "norm_out", bi); conv(e + "conv_out", bi, 2 * cfg["z_channels"], 3)
    conv(prefix + "quant_conv", 2 * cfg["z_channels"], 2 * cfg["z_channels"], 1)
    return S


def vae_decoder_param_shapes(cfg, prefix=""):
    """AutoencoderKL decoder half (model.py:462-533, autoencoder.py:303)"""
    S = {}

    def conv(p, i, o, k):
        S[p + ".weight"] = (o, i, k, k); S[p + ".bias"] = (o,)

    def norm(p, c):
        S[p + ".weight"] = (c,); S[p + ".bias"] = (c,)

    def res(p, ci, co):
        norm(p + ".norm1", ci); conv(p + ".conv1", ci, co, 3); norm(p + ".norm2", co); conv(p + ".conv2", co, co, 3)
        if ci != co:
            conv(p + ".nin_shortcut", ci, co, 1)

    d = prefix + "decoder."
    n = len(cfg["ch_mult"])
    bi = cfg["ch"] * cfg["ch_mult"][-1]
    conv(prefix + "post_quant_conv", cfg["z_channels"], cfg["z_channels"], 1)
    conv(d + "conv_in", cfg["z_channels"], bi, 3)
    res(d + "mid.block_1", bi, bi)
    norm(d + "mid.attn_1.norm", bi)
    for q in ("q", "k", "v", "proj_out"):
        conv(f"{d}mid.attn_1.{q}", bi, bi, 1)
    res(d + "mid.block_2", bi, bi)
    for l in reversed(range(n)):
        bo = cfg["ch"] * cfg["ch_mult"][l]
        for b in range(cfg["num_res_blocks"] + 1):
            res(f"{d}up.{l}.block.{b}", bi, bo); bi = bo
        if l != 0:
            conv(f"{d}up.{l}.upsample.conv", bi, bi, 3)
    norm(d + "norm_out", bi); conv(d + "conv_out", bi, cfg["in_channels"], 3)
    return S


def clip_param_shapes(cfg, prefix=""):
    """HF CLIPTextModel.text_model state-dict (transformers 4.18 modeling_clip)"""
    w = cfg["width"]; S = {}
    S[prefix + "embeddings.token_embedding.weight"] = (cfg["vocab"], w)
    S[prefix + "embeddings.position_embedding.weight"] = (cfg["ctx_len"], w)
    for i in range(cfg["layers"]):
        p = f"{prefix}encoder.layers.{i}"
        for q in ("q_proj", "k_proj", "v_proj", "out_proj"):
            S[f"{p}.self_attn.{q}.weight"] = (w, w); S[f"{p}.self_attn.{q}.bias"] = (w,)
        for nme in ("layer_norm1", "layer_norm2"):
            S[f"{p}.{nme}.weight"] = (w,); S[f"{p}.{nme}.bias"] = (w,)
        S[f"{p}.mlp.fc1.weight"] = (cfg["mlp"], w); S[f"{p}.mlp.fc1.bias"] = (cfg["mlp"],)
        S[f"{p}.mlp.fc2.weight"] = (w, cfg["mlp"]); S[f"{p}.mlp.fc2.bias"] = (w,)
    S[prefix + "final_layer_norm.weight"] = (w,); S[prefix + "final_layer_norm.bias"] = (w,)
    return S


def frnet_param_shapes(cfg, prefix=""):
    """IResNet state-dict (iresnet.py:67-143)"""
    S = {}
    wd = cfg["width"]

    def bn(p, c):
        for k in ("weight", "bias", "running_mean", "running_var"):
            S[f"{p}.{k}"] = (c,)

    S[prefix + "conv1.weight"] = (wd, 3, 3, 3); bn(prefix + "bn1", wd); S[prefix + "prelu.weight"] = (wd,)
    inpl = wd
    for l in range(4):
        planes = wd << l
        for i in range(cfg["layers"][l]):
            p = f"{prefix}layer{l + 1}.{i}"
            bn(p + ".bn1", inpl); S[p + ".conv1.weight"] = (planes, inpl, 3, 3); bn(p + ".bn2", planes)
            S[p + ".prelu.weight"] = (planes,); S[p + ".conv2.weight"] = (planes, planes, 3, 3); bn(p + ".bn3", planes)
            if i == 0:
                S[p + ".downsample.0.weight"] = (planes, inpl, 1, 1); bn(p + ".downsample.1", planes)
            inpl = planes
    C = wd * 8
    bn(prefix + "bn2", C)
    fs = cfg["in_size"] // 16
    S[prefix + "fc.weight"] = (cfg["feat"], C * fs * fs); S[prefix + "fc.bias"] = (cfg["feat"],)
    bn(prefix + "features", cfg["feat"])
    return S


def trainer_param_shapes(unet=UNET_SD14, vae=VAE_SD14, clip=CLIP_L14, fr=FR_R100, es=2, inner=512, max_ids=10):
    """every tensor cb_trainer_create looks up (include/celebbasis_hip.h)"""
    S = {}
    S.update(unet_param_shapes(unet, "model.diffusion_model."))
    S.update(vae_encoder_param_shapes(vae, "first_stage_model."))
    S.update(clip_param_shapes(clip, "cond_stage_model.transformer.text_model."))
    S.update(frnet_param_shapes(fr, "embedding_manager.meta_id_net.id_model."))
    S["embedding_manager.meta_id_net.stylegan_mlp.net.0.weight"] = (es * inner, fr["feat"])
    S["embedding_manager.meta_id_net.stylegan_mlp.net.0.bias"] = (es * inner,)
    S["celeb_embeddings"] = (es, 1 + inner, clip["width"])
    S["id_embeddings"] = (max_ids, es * clip["width"])
    S["id_coefficients"] = (max_ids, es * inner)
    return S


def init_weights(shapes, seed=0, std=0.02):
    """Seeded synthetic weights: conv/linear N(0, fan-in scaled), norms gamma ~ 1, beta ~ 0,
    BatchNorm running_var in [0.5, 1.5]. Zero-modules of the reference are re-randomised so grads are non-trivial."""
    g = torch.Generator().manual_seed(seed)
    W = {}
    for k, shp in shapes.items():
        if k.endswith("running_var"):
            W[k] = 0.5 + torch.rand(shp, generator=g)
        elif k.endswith("running_mean"):
            W[k] = 0.1 * torch.randn(shp, generator=g)
        elif "prelu" in k:
            W[k] = 0.25 + 0.05 * torch.randn(shp, generator=g)
        elif len(shp) == 1 and (("norm" in k or ".bn" in k or "ln" in k or "features" in k or ".downsample.1" in k
                                  or k.endswith("in_layers.0.weight") or k.endswith("out_layers.0.weight") or ".out.0." in k)
                                 and k.endswith("weight")):
            W[k] = 1.0 + 0.1 * torch.randn(shp, generator=g)
        elif len(shp) == 1:
            W[k] = 0.05 * torch.randn(shp, generator=g)
        elif k == "celeb_embeddings":
            es, k1, w = shp
            q, _ = torch.linalg.qr(torch.randn(w, k1 - 1, generator=g))          # orthonormal rows like an SVD basis
            base = q.t().unsqueeze(0).repeat(es, 1, 1).clone()
            for e in range(1, es):
                base[e] = base[e].roll(e, dims=0)
            W[k] = torch.cat([0.02 * torch.randn(es, 1, w, generator=g), base], dim=1).contiguous()
        elif k in ("id_embeddings", "id_coefficients"):
            W[k] = torch.randn(shp, generator=g)
        elif k.endswith("stylegan_mlp.net.0.weight"):
            W[k] = torch.randn(shp, generator=g)                                   # EqualLinear init randn (meta_net.py:30)
        elif "embedding.weight" in k:
            W[k] = std * torch.randn(shp, generator=g)
        else:
            fan_in = int(np.prod(shp[1:]))
            W[k] = torch.randn(shp, generator=g) * (1.0 / math.sqrt(fan_in))
    return W


def formula_table(rows, cols, phase=0.0):
    """RNG-free deterministic table (fixtures reference it instead of storing large random tensors)"""
    i = torch.arange(rows, dtype=torch.float64)[:, None]
    c = torch.arange(cols, dtype=torch.float64)[None, :]
    return (0.02 * torch.sin(0.37 * i + 1.3 * c + 0.011 * i * c + phase)).float()


# ----------------------------------------------------------------------------- UNet (openaimodel.py, attention.py)
def timestep_embedding(t, dim, max_period=10000):
    """util.py:151-171"""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def _gn(W, p, x, eps):
    return F.group_norm(x.float(), 32, W[p + ".weight"], W[p + ".bias"], eps)


def _conv(W, p, x, stride=1, padding=1):
    return F.conv2d(x, W[p + ".weight"], W.get(p + ".bias"), stride=stride, padding=padding)


def _lin(W, p, x):
    return F.linear(x, W[p + ".weight"], W.get(p + ".bias"))


def resblock(W, p, x, emb):
    """openaimodel.py:255-275 ResBlock._forward (no up/down, use_scale_shift_norm=False)"""
    h = _conv(W, p + ".in_layers.2", F.silu(_gn(W, p + ".in_layers.0", x, 1e-5)))
    emb_out = _lin(W, p + ".emb_layers.1", F.silu(emb))
    h = h + emb_out[:, :, None, None]
    h = _conv(W, p + ".out_layers.3", F.silu(_gn(W, p + ".out_layers.0", h, 1e-5)))
    skip = x if (p + ".skip_connection.weight") not in W else _conv(W, p + ".skip_connection", x, padding=0)
    return skip + h


def cross_attention(W, p, x, context, heads):
    """attention.py:170-193"""
    context = x if context is None else context
    q = _lin(W, p + ".to_q", x); k = _lin(W, p + ".to_k", context); v = _lin(W, p + ".to_v", context)
    B, N, C = q.shape
    d = C // heads

    def split(t):
        return t.reshape(B, t.shape[1], heads, d).permute(0, 2, 1, 3).reshape(B * heads, t.shape[1], d)

    q, k, v = split(q), split(k), split(v)
    sim = torch.einsum("bid,bjd->bij", q, k) * (d ** -0.5)
    attn = sim.softmax(dim=-1)
    out = torch.einsum("bij,bjd->bid", attn, v)
    out = out.reshape(B, heads, N, d).permute(0, 2, 1, 3).reshape(B, N, C)
    return _lin(W, p + ".to_out.0", out)


def spatial_transformer(W, p, x, context, heads):
    """attention.py:250-261 + BasicTransformerBlock 211-215 + GEGLU 37-44"""
    b, c, h, w = x.shape
    x_in = x
    x = _gn(W, p + ".norm", x, 1e-6)
    x = _conv(W, p + ".proj_in", x, padding=0)
    x = x.permute(0, 2, 3, 1).reshape(b, h * w, c)
    tb = p + ".transformer_blocks.0"
    ln = lambda n, t: F.layer_norm(t, (c,), W[f"{tb}.{n}.weight"], W[f"{tb}.{n}.bias"], 1e-5)
    x = cross_attention(W, tb + ".attn1", ln("norm1", x), None, heads) + x
    x = cross_attention(W, tb + ".attn2", ln("norm2", x), context, heads) + x
    ff = _lin(W, tb + ".ff.net.0.proj", ln("norm3", x))
    a, gate = ff.chunk(2, dim=-1)
    x = _lin(W, tb + ".ff.net.2", a * F.gelu(gate)) + x
    x = x.reshape(b, h, w, c).permute(0, 3, 1, 2)
    x = _conv(W, p + ".proj_out", x, padding=0)
    return x + x_in


def unet_forward(W, cfg, x, t, context, prefix=""):
    """UNetModel.forward openaimodel.py:710-742"""
    mc = cfg["model_channels"]; heads = cfg["num_heads"]
    emb = _lin(W, prefix + "time_embed.2", F.silu(_lin(W, prefix + "time_embed.0", timestep_embedding(t, mc))))
    hs = []
    h = _conv(W, prefix + "input_blocks.0.0", x)
    hs.append(h)
    ds = 1; idx = 1
    nlev = len(cfg["channel_mult"])
    for level, mult in enumerate(cfg["channel_mult"]):
        for _ in range(cfg["num_res_blocks"]):
            h = resblock(W, f"{prefix}input_blocks.{idx}.0", h, emb)
            if ds in cfg["attention_resolutions"]:
                h = spatial_transformer(W, f"{prefix}input_blocks.{idx}.1", h, context, heads)
            hs.append(h); idx += 1
        if level != nlev - 1:
            h = _conv(W, f"{prefix}input_blocks.{idx}.0.op", h, stride=2, padding=1)       # Downsample openaimodel.py:134-160
            hs.append(h); ds *= 2; idx += 1
    h = resblock(W, prefix + "middle_block.0", h, emb)
    h = spatial_transformer(W, prefix + "middle_block.1", h, context, heads)
    h = resblock(W, prefix + "middle_block.2", h, emb)
    o = 0
    for level, mult in list(enumerate(cfg["channel_mult"]))[::-1]:
        for i in range(cfg["num_res_blocks"] + 1):
            h = torch.cat([h, hs.pop()], dim=1)
            h = resblock(W, f"{prefix}output_blocks.{o}.0", h, emb)
            sub = 1
            if ds in cfg["attention_resolutions"]:
                h = spatial_transformer(W, f"{prefix}output_blocks.{o}.1", h, context, heads); sub = 2
            if level and i == cfg["num_res_blocks"]:
                h = F.interpolate(h, scale_factor=2, mode="nearest")                        # Upsample openaimodel.py:91-119
                h = _conv(W, f"{prefix}output_blocks.{o}.{sub}.conv", h); ds //= 2
            o += 1
    return _conv(W, prefix + "out.2", F.silu(_gn(W, prefix + "out.0", h, 1e-5)))


# ----------------------------------------------------------------------------- VAE encoder (model.py)
def _vres(W, p, x):
    """model.py:121-141 ResnetBlock.forward with temb=None"""
    h = _conv(W, p + ".conv1", F.silu(_gn(W, p + ".norm1", x, 1e-6)))
    h = _conv(W, p + ".conv2", F.silu(_gn(W, p + ".norm2", h, 1e-6)))
    if (p + ".nin_shortcut.weight") in W:
        x = _conv(W, p + ".nin_shortcut", x, padding=0)
    return x + h


def vae_encode_moments(W, cfg, x, prefix=""):
    """Encoder.forward model.py:434-459 + quant_conv autoencoder.py:324-327. x NCHW"""
    e = prefix + "encoder."
    h = _conv(W, e + "conv_in", x)
    n = len(cfg["ch_mult"])
    for l in range(n):
        for b in range(cfg["num_res_blocks"]):
            h = _vres(W, f"{e}down.{l}.block.{b}", h)
        if l != n - 1:
            h = F.pad(h, (0, 1, 0, 1), mode="constant", value=0)                              # model.py:74-76
            h = _conv(W, f"{e}down.{l}.downsample.conv", h, stride=2, padding=0)
    h = _vres(W, e + "mid.block_1", h)
    a = e + "mid.attn_1"                                                                      # AttnBlock model.py:176-202
    hn = _gn(W, a + ".norm", h, 1e-6)
    q = _conv(W, a + ".q", hn, padding=0); k = _conv(W, a + ".k", hn, padding=0); v = _conv(W, a + ".v", hn, padding=0)
    b, c, hh, ww = q.shape
    q = q.reshape(b, c, hh * ww).permute(0, 2, 1); k = k.reshape(b, c, hh * ww)
    w_ = torch.bmm(q, k) * (int(c) ** (-0.5))
    w_ = F.softmax(w_, dim=2)
    v = v.reshape(b, c, hh * ww)
    h_ = torch.bmm(v, w_.permute(0, 2, 1)).reshape(b, c, hh, ww)
    h = h + _conv(W, a + ".proj_out", h_, padding=0)
    h = _vres(W, e + "mid.block_2", h)
    h = _conv(W, e + "conv_out", F.silu(_gn(W, e + "norm_out", h, 1e-6)))
    return _conv(W, prefix + "quant_conv", h, padding=0)


def vae_sample(moments, noise, scale=0.18215):
    """distributions.py:24-37 + ddpm.py:590-597"""
    mean, logvar = torch.chunk(moments, 2, dim=1)
    logvar = torch.clamp(logvar, -30.0, 20.0)
    return scale * (mean + torch.exp(0.5 * logvar) * noise)


def _vattn(W, a, h):
    """AttnBlock.forward model.py:176-202 (single head, scale C^-0.5)"""
    hn = _gn(W, a + ".norm", h, 1e-6)
    q = _conv(W, a + ".q", hn, padding=0); k = _conv(W, a + ".k", hn, padding=0); v = _conv(W, a + ".v", hn, padding=0)
    b, c, hh, ww = q.shape
    w_ = F.softmax(torch.bmm(q.reshape(b, c, hh * ww).permute(0, 2, 1), k.reshape(b, c, hh * ww)) * (int(c) ** (-0.5)), dim=2)
    h_ = torch.bmm(v.reshape(b, c, hh * ww), w_.permute(0, 2, 1)).reshape(b, c, hh, ww)
    return h + _conv(W, a + ".proj_out", h_, padding=0)


def vae_decode(W, cfg, z, prefix=""):
    """AutoencoderKL.decode autoencoder.py:330-333 (post_quant_conv) + Decoder.forward model.py:535-568. z NCHW, unscaled."""
    d = prefix + "decoder."
    h = _conv(W, prefix + "post_quant_conv", z, padding=0)
    h = _conv(W, d + "conv_in", h)
    h = _vres(W, d + "mid.block_1", h)
    h = _vattn(W, d + "mid.attn_1", h)
    h = _vres(W, d + "mid.block_2", h)
    for l in reversed(range(len(cfg["ch_mult"]))):
        for b in range(cfg["num_res_blocks"] + 1):
            h = _vres(W, f"{d}up.{l}.block.{b}", h)
        if l != 0:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")                            # Upsample model.py:52-56
            h = _conv(W, f"{d}up.{l}.upsample.conv", h)
    return _conv(W, d + "conv_out", F.silu(_gn(W, d + "norm_out", h, 1e-6)))


def decode_first_stage(W, cfg, z, scale_factor=0.18215, prefix=""):
    """ddpm.py:761-819: z = 1/scale_factor * z -> decode"""
    return vae_decode(W, cfg, 1.0 / scale_factor * z, prefix)


def ddim_schedule(alphas_cumprod, S, eta=0.0, T=1000, method="uniform"):
    """make_ddim_timesteps util.py:46-60 + make_ddim_sampling_parameters util.py:63-74 + ddim.py:51 — fp32 tables as the
    reference's p_sample_ddim sees them through torch.full (ddim.py:186-189)"""
    if method == "uniform":
        steps = np.asarray(list(range(0, T, T // S))) + 1
    else:
        steps = ((np.linspace(0, np.sqrt(T * .8), S)) ** 2).astype(int) + 1
    ac = alphas_cumprod.float()
    a = ac[steps]
    a_prev = np.asarray([float(ac[0])] + ac[steps[:-1]].tolist())
    # `ndarray / fp32-tensor` dispatches to Tensor.__rtruediv__ = reciprocal() * other: the reciprocal is rounded to fp32 first
    recip32 = (np.float32(1.0) / (1 - a.numpy())).astype(np.float32)
    sig = eta * np.sqrt((1 - a_prev) * recip32 * (1 - a.numpy() / a_prev))
    return dict(timesteps=steps, alphas=a.clone(), alphas_prev=torch.tensor(a_prev, dtype=torch.float32),
                sigmas=torch.tensor(sig, dtype=torch.float32), sqrt_one_minus_alphas=torch.sqrt(1. - a))


def ddim_step(x, e_t, a_t, a_prev, sigma_t, sqrt_one_minus_at, noise=None):
    """ddim.py:186-199 on fp32 0-dim tensors"""
    pred_x0 = (x - sqrt_one_minus_at * e_t) / a_t.sqrt()
    dir_xt = (1. - a_prev - sigma_t ** 2).sqrt() * e_t
    x_prev = a_prev.sqrt() * pred_x0 + dir_xt + (sigma_t * noise if noise is not None else 0.)
    return x_prev, pred_x0


def ddim_sample(W, cfg, x_T, cond, uncond, sched, scale=1.0, noise=None, prefix=""):
    """DDIMSampler.ddim_sampling + p_sample_ddim (ddim.py:113-204) over unet_forward, eps-parameterisation.
    noise: [S, *x.shape] in iteration order when sigmas != 0."""
    x = x_T
    S = len(sched["timesteps"])
    pred = None
    for i in range(S):
        index = S - 1 - i
        t = torch.full((x.shape[0],), int(sched["timesteps"][index]), dtype=torch.long)
        if uncond is None or scale == 1.0:
            e = unet_forward(W, cfg, x, t, cond, prefix)
        else:
            e_u, e_c = unet_forward(W, cfg, torch.cat([x] * 2), torch.cat([t] * 2), torch.cat([uncond, cond]), prefix).chunk(2)
            e = e_u + scale * (e_c - e_u)
        nz = noise[i] if (noise is not None and float(sched["sigmas"][index]) != 0.0) else None
        x, pred = ddim_step(x, e, sched["alphas"][index], sched["alphas_prev"][index], sched["sigmas"][index],
                            sched["sqrt_one_minus_alphas"][index], nz)
    return x, pred


# ----------------------------------------------------------------------------- CLIP text transformer
def clip_encode(W, cfg, embedded, prefix=""):
    """modules.py:295-296 (add position embedding), 384-402 (causal mask, encoder, final LN) with the layer arithmetic of
    transformers==4.18.0 modeling_clip: CLIPAttention (q scaled by d^-0.5, additive finfo.min causal mask, softmax),
    CLIPMLP (quick_gelu), pre-LN CLIPEncoderLayer."""
    w, H = cfg["width"], cfg["heads"]
    B, L, _ = embedded.shape
    x = embedded + W[prefix + "embeddings.position_embedding.weight"][:L]
    mask = torch.full((L, L), torch.finfo(torch.float32).min).triu_(1)                        # modules.py:24-31
    d = w // H
    for i in range(cfg["layers"]):
        p = f"{prefix}encoder.layers.{i}"
        r = x
        h = F.layer_norm(x, (w,), W[p + ".layer_norm1.weight"], W[p + ".layer_norm1.bias"], 1e-5)
        q = _lin(W, p + ".self_attn.q_proj", h) * (d ** -0.5)
        k = _lin(W, p + ".self_attn.k_proj", h); v = _lin(W, p + ".self_attn.v_proj", h)
        sp = lambda t: t.reshape(B, L, H, d).permute(0, 2, 1, 3)
        s = torch.matmul(sp(q), sp(k).transpose(-1, -2)) + mask
        a = torch.matmul(s.softmax(dim=-1), sp(v)).permute(0, 2, 1, 3).reshape(B, L, w)
        x = r + _lin(W, p + ".self_attn.out_proj", a)
        r = x
        h = F.layer_norm(x, (w,), W[p + ".layer_norm2.weight"], W[p + ".layer_norm2.bias"], 1e-5)
        h = _lin(W, p + ".mlp.fc1", h)
        h = h * torch.sigmoid(1.702 * h)
        x = r + _lin(W, p + ".mlp.fc2", h)
    return F.layer_norm(x, (w,), W[prefix + "final_layer_norm.weight"], W[prefix + "final_layer_norm.bias"], 1e-5)


# ----------------------------------------------------------------------------- iresnet (iresnet.py), eval mode
def _bn(W, p, x):
    return F.batch_norm(x, W[p + ".running_mean"], W[p + ".running_var"], W[p + ".weight"], W[p + ".bias"], False, 0.0, 1e-5)


def frnet_forward(W, cfg, x, prefix=""):
    """IResNet.forward iresnet.py:166-181, IBasicBlock.forward 52-63. x NCHW [N,3,S,S]"""
    x = F.conv2d(x, W[prefix + "conv1.weight"], None, 1, 1)
    x = F.prelu(_bn(W, prefix + "bn1", x), W[prefix + "prelu.weight"])
    for l in range(4):
        for i in range(cfg["layers"][l]):
            p = f"{prefix}layer{l + 1}.{i}"
            stride = 2 if i == 0 else 1
            out = _bn(W, p + ".bn1", x)
            out = F.conv2d(out, W[p + ".conv1.weight"], None, 1, 1)
            out = F.prelu(_bn(W, p + ".bn2", out), W[p + ".prelu.weight"])
            out = F.conv2d(out, W[p + ".conv2.weight"], None, stride, 1)
            out = _bn(W, p + ".bn3", out)
            idn = x
            if i == 0:
                idn = _bn(W, p + ".downsample.1", F.conv2d(x, W[p + ".downsample.0.weight"], None, stride, 0))
            x = out + idn
    x = _bn(W, prefix + "bn2", x)
    x = torch.flatten(x, 1)
    x = F.linear(x, W[prefix + "fc.weight"], W[prefix + "fc.bias"])
    return F.batch_norm(x, W[prefix + "features.running_mean"], W[prefix + "features.running_var"], W[prefix + "features.weight"],
                        W[prefix + "features.bias"], False, 0.0, 1e-5)


def face_warp_resize(faces_hwc, out_size=112):
    """meta_net.py:253-260: affine_grid/grid_sample (align_corners=True, zeros) then bilinear resize (align_corners=True).
    faces_hwc [N,S,S,3] -> (warped [N,3,S,S], small [N,3,out,out])"""
    img = faces_hwc.permute(0, 3, 1, 2)
    M = torch.tensor([TRANS_MATRIX], dtype=torch.float32).repeat(img.shape[0], 1, 1)
    grid = F.affine_grid(M, size=list(img.shape), align_corners=True)
    warped = F.grid_sample(img, grid, align_corners=True, mode="bilinear", padding_mode="zeros")
    small = F.interpolate(warped, size=out_size, mode="bilinear", align_corners=True)
    return warped, small


# ----------------------------------------------------------------------------- CelebBasis head + manager hook
def color_jitter_flip(img_hwc, flip, brightness, contrast, saturation, hue, order=(0, 1, 2, 3)):
    """face_id.py:526-532 `trans` after the resize on one HWC image in [-1,1]: RandomHorizontalFlip + ColorJitter + Normalize.
    `order` is ColorJitter's `fn_idx` (torchvision 0.11.3 transforms.ColorJitter.forward applies 0 = brightness, 1 = contrast,
    2 = saturation, 3 = hue in a random permutation drawn per call). The reference hands the transform a PIL image, so the
    arithmetic it executes is PIL's (ImageEnhance.Brightness / Contrast / Color = blend against black / the mean-gray image /
    the gray image; hue = a shift of the H plane of the 8-bit HSV image), quantised to uint8 after every op. This restatement
    keeps the same formulas in float (torchvision's functional_tensor forms) without the intermediate quantisation:
    tests/test_host_cpu.py::test_oracle_jitter_vs_pil pins it against PIL to that quantisation."""
    x = (img_hwc + 1.0) * 0.5
    if flip:
        x = x.flip(1)
    gray = lambda t: 0.2989 * t[..., 0] + 0.587 * t[..., 1] + 0.114 * t[..., 2]
    for op in order:
        if op == 0:
            x = (x * brightness).clamp(0, 1)
        elif op == 1:
            x = (contrast * x + (1.0 - contrast) * gray(x).mean()).clamp(0, 1)
        elif op == 2:
            x = (saturation * x + (1.0 - saturation) * gray(x).unsqueeze(-1)).clamp(0, 1)
        elif hue != 0.0:
            x = _hue_shift(x, hue)
    return x * 2.0 - 1.0


def _hue_shift(x, hue):
    """rgb -> hsv, h += hue (mod 1), hsv -> rgb on an HWC image in [0,1]"""
    r, g, b = x[..., 0], x[..., 1], x[..., 2]
    mx, mn = x.max(dim=-1).values, x.min(dim=-1).values
    d = mx - mn
    sat = torch.where(mx > 0, d / mx.clamp_min(1e-30), torch.zeros_like(mx))
    dd = torch.where(d > 0, d, torch.ones_like(d))
    h = torch.where(mx == r, (g - b) / dd, torch.where(mx == g, 2.0 + (b - r) / dd, 4.0 + (r - g) / dd))
    h = torch.where(d > 0, (h / 6.0) % 1.0, torch.zeros_like(h))
    h = (h + hue) % 1.0
    h6 = h * 6.0
    fi = torch.floor(h6); f = h6 - fi; i = fi.to(torch.int64) % 6
    v = mx
    p_ = (v * (1.0 - sat)).clamp(0, 1); q = (v * (1.0 - f * sat)).clamp(0, 1); t = (v * (1.0 - (1.0 - f) * sat)).clamp(0, 1)
    sel = lambda a0, a1, a2, a3, a4, a5: torch.stack([a0, a1, a2, a3, a4, a5], -1).gather(-1, i.unsqueeze(-1)).squeeze(-1)
    return torch.stack([sel(v, q, p_, p_, t, v), sel(t, v, v, q, p_, p_), sel(p_, p_, t, v, v, q)], dim=-1)


def add_bg(img_hwc, rh, rw, pos_h, pos_w):
    """face_id.py:712-730 `_add_bg` with the random draws made explicit"""
    S = img_hwc.shape[0]
    small = F.interpolate(img_hwc.permute(2, 0, 1)[None], (rh, rw), mode="bilinear", align_corners=True)[0].permute(1, 2, 0)
    out = torch.full_like(img_hwc, -1.0)
    out[pos_h:pos_h + rh, pos_w:pos_w + rw] = small
    return out


def shift_rows(tokens_1d, embedded_2d, z_rows, placeholder, reps):
    """helpers.py:13-41 shift_tensor_dim0 + embedding_manager.py:347-360 (num_ids == 1 branch), restated with explicit index
    arithmetic: rows after each placeholder move right by reps-1, the tail is dropped, rows p..p+reps-1 take z."""
    T = tokens_1d.shape[0]
    toks = tokens_1d.tolist()
    pos = [i for i, tk in enumerate(toks) if tk == placeholder]
    offset = np.zeros(T, dtype=np.int64)
    for p in pos:
        offset[p + 1:] += reps - 1
    keep = T - len(pos) * (reps - 1)
    target = (np.arange(T) + offset)[:keep]
    out = embedded_2d.clone()
    out[torch.as_tensor(target)] = embedded_2d[:keep]
    for p in pos:
        for j in range(reps):
            if target[p] + j < T:
                out[int(target[p]) + j] = z_rows[j]
    return out


def shift_rows_multi(tokens_1d, embedded_2d, z_list, placeholders, reps):
    """helpers.py:6-41 with several placeholder tokens + embedding_manager.py:323-345 / 363-391 (two / three persons):
    every occurrence of placeholders[k] is widened to `reps` rows that take z_list[k]; later rows move right, the tail is dropped."""
    T = tokens_1d.shape[0]
    toks = tokens_1d.tolist()
    pos = [(i, k) for k, ph in enumerate(placeholders) for i, tk in enumerate(toks) if tk == ph]
    offset = np.zeros(T, dtype=np.int64)
    for p, _ in pos:
        offset[p + 1:] += reps - 1
    keep = T - len(pos) * (reps - 1)
    target = (np.arange(T) + offset)[:keep]
    out = embedded_2d.clone()
    out[torch.as_tensor(target)] = embedded_2d[:keep]
    for p, k in pos:
        for j in range(reps):
            if p < keep and target[p] + j < T:
                out[int(target[p]) + j] = z_list[k][j]
    return out


def celeb_head(W, v, es, inner):
    """meta_net.py:266-289: EqualLinear(lr_mul=1)+LeakyReLU(0.2) -> (b,e,h=1,d) -> L2 norm over d -> einsum with basis + mean.
    v: [R,512] normalised FR features. returns (coef [R,es,inner], z [R,es,width])"""
    x = F.leaky_relu(F.linear(v, W["embedding_manager.meta_id_net.stylegan_mlp.net.0.weight"] * 1.0,
                              W["embedding_manager.meta_id_net.stylegan_mlp.net.0.bias"] * 1.0), 0.2)
    x = x.reshape(v.shape[0], es, 1, inner)
    x = F.normalize(x, dim=-1, p=2)
    c_all = W["celeb_embeddings"]
    c_mean, base = c_all[:, 0], c_all[:, 1:]
    z = torch.einsum("behk,ekc->behc", x, base) + c_mean[None, :, None, :]
    return x.reshape(v.shape[0], es, inner), z.reshape(v.shape[0], es, -1)


def make_schedule(linear_start=0.00085, linear_end=0.012, timesteps=1000):
    """ddpm.py:126-178 + util.py:21-25 ('linear' = sqrt-space linspace in float64)"""
    betas = (torch.linspace(linear_start ** 0.5, linear_end ** 0.5, timesteps, dtype=torch.float64) ** 2).numpy()
    ac = np.cumprod(1.0 - betas, axis=0)
    return dict(betas=torch.tensor(betas, dtype=torch.float32), alphas_cumprod=torch.tensor(ac, dtype=torch.float32),
                sqrt_alphas_cumprod=torch.tensor(np.sqrt(ac), dtype=torch.float32),
                sqrt_one_minus_alphas_cumprod=torch.tensor(np.sqrt(1.0 - ac), dtype=torch.float32))


def adamw_step(p, g, m, v, step, lr, b1=0.9, b2=0.999, eps=1e-8, wd=1e-2):
    """torch.optim.AdamW single-tensor update (ddpm.py:1454 uses torch defaults)"""
    p = p * (1 - lr * wd)
    m = m + (g - m) * (1 - b1)
    v = v * b2 + (1 - b2) * g * g
    bc1 = 1 - b1 ** step; bc2 = 1 - b2 ** step
    denom = v.sqrt() / math.sqrt(bc2) + eps
    p = p - (lr / bc1) * (m / denom)
    return p, m, v


class TrainState:
    """everything the step mutates: the EqualLinear parameters, Adam moments, step count, EMA dictionaries"""

    def __init__(self, W):
        self.Wm = W["embedding_manager.meta_id_net.stylegan_mlp.net.0.weight"].clone()
        self.bm = W["embedding_manager.meta_id_net.stylegan_mlp.net.0.bias"].clone()
        self.mW = torch.zeros_like(self.Wm); self.vW = torch.zeros_like(self.Wm)
        self.mb = torch.zeros_like(self.bm); self.vb = torch.zeros_like(self.bm)
        self.step = 0
        self.id_emb = W["id_embeddings"].clone() if "id_embeddings" in W else None
        self.id_coef = W["id_coefficients"].clone() if "id_coefficients" in W else None
        # use_rm_mlp ablation (meta_net.py:173-175): the coefficients themselves are the parameter, (max_ids, es, h=1, inner)
        self.coef = W["embedding_manager.meta_id_net.coef"].clone() if "embedding_manager.meta_id_net.coef" in W else None
        if self.coef is not None:
            self.mc = torch.zeros_like(self.coef); self.vc = torch.zeros_like(self.coef)


def train_step(W, st, batch, lr, unet=UNET_SD14, vae=VAE_SD14, clip=CLIP_L14, fr=FR_R100, es=2, inner=512, momentum=0.99,
               scale=0.18215, placeholder=PLACEHOLDER_SKS, sched=None, return_aux=False, use_rm_mlp=False):
    """One LatentDiffusion.training_step + backward + AdamW.step (SURVEY.md §3.1 hot loop):
    ddpm.py:357-370 -> 921-936 -> 702-759 (VAE encode) -> 599-611 (conditioning with the manager hook,
    embedding_manager.py:279-394) -> 1069-1116 (q_sample, UNet, l2 loss) -> autograd -> AdamW (1442-1454).
    batch: image [B,S,S,3], faces [B,S,S,6], ids [B,2], tokens [B,77], t [B], noise [B,4,h,w], post_noise [B,4,h,w].
    use_rm_mlp (meta_net.py:173-175, 272-273): x = coef[id_idx] replaces FR-net + MLP + L2-norm; AdamW runs on the whole
    (max_ids, es, 1, inner) parameter (rows of identities absent from the batch get a zero gradient, i.e. only weight decay)."""
    sched = sched or make_schedule()
    B = batch["tokens"].shape[0]
    W = dict(W)
    Wm = st.Wm.clone().requires_grad_(True); bm = st.bm.clone().requires_grad_(True)
    W["embedding_manager.meta_id_net.stylegan_mlp.net.0.weight"] = Wm
    W["embedding_manager.meta_id_net.stylegan_mlp.net.0.bias"] = bm
    with torch.no_grad():
        x = batch["image"].permute(0, 3, 1, 2)                                                 # ddpm.py:344-350
        mom = vae_encode_moments(W, vae, x, "first_stage_model.")
        z0 = vae_sample(mom, batch["post_noise"], scale)
        faces = torch.cat(batch["faces"].chunk(2, -1), 0)                                      # meta_net.py:336-337
        feat = small = None
        if not use_rm_mlp:
            _, small = face_warp_resize(faces, fr["in_size"])
            feat = frnet_forward(W, fr, small, "embedding_manager.meta_id_net.id_model.")
            v = F.normalize(feat, dim=-1, p=2)
    if use_rm_mlp:
        cp = st.coef.clone().requires_grad_(True)
        idx = torch.flatten(batch["ids"])                                                       # meta_net.py:338-340: flatten(id_multi), row-major
        x = cp[idx]                                                                             # (2B, es, 1, inner), NOT normalised
        c_all = W["celeb_embeddings"]
        z = (torch.einsum("behk,ekc->behc", x, c_all[:, 1:]) + c_all[:, 0][None, :, None, :]).reshape(2 * B, es, -1)
        coef = x.reshape(2 * B, es, inner)
    else:
        coef, z = celeb_head(W, v, es, inner)
    with torch.no_grad():                                                                       # embedding_manager.py:484-489
        for b in range(B):
            i = int(batch["ids"][b, 0])
            if st.id_emb is not None:
                st.id_emb[i] = momentum * st.id_emb[i] + (1 - momentum) * z[b].reshape(-1)
                st.id_coef[i] = momentum * st.id_coef[i] + (1 - momentum) * coef[b].reshape(-1)
    tok_emb = W["cond_stage_model.transformer.text_model.embeddings.token_embedding.weight"][batch["tokens"]]
    embedded = torch.stack([shift_rows(batch["tokens"][b], tok_emb[b], z[b], placeholder, es) for b in range(B)])
    cond = clip_encode(W, clip, embedded, "cond_stage_model.transformer.text_model.")
    t = batch["t"]
    sa = sched["sqrt_alphas_cumprod"][t].reshape(B, 1, 1, 1); sb = sched["sqrt_one_minus_alphas_cumprod"][t].reshape(B, 1, 1, 1)
    xt = sa * z0 + sb * batch["noise"]                                                          # ddpm.py:289-292
    eps_hat = unet_forward(W, unet, xt, t, cond, "model.diffusion_model.")
    loss = F.mse_loss(batch["noise"], eps_hat, reduction="none").mean([1, 2, 3]).mean()         # ddpm.py:1081-1093 (logvar == 0)
    st.step += 1
    if use_rm_mlp:
        (gc,) = torch.autograd.grad(loss, [cp])
        st.coef, st.mc, st.vc = adamw_step(st.coef, gc, st.mc, st.vc, st.step, lr)
        gW, gb = gc, None
    else:
        gW, gb = torch.autograd.grad(loss, [Wm, bm])
        st.Wm, st.mW, st.vW = adamw_step(st.Wm, gW, st.mW, st.vW, st.step, lr)
        st.bm, st.mb, st.vb = adamw_step(st.bm, gb, st.mb, st.vb, st.step, lr)
    if return_aux:
        return loss.detach(), dict(gW=gW, gb=gb, z0=z0, coef=coef.detach(), z=z.detach(), cond=cond.detach(),
                                   eps_hat=eps_hat.detach(), moments=mom, feat=feat, small=small)
    return loss.detach()
