"""Generate tests/golden/*.pt by running the REFERENCE's own modules (imported from /root/reference, read-only) on
seeded inputs. Runs only in the build container (the reference never travels to the GPU box); the resulting
fixtures are data (inputs + expected outputs), committed next to this script.

    python oracle/gen_golden.py            # writes tests/golden/*.pt and tests/golden/MANIFEST.json

Import shims (harness side only, no reference edits): omegaconf / pytorch_lightning / torchvision / taming / kornia /
clip are absent from this image; the reference touches them only at import time or in code paths the fixtures do not
exercise.
"""
import hashlib
import json
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("CELEBBASIS_REFERENCE", "/root/reference")
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)
sys.path.insert(0, REF)

from oracle import cb_oracle as O  # noqa: E402


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def install_shims():
    class ListConfig(list):
        pass
    _stub("omegaconf", ListConfig=ListConfig, OmegaConf=object)
    _stub("omegaconf.listconfig", ListConfig=ListConfig)

    class LightningModule(nn.Module):
        @property
        def device(self):
            return torch.device("cpu")
    pl = _stub("pytorch_lightning", LightningModule=LightningModule)
    _stub("pytorch_lightning.utilities")
    _stub("pytorch_lightning.utilities.distributed", rank_zero_only=lambda f: f)
    pl.utilities = sys.modules["pytorch_lightning.utilities"]
    tv = _stub("torchvision")
    _stub("torchvision.utils", make_grid=lambda *a, **k: None)
    tr = _stub("torchvision.transforms")
    for n in ("Compose", "Resize", "ToTensor", "Normalize", "RandomHorizontalFlip", "ColorJitter", "CenterCrop", "RandomCrop"):
        setattr(tr, n, lambda *a, **k: None)
    tr.functional = _stub("torchvision.transforms.functional")
    tv.transforms = tr
    tv.utils = sys.modules["torchvision.utils"]
    _stub("taming"); _stub("taming.modules"); _stub("taming.modules.vqvae")
    _stub("taming.modules.vqvae.quantize", VectorQuantizer2=object)
    _stub("kornia")
    _stub("clip")


def sha(obj):
    return hashlib.sha256(obj.encode()).hexdigest()


def keys_digest(shapes):
    lines = sorted(f"{k}:{tuple(int(x) for x in v)}" for k, v in shapes.items())
    return dict(n=len(lines), params=int(sum(int(np.prod(v)) for v in shapes.values())), sha256=sha("\n".join(lines)))


def sd_shapes(module, prefix="", skip=()):
    return {prefix + k: tuple(v.shape) for k, v in module.state_dict().items()
            if v.dtype.is_floating_point and not any(s in k for s in skip)}


def load_into(module, W, prefix=""):
    sd = module.state_dict()
    new = {}
    for k, v in sd.items():
        if (prefix + k) in W:
            assert tuple(W[prefix + k].shape) == tuple(v.shape), (k, W[prefix + k].shape, v.shape)
            new[k] = W[prefix + k].clone()
        else:
            assert not v.dtype.is_floating_point or "num_batches" in k, f"oracle table lacks {prefix + k}"
            new[k] = v
    module.load_state_dict(new, strict=True)


MINI_UNET = dict(in_channels=4, out_channels=4, model_channels=32, num_res_blocks=1, channel_mult=(1, 2, 2),
                 attention_resolutions=(1, 2), num_heads=2, context_dim=48)
MINI_VAE = dict(ch=32, ch_mult=(1, 2, 2), num_res_blocks=1, in_channels=3, z_channels=4)
MINI_FR = dict(layers=(1, 1, 1, 1), width=64, in_size=112, feat=512)
MINI_CLIP = dict(vocab=120, width=64, layers=2, heads=4, ctx_len=77, mlp=128)


def gen_schedule(manifest):
    from ldm.modules.diffusionmodules.util import make_beta_schedule
    betas = make_beta_schedule("linear", 1000, linear_start=0.00085, linear_end=0.012)   # util.py:21-25
    alphas_cumprod = np.cumprod(1.0 - betas, axis=0)                                         # ddpm.py:133-134
    fx = dict(betas=torch.tensor(betas, dtype=torch.float32), alphas_cumprod=torch.tensor(alphas_cumprod, dtype=torch.float32),
              sqrt_alphas_cumprod=torch.tensor(np.sqrt(alphas_cumprod), dtype=torch.float32),
              sqrt_one_minus_alphas_cumprod=torch.tensor(np.sqrt(1.0 - alphas_cumprod), dtype=torch.float32))
    # q_sample through the reference's own extract_into_tensor (util.py:96-99, ddpm.py:289-292)
    from ldm.modules.diffusionmodules.util import extract_into_tensor
    g = torch.Generator().manual_seed(11)
    x0 = torch.randn(3, 4, 8, 8, generator=g); eps = torch.randn(3, 4, 8, 8, generator=g); t = torch.tensor([0, 500, 999])
    fx["q_x0"], fx["q_eps"], fx["q_t"] = x0, eps, t
    fx["q_xt"] = (extract_into_tensor(fx["sqrt_alphas_cumprod"], t, x0.shape) * x0 +
                  extract_into_tensor(fx["sqrt_one_minus_alphas_cumprod"], t, x0.shape) * eps)
    from ldm.modules.diffusionmodules.util import timestep_embedding
    fx["temb_t"] = torch.tensor([0, 1, 37, 999]); fx["temb"] = timestep_embedding(fx["temb_t"], 320)
    torch.save(fx, os.path.join(OUT, "schedule.pt"))
    manifest["schedule.pt"] = "ldm/modules/diffusionmodules/util.py make_beta_schedule, extract_into_tensor, timestep_embedding; ddpm.py:126-178,289-292"


def gen_shift_rows(manifest):
    from ldm.modules.id_embedding.helpers import get_rep_pos, shift_tensor_dim0
    cases = []
    g = torch.Generator().manual_seed(3)
    specs = [
        (torch.LongTensor([0, 1, 2, 2, 3, 4, 5, 6, 7, 99] + [99] * 20), 2, 2),          # helpers.py:44-54 own example (token 2)
        (torch.LongTensor([49406, 320, 1125, 539, 320, 1710, 539, 48136, 2533, 49407] + [49407] * 67), 48136, 2),
        (torch.LongTensor([49406, 48136] + [49407] * 75), 48136, 2),
        (torch.LongTensor([49406] + [5] * 74 + [48136, 49407]), 48136, 2),                # placeholder near the end (tail dropped)
        (torch.LongTensor([49406, 48136, 7, 48136, 9] + [49407] * 72), 48136, 3),         # two occurrences, reps 3
        (torch.LongTensor([49406, 1, 2, 3] + [49407] * 73), 48136, 2),                    # no placeholder
    ]
    for toks, ph, reps in specs:
        emb = torch.randn(toks.shape[0], 6, generator=g)
        z = torch.randn(reps, 6, generator=g)
        pos = get_rep_pos(toks, [ph])
        out, final = shift_tensor_dim0(emb.clone(), pos, reps)
        for one_pos in final[0]:                      # embedding_manager.py:359-360
            out[one_pos] = z
        cases.append(dict(tokens=toks, placeholder=ph, reps=reps, embedded=emb, z=z, out=out))
    torch.save(cases, os.path.join(OUT, "shift_rows.pt"))
    manifest["shift_rows.pt"] = "ldm/modules/id_embedding/helpers.py:6-41 + embedding_manager.py:347-360"
    # several placeholder tokens in one prompt (two / three persons, embedding_manager.py:323-345, 363-391)
    multi = []
    mspecs = [
        (torch.LongTensor([0, 1, 2, 2, 3, 4, 5, 6, 7, 99] + [99] * 20), [2, 6], 2),                 # helpers.py:44-54, its own rep_tokens
        (torch.LongTensor([49406, 320, 1125, 539, 48136, 2533, 537, 662, 2533, 49407] + [49407] * 67), [48136, 662], 2),   # sks ... ks
        (torch.LongTensor([49406, 662, 267, 48136, 537, 4236, 49407] + [49407] * 70), [48136, 662, 4236], 2),               # ks, sks and ata
        (torch.LongTensor([49406, 320, 1125, 539, 48136, 2533, 49407] + [49407] * 70), [48136, 662], 2),                    # second token absent
    ]
    for toks, phs, reps in mspecs:
        emb = torch.randn(toks.shape[0], 6, generator=g)
        zs = [torch.randn(reps, 6, generator=g) for _ in phs]
        out, final = shift_tensor_dim0(emb.clone(), get_rep_pos(toks, phs), reps)
        for k in range(len(phs)):
            for one_pos in final[k]:
                out[one_pos] = zs[k]
        multi.append(dict(tokens=toks, placeholders=phs, reps=reps, embedded=emb, z=zs, out=out))
    torch.save(multi, os.path.join(OUT, "shift_rows_multi.pt"))
    manifest["shift_rows_multi.pt"] = "ldm/modules/id_embedding/helpers.py:6-41 + embedding_manager.py:323-345, 363-391 (several placeholders)"


def gen_unet(manifest):
    from ldm.modules.diffusionmodules.openaimodel import UNetModel
    cfg = MINI_UNET
    torch.manual_seed(0)
    m = UNetModel(image_size=32, in_channels=cfg["in_channels"], model_channels=cfg["model_channels"], out_channels=cfg["out_channels"],
                  num_res_blocks=cfg["num_res_blocks"], attention_resolutions=list(cfg["attention_resolutions"]),
                  channel_mult=list(cfg["channel_mult"]), num_heads=cfg["num_heads"], use_spatial_transformer=True,
                  transformer_depth=1, context_dim=cfg["context_dim"], use_checkpoint=True, legacy=False).eval()
    shapes = O.unet_param_shapes(cfg)
    assert keys_digest(shapes) == keys_digest(sd_shapes(m)), "mini UNet key table mismatch"
    W = O.init_weights(shapes, seed=7)
    load_into(m, W)
    g = torch.Generator().manual_seed(21)
    x = torch.randn(2, 4, 16, 16, generator=g); t = torch.tensor([3, 721]); ctx = torch.randn(2, 77, cfg["context_dim"], generator=g)
    dout = torch.randn(2, 4, 16, 16, generator=g)
    ctx_r = ctx.clone().requires_grad_(True)
    out = m(x, t, context=ctx_r)
    (dctx,) = torch.autograd.grad((out * dout).sum(), [ctx_r])
    torch.save(dict(cfg=cfg, weight_seed=7, x=x, t=t, context=ctx, dout=dout, out=out.detach(), dcontext=dctx),
               os.path.join(OUT, "unet_mini.pt"))
    manifest["unet_mini.pt"] = "ldm/modules/diffusionmodules/openaimodel.py UNetModel.forward:710-742 + autograd to context"
    # full SD v1-4 key table (aigc_id.yaml:39-54): built on the meta device, no memory
    with torch.device("meta"):
        full = UNetModel(image_size=32, in_channels=4, model_channels=320, out_channels=4, num_res_blocks=2,
                         attention_resolutions=[4, 2, 1], channel_mult=[1, 2, 4, 4], num_heads=8, use_spatial_transformer=True,
                         transformer_depth=1, context_dim=768, use_checkpoint=True, legacy=False)
    return keys_digest(sd_shapes(full))


def gen_vae(manifest):
    from ldm.modules.diffusionmodules.model import Encoder
    from ldm.modules.distributions.distributions import DiagonalGaussianDistribution
    cfg = MINI_VAE
    enc = Encoder(ch=cfg["ch"], out_ch=3, ch_mult=cfg["ch_mult"], num_res_blocks=cfg["num_res_blocks"], attn_resolutions=[],
                  dropout=0.0, in_channels=3, resolution=64, z_channels=cfg["z_channels"], double_z=True).eval()
    quant = nn.Conv2d(2 * cfg["z_channels"], 2 * cfg["z_channels"], 1)                           # autoencoder.py:302
    shapes = O.vae_encoder_param_shapes(cfg)
    ref_shapes = sd_shapes(enc, "encoder."); ref_shapes.update(sd_shapes(quant, "quant_conv."))
    assert keys_digest(shapes) == keys_digest(ref_shapes), "mini VAE key table mismatch"
    W = O.init_weights(shapes, seed=9)
    load_into(enc, W, "encoder."); load_into(quant, W, "quant_conv.")
    g = torch.Generator().manual_seed(31)
    x = torch.rand(2, 3, 64, 64, generator=g) * 2 - 1
    with torch.no_grad():
        moments = quant(enc(x))
        post = DiagonalGaussianDistribution(moments)
        torch.manual_seed(77)
        noise_same_stream = None
        z = post.sample()                                                                        # distributions.py:35-37
        torch.manual_seed(77)
        noise_same_stream = torch.randn(post.mean.shape)
    torch.save(dict(cfg=cfg, weight_seed=9, x=x, moments=moments, noise=noise_same_stream, z_scaled=0.18215 * z),
               os.path.join(OUT, "vae_mini.pt"))
    manifest["vae_mini.pt"] = "ldm/modules/diffusionmodules/model.py Encoder.forward:434-459, distributions.py:24-37, ddpm.py:597"
    with torch.device("meta"):
        full = Encoder(ch=128, out_ch=3, ch_mult=(1, 2, 4, 4), num_res_blocks=2, attn_resolutions=[], dropout=0.0, in_channels=3,
                       resolution=512, z_channels=4, double_z=True)
        fq = nn.Conv2d(8, 8, 1)
    fs = sd_shapes(full, "encoder."); fs.update(sd_shapes(fq, "quant_conv."))
    return keys_digest(fs)


def gen_vae_decoder(manifest):
    """reference Decoder + post_quant_conv on a seeded latent (sampling path)"""
    from ldm.modules.diffusionmodules.model import Decoder
    cfg = MINI_VAE
    dec = Decoder(ch=cfg["ch"], out_ch=3, ch_mult=cfg["ch_mult"], num_res_blocks=cfg["num_res_blocks"], attn_resolutions=[],
                  dropout=0.0, in_channels=3, resolution=64, z_channels=cfg["z_channels"]).eval()
    pq = nn.Conv2d(cfg["z_channels"], cfg["z_channels"], 1)                                       # autoencoder.py:303
    shapes = O.vae_decoder_param_shapes(cfg)
    ref_shapes = sd_shapes(dec, "decoder."); ref_shapes.update(sd_shapes(pq, "post_quant_conv."))
    assert keys_digest(shapes) == keys_digest(ref_shapes), "mini VAE decoder key table mismatch"
    W = O.init_weights(shapes, seed=19)
    load_into(dec, W, "decoder."); load_into(pq, W, "post_quant_conv.")
    g = torch.Generator().manual_seed(41)
    f = 2 ** (len(cfg["ch_mult"]) - 1)
    z = torch.randn(2, cfg["z_channels"], 64 // f, 64 // f, generator=g) * 0.18215 * 4.0
    with torch.no_grad():
        img = dec(pq(1.0 / 0.18215 * z))                                                         # ddpm.py:772 + autoencoder.py:330-333
        img01 = torch.clamp((img + 1.0) / 2.0, min=0.0, max=1.0)                                  # stable_txt2img.py:343
    torch.save(dict(cfg=cfg, weight_seed=19, z=z, scale_factor=0.18215, image=img, image01=img01), os.path.join(OUT, "vae_dec_mini.pt"))
    manifest["vae_dec_mini.pt"] = ("ldm/modules/diffusionmodules/model.py Decoder.forward:535-568, autoencoder.py:330-333, ddpm.py:772, "
                                   "scripts/stable_txt2img.py:343")
    with torch.device("meta"):
        full = Decoder(ch=128, out_ch=3, ch_mult=(1, 2, 4, 4), num_res_blocks=2, attn_resolutions=[], dropout=0.0, in_channels=3,
                       resolution=512, z_channels=4)
        fq = nn.Conv2d(4, 4, 1)
    fs = sd_shapes(full, "decoder."); fs.update(sd_shapes(fq, "post_quant_conv."))
    return keys_digest(fs)


def gen_ddim(manifest):
    """reference DDIMSampler (ddim.py) driving the reference mini UNetModel: schedules, one p_sample_ddim update, and short
    guided sampling runs at eta 0 and eta 0.5"""
    from ldm.models.diffusion.ddim import DDIMSampler
    from ldm.modules.diffusionmodules.openaimodel import UNetModel
    from ldm.modules.diffusionmodules.util import make_ddim_timesteps, make_ddim_sampling_parameters
    cfg = MINI_UNET
    unet = UNetModel(image_size=32, in_channels=cfg["in_channels"], model_channels=cfg["model_channels"], out_channels=cfg["out_channels"],
                     num_res_blocks=cfg["num_res_blocks"], attention_resolutions=list(cfg["attention_resolutions"]),
                     channel_mult=list(cfg["channel_mult"]), num_heads=cfg["num_heads"], use_spatial_transformer=True,
                     transformer_depth=1, context_dim=cfg["context_dim"], use_checkpoint=True, legacy=False).eval()
    W = O.init_weights(O.unet_param_shapes(cfg), seed=7)
    load_into(unet, W)
    sch = O.make_schedule()

    class Model:                      # the surface DDIMSampler touches on LatentDiffusion (ddim.py:16,28-34,120,172-179)
        num_timesteps = 1000
        parameterization = "eps"
        device = torch.device("cpu")
        betas = sch["betas"]
        alphas_cumprod = sch["alphas_cumprod"]
        alphas_cumprod_prev = torch.cat([torch.ones(1), sch["alphas_cumprod"][:-1]])

        def apply_model(self, x, t, c):
            return unet(x, t, context=c)

    DDIMSampler.register_buffer = lambda self, name, attr: setattr(self, name, attr)          # harness side: no .to("cuda") on a CPU box
    out = dict(cfg=cfg, weight_seed=7)
    # (a) schedule tables for the shipped setting (50 uniform steps) and the short ones used below
    for S, eta in ((50, 0.0), (5, 0.0), (4, 0.5)):
        ts = make_ddim_timesteps("uniform", S, 1000, verbose=False)
        sig, a, ap = make_ddim_sampling_parameters(sch["alphas_cumprod"], ts, eta, verbose=False)
        out[f"sched_{S}_{eta}"] = dict(timesteps=torch.tensor(ts), sigmas=torch.tensor(np.asarray(sig)), alphas=torch.as_tensor(a).clone(),
                                       alphas_prev=torch.tensor(np.asarray(ap)), sqrt_one_minus_alphas=torch.as_tensor(np.sqrt(1. - a)).clone())
    out["quad_10"] = torch.tensor(make_ddim_timesteps("quad", 10, 1000, verbose=False))
    g = torch.Generator().manual_seed(51)
    n = 2
    x_T = torch.randn(n, 4, 16, 16, generator=g)
    cond = torch.randn(n, 77, cfg["context_dim"], generator=g); uncond = torch.randn(n, 77, cfg["context_dim"], generator=g)
    out.update(x_T=x_T, cond=cond, uncond=uncond)
    sampler = DDIMSampler(Model())
    with torch.no_grad():
        # (b) a single guided update in the middle of a 5-step schedule
        sampler.make_schedule(5, ddim_eta=0.0, verbose=False)
        ts = torch.full((n,), int(sampler.ddim_timesteps[2]), dtype=torch.long)
        x1, p0 = sampler.p_sample_ddim(x_T, cond, ts, index=2, unconditional_guidance_scale=7.5, unconditional_conditioning=uncond)
        out["step_index2"] = dict(x_prev=x1, pred_x0=p0)
        # (c) full guided runs
        s0, inter0 = sampler.sample(S=5, batch_size=n, shape=[4, 16, 16], conditioning=cond, verbose=False, unconditional_guidance_scale=7.5,
                                    unconditional_conditioning=uncond, eta=0.0, x_T=x_T)
        out["run_5_eta0_cfg7.5"] = dict(samples=s0, pred_x0=inter0["pred_x0"][-1])
        s1, _ = sampler.sample(S=5, batch_size=n, shape=[4, 16, 16], conditioning=cond, verbose=False, unconditional_guidance_scale=1.0,
                               unconditional_conditioning=None, eta=0.0, x_T=x_T)
        out["run_5_eta0_nocfg"] = dict(samples=s1)
        torch.manual_seed(5)
        s2, _ = sampler.sample(S=4, batch_size=n, shape=[4, 16, 16], conditioning=cond, verbose=False, unconditional_guidance_scale=3.0,
                               unconditional_conditioning=uncond, eta=0.5, x_T=x_T)
        torch.manual_seed(5)
        noise = torch.stack([torch.randn(n, 4, 16, 16) for _ in range(4)])       # the stream noise_like drew, ddim.py:194
        out["run_4_eta0.5_cfg3"] = dict(samples=s2, noise=noise)
    torch.save(out, os.path.join(OUT, "ddim_mini.pt"))
    manifest["ddim_mini.pt"] = ("ldm/models/diffusion/ddim.py DDIMSampler.make_schedule:25-54, sample:56-111, ddim_sampling:113-163, "
                                "p_sample_ddim:166-204; util.py make_ddim_timesteps:46-60, make_ddim_sampling_parameters:63-74 — over the "
                                "reference UNetModel (mini config)")


def gen_templates():
    """sha256 + length of the caption template list (ldm/data/face_id.py:17-99), read without importing torchvision"""
    import ast
    tree = ast.parse(open(os.path.join(REF, "ldm", "data", "face_id.py")).read())
    for node in tree.body:
        if isinstance(node, ast.Assign) and getattr(node.targets[0], "id", "") == "imagenet_templates_small":
            lst = ast.literal_eval(node.value)
            return dict(sha256=hashlib.sha256("\n".join(lst).encode()).hexdigest(), count=len(lst))
    raise RuntimeError("imagenet_templates_small not found")


def gen_config_params(manifest):
    """the PARAMETER VALUES of the reference's personalization config (configs/stable-diffusion/aigc_id.yaml), parsed and re-emitted
    as canonical JSON: data the drop-in tests instantiate the host mirror from (the reference YAML itself cannot travel)"""
    import yaml
    cfg = yaml.safe_load(open(os.path.join(REF, "configs", "stable-diffusion", "aigc_id.yaml")))
    with open(os.path.join(OUT, "aigc_id_params.json"), "w") as f:
        json.dump(cfg, f, separators=(",", ":"), sort_keys=True)
    manifest["aigc_id_params.json"] = "yaml.safe_load(configs/stable-diffusion/aigc_id.yaml) re-emitted as sorted JSON (target strings + parameter values)"


def gen_basis(manifest):
    """the reference's own FrozenCLIPEmbedder._get_celeb_embeddings (modules.py:472-624) run unbound on a stand-in `self`:
    real name list (infer_images/wiki_names_v2.txt) with the real BPE ids the reference recorded (infer_images/token_len.txt),
    a seeded token-embedding table at reduced width (no CLIP weights exist offline)."""
    import re
    import tempfile
    from ldm.modules.encoders.modules import FrozenCLIPEmbedder
    tab = {}
    for line in open(os.path.join(REF, "infer_images", "token_len.txt")):
        m = re.match(r"\d+ (.*): len=(\d+), token=\[(.*)\]", line.strip())
        if m:
            tab[m.group(1)] = [int(x) for x in m.group(3).split(",") if x.strip()]
    names = [n for n in open(os.path.join(REF, "infer_images", "wiki_names_v2.txt")).read().splitlines() if n in tab]   # keeps duplicates
    width, K = 96, 64
    table = torch.randn(49408, width, generator=torch.Generator().manual_seed(61)) * 0.02

    def tokenizer(name, **kw):
        ids = [49406] + tab[name] + [49407] * (77 - 1 - len(tab[name]))
        return {"input_ids": torch.tensor([ids])}

    out = dict(table_seed=61, width=width, n_components=K, names=names, token_table={n: tab[n] for n in sorted(set(names))})
    with tempfile.TemporaryDirectory() as td:
        txt = os.path.join(td, "names.txt")
        open(txt, "w").write("\n".join(names))
        for flatten in (False, True):
            me = types.SimpleNamespace(celeb_txt=txt, rm_repeats=True, tokenizer=tokenizer, max_length=77, dev=False,
                                       transformer=lambda input_ids, only_embedding: table[input_ids], use_flatten=flatten,
                                       num_embeds_per_token=2, use_sample_reduce=False, n_samples=513, use_svd=True, device="cpu")
            FrozenCLIPEmbedder._get_celeb_embeddings(me, n_components=K)
            out["basis_flatten" if flatten else "basis"] = me.celeb_embeddings.clone()
    torch.save(out, os.path.join(OUT, "basis_mini.pt"))
    manifest["basis_mini.pt"] = ("ldm/modules/encoders/modules.py FrozenCLIPEmbedder._get_celeb_embeddings:472-619 (use_svd, rm_repeats, "
                                 "flatten on/off) on infer_images/wiki_names_v2.txt with the BPE ids of infer_images/token_len.txt")


def _fr_reinit(net, seed):
    """the reference initialises convs with N(0, 0.1) which overflows at depth with random BN stats; re-seed small"""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for k, v in net.state_dict().items():
            if not v.dtype.is_floating_point:
                continue
            if k.endswith("running_var"):
                v.copy_(0.5 + torch.rand(v.shape, generator=g))
            elif k.endswith("running_mean"):
                v.copy_(0.1 * torch.randn(v.shape, generator=g))
            elif v.dim() > 1:
                fan = int(np.prod(v.shape[1:]))
                v.copy_(torch.randn(v.shape, generator=g) / np.sqrt(fan))
            elif "prelu" in k:
                v.copy_(0.25 + 0.05 * torch.randn(v.shape, generator=g))
            elif k.endswith("weight"):
                v.copy_(1.0 + 0.1 * torch.randn(v.shape, generator=g))
            else:
                v.copy_(0.05 * torch.randn(v.shape, generator=g))


def gen_frnet(manifest):
    from ldm.modules.id_embedding.iresnet import IResNet, IBasicBlock, iresnet100
    cfg = MINI_FR
    net = IResNet(IBasicBlock, list(cfg["layers"])).eval()
    shapes = O.frnet_param_shapes(cfg)
    assert keys_digest(shapes) == keys_digest(sd_shapes(net)), "mini iresnet key table mismatch"
    W = O.init_weights(shapes, seed=13)
    load_into(net, W)
    g = torch.Generator().manual_seed(41)
    x = torch.rand(2, 3, 112, 112, generator=g) * 2 - 1
    with torch.no_grad():
        out = net(x)
    torch.save(dict(cfg=cfg, weight_seed=13, x=x, out=out), os.path.join(OUT, "frnet_mini.pt"))
    manifest["frnet_mini.pt"] = "ldm/modules/id_embedding/iresnet.py IResNet.forward:166-181 (eval)"
    with torch.device("meta"):
        full = iresnet100()
    return keys_digest(sd_shapes(full))


def gen_manager(manifest):
    """EmbeddingManagerId.forward in training mode (embedding_manager.py:279-394) with MetaIdNet (meta_net.py:250-346);
    the FR backbone is a seeded 1-1-1-1 iresnet (weights file absent) and the embedder is a minimal stand-in for
    FrozenCLIPEmbedder's tokenizer / token-embedding attributes (the manager only uses them in __init__)."""
    import ldm.modules.id_embedding.meta_net as meta_net
    from ldm.modules.id_embedding.iresnet import IResNet, IBasicBlock

    def load_fr_net(self):
        self.id_model = IResNet(IBasicBlock, [1, 1, 1, 1]).eval()
        for p in self.id_model.parameters():
            p.requires_grad = False
    meta_net.MetaIdNet.load_fr_net = load_fr_net
    from ldm.modules.embedding_manager import EmbeddingManagerId

    width, inner, es, vocab = 768, 16, 2, 50000
    table = O.formula_table(vocab, width)
    tok_of = {"sks": 48136, "ks": 662, "face": 1710}

    class Tok:
        def __call__(self, s, **kw):
            ids = torch.full((1, 77), 49407, dtype=torch.long); ids[0, 0] = 49406; ids[0, 1] = tok_of[s]
            return {"input_ids": ids}

    class Emb(nn.Module):
        def forward(self, ids, **kw):
            return table[ids]
    embedder = types.SimpleNamespace(tokenizer=Tok(), transformer=types.SimpleNamespace(
        text_model=types.SimpleNamespace(embeddings=Emb())))
    torch.manual_seed(5)
    mgr = EmbeddingManagerId(embedder, placeholder_strings=["sks", "ks"], initializer_words=["face", "face"], max_ids=3,
                             num_embeds_per_token=es, meta_mlp_depth=1, loss_type="none", meta_inner_dim=inner, meta_heads=1,
                             use_rm_mlp=False, test_mode="coefficient", momentum=0.99, save_fp16=False)
    mgr.train()
    n_train = sum(p.numel() for p in mgr.meta_id_net.stylegan_mlp.parameters())
    # seeded weights under the oracle's names
    fr_shapes = O.frnet_param_shapes(MINI_FR, "embedding_manager.meta_id_net.id_model.")
    W = O.init_weights(fr_shapes, seed=17)
    load_into(mgr.meta_id_net.id_model, W, "embedding_manager.meta_id_net.id_model.")
    g = torch.Generator().manual_seed(51)
    Wm = torch.randn(es * inner, 512, generator=g); bm = 0.1 * torch.randn(es * inner, generator=g)
    with torch.no_grad():
        mgr.meta_id_net.stylegan_mlp.net[0].weight.copy_(Wm); mgr.meta_id_net.stylegan_mlp.net[0].bias.copy_(bm)
    basis = O.init_weights({"celeb_embeddings": (es, 1 + inner, width)}, seed=19)["celeb_embeddings"]
    id_coef0 = torch.randn(3, es, 1, inner, generator=g); id_emb0 = torch.randn(3, es, width, generator=g)
    mgr.id_coefficients = [id_coef0[i].clone() for i in range(3)]
    mgr.id_embeddings = [id_emb0[i].clone() for i in range(3)]
    B, S = 2, 64
    faces = torch.rand(B, S, S, 6, generator=g) * 2 - 1
    ids = torch.tensor([[1, 1], [2, 2]])
    tokens = torch.full((B, 77), 49407, dtype=torch.long); tokens[:, 0] = 49406
    tokens[0, 1:9] = torch.tensor([320, 1125, 539, 320, 1710, 539, 48136, 2533]); tokens[1, 1:4] = torch.tensor([48136, 1125, 320])
    embedded = table[tokens].clone()
    out = mgr(tokens, embedded.clone(), None, {"faces": faces, "ids": ids, "num_ids": torch.tensor([1, 1])}, basis)
    dout = O.formula_table(out.shape[0] * out.shape[1], out.shape[2], 0.9).reshape(out.shape) * 50.0
    gW, gb = torch.autograd.grad((out * dout).sum(), [mgr.meta_id_net.stylegan_mlp.net[0].weight, mgr.meta_id_net.stylegan_mlp.net[0].bias])
    torch.save(dict(fr_cfg=MINI_FR, fr_weight_seed=17, basis_seed=19, inner=inner, es=es, Wm=Wm, bm=bm, basis=basis,
                    id_coef0=id_coef0.reshape(3, -1), id_emb0=id_emb0.reshape(3, -1), faces=faces, ids=ids, tokens=tokens,
                    vocab=vocab, out=out.detach(), gW=gW, gb=gb, n_trainable=n_train,
                    id_coef1=torch.stack([c.detach().reshape(-1) for c in mgr.id_coefficients]),
                    id_emb1=torch.stack([e.detach().reshape(-1) for e in mgr.id_embeddings])),
               os.path.join(OUT, "manager_mini.pt"))
    manifest["manager_mini.pt"] = ("ldm/modules/embedding_manager.py EmbeddingManagerId.forward:279-394, _momentum_update:452-490; "
                                   "meta_net.py forward_multi_faces:329-346, _celebs_forward:250-302")


def gen_clip(manifest):
    """third-party arithmetic (transformers CLIP text layers): cross-check the restatement against the INSTALLED transformers
    CLIPTextModel on the same weights — the published algorithm, not the reference's monkey-patched wrapper."""
    from transformers import CLIPTextConfig, CLIPTextModel
    cfg = MINI_CLIP
    hf = CLIPTextModel(CLIPTextConfig(vocab_size=cfg["vocab"], hidden_size=cfg["width"], intermediate_size=cfg["mlp"],
                                      num_hidden_layers=cfg["layers"], num_attention_heads=cfg["heads"],
                                      max_position_embeddings=cfg["ctx_len"], hidden_act="quick_gelu", layer_norm_eps=1e-5,
                                      eos_token_id=cfg["vocab"] - 1, bos_token_id=cfg["vocab"] - 2, pad_token_id=1)).eval()
    hf_prefix = "text_model." if any(k.startswith("text_model.") for k in hf.state_dict()) else ""
    shapes = O.clip_param_shapes(cfg, hf_prefix)
    assert keys_digest(shapes) == keys_digest(sd_shapes(hf, skip=("position_ids",))), "mini CLIP key table mismatch"
    W = O.init_weights(shapes, seed=23)
    load_into(hf, W)
    g = torch.Generator().manual_seed(61)
    tokens = torch.randint(0, cfg["vocab"] - 2, (2, 77), generator=g); tokens[:, -1] = cfg["vocab"] - 1
    with torch.no_grad():
        out = hf(input_ids=tokens).last_hidden_state
    import transformers
    torch.save(dict(cfg=cfg, weight_seed=23, weight_prefix=hf_prefix, tokens=tokens, last_hidden_state=out, transformers_version=transformers.__version__),
               os.path.join(OUT, "clip_mini.pt"))
    manifest["clip_mini.pt"] = f"installed transformers=={transformers.__version__} CLIPTextModel (published CLIP text algorithm; reference pins 4.18.0)"
    with torch.device("meta"):
        full = CLIPTextModel(CLIPTextConfig(vocab_size=49408, hidden_size=768, intermediate_size=3072, num_hidden_layers=12,
                                            num_attention_heads=12, max_position_embeddings=77, hidden_act="quick_gelu"))
    fs = {k[len("text_model."):] if k.startswith("text_model.") else k: v for k, v in sd_shapes(full, skip=("position_ids",)).items()}
    return keys_digest(fs)



STEP_UNET = dict(in_channels=4, out_channels=4, model_channels=32, num_res_blocks=1, channel_mult=(1, 2, 2),
                 attention_resolutions=(1, 2), num_heads=2, context_dim=768)
STEP_CLIP = dict(vocab=50000, width=768, layers=1, heads=12, ctx_len=77, mlp=128)


def gen_step(manifest):
    """SURVEY C-8 item 6: K optimiser steps of the COMPOSED training step, run by the reference's own code —
    LatentDiffusion.forward / get_learned_conditioning / p_losses / apply_model (ddpm.py:926-936, 599-611, 1069-1116, 948-1049),
    DDPM.register_schedule / q_sample / get_loss (:126-178, 289-307), DiffusionWrapper.forward (:1537-1556) unbound on a stand-in
    `self`, over the reference UNetModel, Encoder + DiagonalGaussianDistribution (encode_first_stage + get_first_stage_encoding),
    EmbeddingManagerId + MetaIdNet (training mode) and torch.optim.AdamW exactly as configure_optimizers builds it (:1442-1454).
    The one piece that is NOT reference code is the CLIP layer stack (the reference wrapper cannot be imported offline, SURVEY
    C-5): the oracle's restatement, wired at the reference's hook point (token embedding -> manager -> + positions, layers).
    Two runs: the default MLP path (3 steps) and `use_rm_mlp` (2 steps, incl. the reference's flatten(ids) pairing)."""
    out, run = step_harness()
    run(False, 3, 3, lambda s_: torch.tensor([[1, 1], [2, 2]]), "mlp")
    run(True, 2, 10, lambda s_: torch.tensor([[1, 1], [2, 2]]) if s_ == 0 else torch.tensor([[4, 4], [4, 4]]), "rm_mlp")
    torch.save(out, os.path.join(OUT, "step_mini.pt"))
    manifest["step_mini.pt"] = ("ldm/models/diffusion/ddpm.py LatentDiffusion.forward:926-936, get_learned_conditioning:599-611, p_losses:1069-1116, "
                                "apply_model:948-1049, DDPM.register_schedule:126-178, q_sample:289-292, get_loss:294-307, DiffusionWrapper.forward:"
                                "1537-1556, torch.optim.AdamW as configure_optimizers:1442-1454 — over the reference UNetModel, Encoder, "
                                "DiagonalGaussianDistribution, EmbeddingManagerId / MetaIdNet (default and use_rm_mlp); CLIP layers = oracle restatement")


def step_harness(inner=16):
    """the reference's composed training step on stand-in objects (see gen_step): returns (out, run) where
    run(use_rm_mlp, K, max_ids, ids_of, tag, slim=False, fr_noise=0.0) records a K-step trajectory into out[tag].
    slim: per step only the loss and the EMA dictionary rows (inputs are re-drawn from their seeds: oracle/replay.py::ref_stream_batch);
    fr_noise: relative size of seeded noise added to the FR-net's features (sensitivity runs of oracle/gen_horizon_ref.py)."""
    import ldm.modules.id_embedding.meta_net as meta_net
    from ldm.modules.id_embedding.iresnet import IResNet, IBasicBlock
    from ldm.models.diffusion.ddpm import LatentDiffusion, DDPM, DiffusionWrapper
    from ldm.modules.diffusionmodules.openaimodel import UNetModel
    from ldm.modules.diffusionmodules.model import Encoder
    from ldm.modules.distributions.distributions import DiagonalGaussianDistribution

    def load_fr_net(self):
        self.id_model = IResNet(IBasicBlock, [1, 1, 1, 1]).eval()
        for p in self.id_model.parameters():
            p.requires_grad = False
    meta_net.MetaIdNet.load_fr_net = load_fr_net
    from ldm.modules.embedding_manager import EmbeddingManagerId

    ucfg, vcfg, ccfg, frcfg = STEP_UNET, MINI_VAE, STEP_CLIP, MINI_FR
    es, width, scale = 2, 768, 0.18215                          # inner: 16 in every fixture; 512 (the real head width) in gen_horizon_ref.py --inner 512
    table = O.formula_table(ccfg["vocab"], width)
    tok_of = {"sks": 48136, "ks": 662, "face": 1710}
    lr = 2 * 5e-3                                              # bs * base_lr (main_id_embed.py:778-782)
    out = dict(unet=ucfg, vae=vcfg, clip=ccfg, fr=frcfg, es=es, inner=inner, lr=lr, scale=scale,
               seeds=dict(unet=71, vae=72, clip=73, fr=74, basis=75))
    Wu = O.init_weights(O.unet_param_shapes(ucfg, "model.diffusion_model."), seed=71)
    Wv = O.init_weights(O.vae_encoder_param_shapes(vcfg, "first_stage_model."), seed=72)
    Wc = O.init_weights(O.clip_param_shapes(ccfg, "cond_stage_model.transformer.text_model."), seed=73)
    Wc["cond_stage_model.transformer.text_model.embeddings.token_embedding.weight"] = table
    Wf = O.init_weights(O.frnet_param_shapes(frcfg, "embedding_manager.meta_id_net.id_model."), seed=74)
    basis = O.init_weights({"celeb_embeddings": (es, 1 + inner, width)}, seed=75)["celeb_embeddings"]

    unet = UNetModel(image_size=16, in_channels=4, model_channels=ucfg["model_channels"], out_channels=4, num_res_blocks=ucfg["num_res_blocks"],
                     attention_resolutions=list(ucfg["attention_resolutions"]), channel_mult=list(ucfg["channel_mult"]),
                     num_heads=ucfg["num_heads"], use_spatial_transformer=True, transformer_depth=1, context_dim=width,
                     use_checkpoint=True, legacy=False)
    load_into(unet, Wu, "model.diffusion_model.")
    enc = Encoder(ch=vcfg["ch"], out_ch=3, ch_mult=vcfg["ch_mult"], num_res_blocks=vcfg["num_res_blocks"], attn_resolutions=[], dropout=0.0,
                  in_channels=3, resolution=64, z_channels=vcfg["z_channels"], double_z=True).eval()
    quant = nn.Conv2d(2 * vcfg["z_channels"], 2 * vcfg["z_channels"], 1)
    load_into(enc, Wv, "first_stage_model.encoder."); load_into(quant, Wv, "first_stage_model.quant_conv.")
    for m_ in (unet, enc, quant):
        for p_ in m_.parameters():
            p_.requires_grad = False                            # ddpm.py:491-500: everything pretrained is frozen

    class Tok:
        def __call__(self, s, **kw):
            ids = torch.full((1, 77), 49407, dtype=torch.long); ids[0, 0] = 49406; ids[0, 1] = tok_of[s]
            return {"input_ids": ids}

    class Emb(nn.Module):
        def forward(self, ids, **kw):
            return table[ids]

    def make_manager(use_rm_mlp, max_ids):
        embedder = types.SimpleNamespace(tokenizer=Tok(), transformer=types.SimpleNamespace(text_model=types.SimpleNamespace(embeddings=Emb())))
        torch.manual_seed(5)
        mgr = EmbeddingManagerId(embedder, placeholder_strings=["sks", "ks"], initializer_words=["face", "face"], max_ids=max_ids,
                                 num_embeds_per_token=es, meta_mlp_depth=1, loss_type="none", meta_inner_dim=inner, meta_heads=1,
                                 use_rm_mlp=use_rm_mlp, test_mode="coefficient", momentum=0.99, save_fp16=False)
        load_into(mgr.meta_id_net.id_model, Wf, "embedding_manager.meta_id_net.id_model.")
        return mgr

    class CondStage:                                             # FrozenCLIPEmbedder.encode with the reference's hook order (modules.py:176-298)
        def __init__(self):
            self.celeb_embeddings = basis

        def encode(self, tokens, embedding_manager=None, face_img=None, image_ori=None):
            embedded = table[tokens].clone()
            embedded = embedding_manager(tokens, embedded, face_img, image_ori, self.celeb_embeddings)
            return O.clip_encode(Wc, ccfg, embedded, "cond_stage_model.transformer.text_model.")

    class Stub(nn.Module):                                       # the attributes LatentDiffusion's methods read
        @property
        def device(self):
            return torch.device("cpu")

    def run(use_rm_mlp, K, max_ids, ids_of, tag, slim=False, fr_noise=0.0):
        mgr = make_manager(use_rm_mlp, max_ids)
        if fr_noise:
            gn = torch.Generator().manual_seed(4242)
            mgr.meta_id_net.id_model.register_forward_hook(
                lambda mod, inp, o: o + fr_noise * o.norm() / (o.numel() ** 0.5) * torch.randn(o.shape, generator=gn))
        g = torch.Generator().manual_seed(81 + int(use_rm_mlp))
        Wm = torch.randn(es * inner, 512, generator=g); bm = 0.1 * torch.randn(es * inner, generator=g)
        with torch.no_grad():
            mgr.meta_id_net.stylegan_mlp.net[0].weight.copy_(Wm); mgr.meta_id_net.stylegan_mlp.net[0].bias.copy_(bm)
        id_coef0 = torch.randn(max_ids, es, 1, inner, generator=g); id_emb0 = torch.randn(max_ids, es, width, generator=g)
        mgr.id_coefficients = [id_coef0[i].clone() for i in range(max_ids)]
        mgr.id_embeddings = [id_emb0[i].clone() for i in range(max_ids)]
        coef0 = None
        if use_rm_mlp:
            coef0 = torch.randn(tuple(mgr.meta_id_net.coef.shape), generator=g)
            with torch.no_grad():
                mgr.meta_id_net.coef.copy_(coef0)
        me = Stub()
        me.parameterization, me.v_posterior, me.loss_type, me.learn_logvar = "eps", 0.0, "l2", False
        DDPM.register_schedule(me, beta_schedule="linear", timesteps=1000, linear_start=0.00085, linear_end=0.012)
        me.logvar = torch.full((1000,), 0.0)                    # ddpm.py:110-113 logvar_init = 0
        me.l_simple_weight, me.original_elbo_weight, me.embedding_reg_weight = 1.0, 0.0, 0.0
        me.shorten_cond_schedule, me.cond_stage_trainable, me.cond_stage_forward = False, True, None
        dw = DiffusionWrapper.__new__(DiffusionWrapper); nn.Module.__init__(dw)
        dw.diffusion_model, dw.conditioning_key = unet, "crossattn"
        me.model, me.embedding_manager, me.cond_stage_model = dw, mgr, CondStage()
        for name in ("q_sample", "get_loss"):
            setattr(me, name, types.MethodType(getattr(DDPM, name), me))
        for name in ("forward", "p_losses", "apply_model", "get_learned_conditioning"):
            setattr(me, name, types.MethodType(getattr(LatentDiffusion, name), me))
        me.train(); mgr.train()
        # ddpm.py:504-507 flips every returned parameter to requires_grad=True; the frozen FR-net still gets no gradient (no_grad forward)
        params = mgr.embedding_parameters() + mgr.trainable_parameters()
        for p_ in params:
            p_.requires_grad = True
        opt = torch.optim.AdamW(params, lr=lr)                  # ddpm.py:1454
        B, S = 2, 64
        rec = dict(Wm=Wm, bm=bm, id_coef0=id_coef0.reshape(max_ids, -1), id_emb0=id_emb0.reshape(max_ids, -1), coef0=coef0, steps=[])
        for s_ in range(K):
            gi = torch.Generator().manual_seed(2000 + 10 * int(use_rm_mlp) + s_)        # images are re-drawn from this seed by the tests
            image = torch.rand(B, S, S, 3, generator=gi) * 2 - 1; faces = torch.rand(B, S, S, 6, generator=gi) * 2 - 1
            ids = ids_of(s_)
            tokens = torch.full((B, 77), 49407, dtype=torch.long); tokens[:, 0] = 49406
            tokens[0, 1:9] = torch.tensor([320, 1125, 539, 320, 1710, 539, 48136, 2533]); tokens[1, 1:5] = torch.tensor([1125, 48136, 1125, 320])
            with torch.no_grad():                               # get_input: encode_first_stage + get_first_stage_encoding (ddpm.py:702-759, 590-597)
                post = DiagonalGaussianDistribution(quant(enc(image.permute(0, 3, 1, 2))))
                torch.manual_seed(900 + s_)
                x = scale * post.sample()
                torch.manual_seed(900 + s_)
                post_noise = torch.randn(post.mean.shape)
            torch.manual_seed(1000 + s_)                        # LatentDiffusion.forward draws t, p_losses draws the noise
            loss, loss_dict = me(x, tokens, face_img=None, image_ori={"faces": faces, "ids": ids, "num_ids": torch.tensor([1, 1])})
            torch.manual_seed(1000 + s_)
            t = torch.randint(0, 1000, (B,)).long(); noise = torch.randn_like(x)
            opt.zero_grad()
            loss.backward()
            gmain = (mgr.meta_id_net.coef.grad if use_rm_mlp else mgr.meta_id_net.stylegan_mlp.net[0].weight.grad).detach().clone()
            assert all(p_.grad is None for p_ in mgr.meta_id_net.id_model.parameters())       # FR-net never updated
            opt.step()
            mgr.id_coefficients = [c.detach() for c in mgr.id_coefficients]                 # cut the EMA's autograd chain (values unchanged)
            mgr.id_embeddings = [e.detach() for e in mgr.id_embeddings]
            if slim:
                rec["steps"].append(dict(loss=float(loss.detach()), id_coef=torch.stack([c.reshape(-1) for c in mgr.id_coefficients]).clone()))
                if (s_ + 1) % 100 == 0:
                    print(f"[gen_step {tag}] step {s_ + 1}/{K} loss {float(loss):.6f}", flush=True)
                continue
            rec["steps"].append(dict(image_seed=2000 + 10 * int(use_rm_mlp) + s_, ids=ids, tokens=tokens, t=t, noise=noise, post_noise=post_noise,
                                     loss=loss.detach().clone(), z0=x, grad=gmain if s_ == 0 else None,
                                     loss_simple=loss_dict["train/loss_simple"].detach().clone()))
        rec["W_final"] = mgr.meta_id_net.stylegan_mlp.net[0].weight.detach().clone()
        rec["b_final"] = mgr.meta_id_net.stylegan_mlp.net[0].bias.detach().clone()
        rec["coef_final"] = mgr.meta_id_net.coef.detach().clone() if use_rm_mlp else None
        rec["id_coef_final"] = torch.stack([c.reshape(-1) for c in mgr.id_coefficients])
        rec["id_emb_final"] = torch.stack([e.reshape(-1) for e in mgr.id_embeddings])
        out[tag] = rec

    return out, run


def main():
    os.makedirs(OUT, exist_ok=True)
    import transformers  # noqa: F401  (resolve its optional-dependency probes before the torchvision stub exists)
    from transformers import CLIPTokenizer, CLIPTextModel  # noqa: F401
    install_shims()
    torch.set_num_threads(8)
    only = None
    for a in sys.argv[1:]:
        if a.startswith("--only="):          # e.g. --only=vae_dec,ddim : regenerate some fixtures, keep the other entries
            only = set(a[len("--only="):].split(","))
    manifest, digests = {}, {}
    if only is not None:
        manifest = json.load(open(os.path.join(OUT, "MANIFEST.json")))
        digests = json.load(open(os.path.join(OUT, "state_dict_digests.json")))
    want = lambda name: only is None or name in only
    if want("schedule"):
        gen_schedule(manifest)
    if want("shift_rows"):
        gen_shift_rows(manifest)
    if want("unet"):
        digests["unet_sd14"] = gen_unet(manifest)
    if want("vae"):
        digests["vae_encoder_sd14"] = gen_vae(manifest)
    if want("vae_dec"):
        digests["vae_decoder_sd14"] = gen_vae_decoder(manifest)
    if want("ddim"):
        gen_ddim(manifest)
    if want("basis"):
        gen_basis(manifest)
    if want("templates"):
        digests["imagenet_templates_small"] = gen_templates()
        manifest["state_dict_digests.json:imagenet_templates_small"] = "sha256 + count of ldm/data/face_id.py:17-99 imagenet_templates_small (gen_templates)"
    if want("config"):
        gen_config_params(manifest)
    if want("frnet"):
        digests["iresnet100"] = gen_frnet(manifest)
    if want("manager"):
        gen_manager(manifest)
    if want("clip"):
        digests["clip_l14_text"] = gen_clip(manifest)
    if want("step"):
        gen_step(manifest)
    with open(os.path.join(OUT, "state_dict_digests.json"), "w") as f:
        json.dump(digests, f, indent=1, sort_keys=True)
    manifest["state_dict_digests.json"] = "sha256 over sorted 'key:shape' lines of the reference modules' float state_dict at SD v1-4 sizes"
    # fixtures written by oracle/gen_horizon.py (minutes of CPU each, not regenerated here): ORACLE trajectories, not reference outputs
    manifest["horizon_sd14_b1.pt"] = ("oracle/gen_horizon.py fp32: 100 optimiser steps of oracle.cb_oracle.train_step (the restatement step_mini.pt pins) "
                                      "at SD v1-4 size, bs=1, replay stream oracle/replay.py — losses, step coefficients, EMA dictionary rows, sampled MLP gradient rows")
    manifest["horizon_sd14_b4.pt"] = "oracle/gen_horizon.py bf16: the same for 20 steps at bs=4, lr 2e-2 (fp32 oracle; checks the bf16 step)"
    manifest["horizon_sd14_b1_800.pt"] = ("oracle/gen_horizon.py fp32long: the fp32 run continued to BASELINE config 2's own length, 800 optimiser steps "
                                          "(aigc_id.yaml:135); slim: losses, step coefficients, EMA dictionary rows")
    manifest["horizon_sd14_b1_photo.pt"] = ("oracle/gen_horizon.py fp32photo: 800 steps of the fp32 run on the replay stream with low-pass (photo-like) face crops "
                                            "(oracle/replay.py::low_pass); slim")
    for tag, ws, bs in (("photo2", 9, 3000), ("photo3", 10, 5000)):
        manifest[f"horizon_sd14_b1_{tag}.pt"] = (f"oracle/gen_horizon.py fp32{tag}: another draw of the 800-step photo-like run — frozen weights init_weights({ws}), batch stream "
                                                 f"seeds {bs}+; slim: loss per step, EMA dictionary rows of every 4th step")
    manifest["horizon_sd14_b1_noise2.pt"] = ("oracle/gen_horizon.py fp32noise2: another draw of the DEFAULT (white-noise) 800-step run — frozen weights init_weights(11), "
                                             "batch stream seeds 7000+; slim: loss per step, EMA dictionary rows of every 4th step")
    manifest["warp_linspace_512.json"] = ("torch.linspace(-1, 1, 512) (the x / y base grid of F.affine_grid, meta_net.py:253-257) as float32 hex on the fixture host; "
                                          "fallback data of oracle/replay.py::fixture_host_warp_grid")
    manifest["horizon_mini_ref800.pt"] = ("oracle/gen_horizon_ref.py: 800 optimiser steps of the REFERENCE's composed step (the gen_step harness: ddpm.py "
                                          "LatentDiffusion.forward / p_losses / q_sample / get_loss, UNetModel, Encoder, EmbeddingManagerId + MetaIdNet, AdamW) at "
                                          "mini size — per-step loss + EMA dictionary rows; the same with seeded FR-feature noise (3e-6 / 2.5e-5 / 2e-4); "
                                          "the oracle's trajectory on the same stream")
    with open(os.path.join(OUT, "MANIFEST.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    for k in sorted(os.listdir(OUT)):
        print(f"{os.path.getsize(os.path.join(OUT, k)):9d}  {k}")


if __name__ == "__main__":
    main()
