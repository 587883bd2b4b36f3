"""Seeded synthetic weights / replay batches at the true SD v1-4 shapes (no checkpoints or datasets exist offline:
SURVEY.md §8 C-7, §8 d). Used by bench.py, smoke() and the examples; parity tests use the oracle's own generator."""
import math

import torch

from .ldm.models.autoencoder import AutoencoderKL
from .ldm.modules.diffusionmodules.openaimodel import UNetModel
from .ldm.modules.encoders.modules import FrozenCLIPEmbedder
from .ldm.modules.id_embedding.iresnet import IResNet

UNET_SD14 = dict(in_channels=4, out_channels=4, model_channels=320, num_res_blocks=2, channel_mult=(1, 2, 4, 4),
                 attention_resolutions=(4, 2, 1), num_heads=8, context_dim=768)
VAE_SD14 = dict(ch=128, ch_mult=(1, 2, 4, 4), num_res_blocks=2, in_channels=3, z_channels=4)
CLIP_L14 = dict(vocab=49408, width=768, layers=12, heads=12, ctx_len=77, mlp=3072)
SKS = 48136
# "a photo of a face of sks person" with the reference's quoted ids (modules.py:198-262); 1710 = face, 2533 = person
PROMPT_IDS = [49406, 320, 1125, 539, 320, 1710, 539, SKS, 2533, 49407]


def trainer_shapes(unet=UNET_SD14, vae=VAE_SD14, clip=CLIP_L14, fr_layers=(3, 13, 30, 3), fr_width=64, fr_in=112, es=2, inner=512,
                   max_ids=10):
    S = {}
    S.update({"model.diffusion_model." + k: v for k, v in UNetModel.param_shapes(unet).items()})
    S.update({"first_stage_model." + k: v for k, v in AutoencoderKL.param_shapes(vae).items()
              if k.startswith(AutoencoderKL.ENC_KEYS)})          # the training step never touches the decoder half
    S.update({"cond_stage_model.transformer.text_model." + k: v for k, v in FrozenCLIPEmbedder.param_shapes(clip).items()})
    S.update({"embedding_manager.meta_id_net.id_model." + k: v for k, v in IResNet.param_shapes(fr_layers, 512, fr_width, fr_in).items()})
    S["embedding_manager.meta_id_net.stylegan_mlp.net.0.weight"] = (es * inner, 512)
    S["embedding_manager.meta_id_net.stylegan_mlp.net.0.bias"] = (es * inner,)
    S["celeb_embeddings"] = (es, 1 + inner, clip["width"])
    S["id_embeddings"] = (max_ids, es * clip["width"])
    S["id_coefficients"] = (max_ids, es * inner)
    return S


def decoder_shapes(vae=VAE_SD14):
    """first_stage_model.decoder.* + post_quant_conv.* (sampling path only)"""
    return {"first_stage_model." + k: v for k, v in AutoencoderKL.param_shapes(vae).items() if k.startswith(AutoencoderKL.DEC_KEYS)}


def synthetic_state(shapes, seed=0, device="cpu"):
    """fan-in scaled normals for weights, ~1 gammas, small betas, positive running_var; zero-modules re-randomised so that
    the output and the gradients are non-trivial (with the true zero-init `out` conv the loss would be constant)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    out = {}
    for name, shape in shapes.items():
        leaf = name.rsplit(".", 1)[-1]
        if leaf == "running_var":
            t = 0.5 + torch.rand(shape, generator=g)
        elif leaf == "running_mean":
            t = 0.1 * torch.randn(shape, generator=g)
        elif name == "celeb_embeddings":
            es, k1, w = shape
            q, _ = torch.linalg.qr(torch.randn(w, k1 - 1, generator=g))
            t = torch.cat([0.02 * torch.randn(es, 1, w, generator=g), q.t().unsqueeze(0).repeat(es, 1, 1)], dim=1)
        elif name in ("id_embeddings", "id_coefficients") or name.endswith("stylegan_mlp.net.0.weight"):
            t = torch.randn(shape, generator=g)
        elif len(shape) == 1:
            is_scale = leaf == "weight"
            t = (1.0 + 0.1 * torch.randn(shape, generator=g)) if is_scale else 0.05 * torch.randn(shape, generator=g)
            if "prelu" in name:
                t = 0.25 + 0.05 * torch.randn(shape, generator=g)
        elif "embedding.weight" in name:
            t = 0.02 * torch.randn(shape, generator=g)
        else:
            fan_in = 1
            for v in shape[1:]:
                fan_in *= v
            t = torch.randn(shape, generator=g) / math.sqrt(fan_in)
        out[name] = t.contiguous().to(device)
    return out


def synthetic_batch(B, step, image_size=512, latent=64, seed=23, device="cpu", vocab=49408, identity=1):
    """one replay-stream element (SURVEY §8 d): image/faces U[-1,1], the fixed prompt ids, t ~ randint, eps/posterior-eps ~ N(0,1)
    from a generator keyed by (seed, step) — the same stream can be fed to the oracle."""
    g = torch.Generator(device="cpu").manual_seed(seed * 1000003 + step)
    tokens = torch.full((B, 77), 49407, dtype=torch.int64)
    tokens[:, :len(PROMPT_IDS)] = torch.tensor(PROMPT_IDS)
    b = dict(image=torch.rand(B, image_size, image_size, 3, generator=g) * 2 - 1,
             faces=torch.rand(B, image_size, image_size, 6, generator=g) * 2 - 1,
             ids=torch.full((B, 2), identity, dtype=torch.int64), tokens=tokens,
             t=torch.randint(0, 1000, (B,), generator=g), noise=torch.randn(B, 4, latent, latent, generator=g),
             post_noise=torch.randn(B, 4, latent, latent, generator=g))
    return {k: v.to(device).contiguous() for k, v in b.items()}
