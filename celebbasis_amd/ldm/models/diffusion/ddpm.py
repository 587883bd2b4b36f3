"""Host-side mirror of ldm/models/diffusion/ddpm.py `LatentDiffusion` (reference :439-1528) restricted to the CelebBasis
personalization path: config-driven construction (aigc_id.yaml:1-90), SD state-dict loading by the reference's key prefixes,
`training_step` = one native cb_trainer_step (VAE encode -> manager -> CLIP -> q_sample -> UNet -> loss -> dgrad -> AdamW),
`on_save_checkpoint` writing the reference's embeddings_gs-*.pt format."""
import os

import numpy as np
import torch

from .... import lib as _lib
from ...modules.frozen import context_for
from ...util import instantiate_from_config


def make_beta_schedule_linear(n_timestep, linear_start, linear_end):
    """util.py:21-25 'linear' (sqrt-space linspace, float64)"""
    return (torch.linspace(linear_start ** 0.5, linear_end ** 0.5, n_timestep, dtype=torch.float64) ** 2).numpy()


class DiffusionWrapper:
    def __init__(self, diffusion_model, conditioning_key):
        self.diffusion_model, self.conditioning_key = diffusion_model, conditioning_key


class LatentDiffusion:
    def __init__(self, first_stage_config, cond_stage_config, unet_config, personalization_config=None, timesteps=1000,
                 beta_schedule="linear", linear_start=1e-4, linear_end=2e-2, num_timesteps_cond=1, cond_stage_key="caption",
                 first_stage_key="image", cond_stage_trainable=False, conditioning_key="crossattn", scale_factor=1.0,
                 image_size=64, channels=4, use_ema=False, embedding_reg_weight=0.0, unfreeze_model=False, model_lr=0.0,
                 log_every_t=200, monitor=None, parameterization="eps", loss_type="l2", **ignored):
        if beta_schedule != "linear" or parameterization != "eps" or loss_type != "l2" or conditioning_key != "crossattn":
            raise NotImplementedError("only the shipped CelebBasis configuration (linear / eps / l2 / crossattn) is on the hot path")
        if use_ema or unfreeze_model:
            raise NotImplementedError("use_ema / unfreeze_model are off in aigc_id.yaml and outside the hot path")
        self.first_stage_key, self.cond_stage_key = first_stage_key, cond_stage_key
        self.cond_stage_trainable, self.scale_factor = cond_stage_trainable, scale_factor
        self.image_size, self.channels, self.num_timesteps = image_size, channels, int(timesteps)
        self.linear_start, self.linear_end = linear_start, linear_end
        self.learning_rate = None            # set by the caller, main_id_embed.py:779
        self.training = True
        self.global_step = 0
        self.model = DiffusionWrapper(instantiate_from_config(unet_config), conditioning_key)
        self.first_stage_model = instantiate_from_config(first_stage_config)
        self.cond_stage_model = instantiate_from_config(cond_stage_config)
        self.embedding_manager = instantiate_from_config(personalization_config, embedder=self.cond_stage_model) \
            if personalization_config is not None else None
        self.register_schedule(linear_start, linear_end, timesteps)
        self._trainer = None

    # ---- ddpm.py:126-178
    def register_schedule(self, linear_start, linear_end, timesteps):
        betas = make_beta_schedule_linear(timesteps, linear_start, linear_end)
        ac = np.cumprod(1.0 - betas, axis=0)
        f32 = lambda a: torch.tensor(a, dtype=torch.float32)
        self.betas, self.alphas_cumprod = f32(betas), f32(ac)
        self.alphas_cumprod_prev = f32(np.append(1.0, ac[:-1]))
        self.sqrt_alphas_cumprod, self.sqrt_one_minus_alphas_cumprod = f32(np.sqrt(ac)), f32(np.sqrt(1.0 - ac))

    # ---- checkpoints: main_id_embed.py:25-44 loads with strict=False under these prefixes
    def load_state_dict(self, sd, strict=False):
        m1, _ = self.model.diffusion_model.load_state_dict(sd, strict=False, prefix="model.diffusion_model.")
        m2, _ = self.first_stage_model.load_state_dict(sd, strict=False, prefix="first_stage_model.")
        m3, _ = self.cond_stage_model.load_state_dict(sd, strict=False, prefix="cond_stage_model.")
        missing = m1 + m2 + m3
        if self.embedding_manager is not None:
            m4, _ = self.embedding_manager.meta_id_net.id_model.load_state_dict(sd, strict=False,
                                                                                prefix="embedding_manager.meta_id_net.id_model.")
            missing += m4
            if "celeb_embeddings" in sd:
                self.cond_stage_model.load_celeb_basis(sd["celeb_embeddings"])
            tk = "cond_stage_model." + self.cond_stage_model.PREFIX + "embeddings.token_embedding.weight"
            if tk in sd:       # embedding_manager.py:239-253: id_embeddings start at the initializer word's embedding
                self.embedding_manager.init_id_embeddings(sd[tk])
        if strict and missing:
            raise RuntimeError(f"missing keys: {missing[:5]}…")
        self._trainer = None
        return missing, []

    # ---- inference surface used by DDIMSampler / stable_txt2img.py
    @property
    def device(self):
        return torch.device("cuda", torch.cuda.current_device())

    def ema_scope(self, context=None):
        """ddpm.py:180-193; use_ema is False in aigc_id.yaml, so the scope is a no-op"""
        import contextlib
        return contextlib.nullcontext()

    def get_learned_conditioning(self, c, face_img=None, image_ori=None):
        """ddpm.py:599-611: cond_stage_model.encode(c, embedding_manager=..., face_img=..., image_ori=...)"""
        return self.cond_stage_model.encode(c, embedding_manager=self.embedding_manager, face_img=face_img, image_ori=image_ori)

    def apply_model(self, x_noisy, t, cond, return_ids=False):
        """ddpm.py:948-1049 with conditioning_key 'crossattn': UNet(x, t, context=cond)"""
        if isinstance(cond, dict):
            cond = cond["c_crossattn"][0] if isinstance(cond["c_crossattn"], (list, tuple)) else cond["c_crossattn"]
        if isinstance(cond, (list, tuple)):
            cond = cond[0]
        unet = self.model.diffusion_model.native(x_noisy.device)
        return unet.forward(x_noisy.float().contiguous(), t.to(torch.int64).contiguous(), cond.float().contiguous())

    def decode_first_stage(self, z, predict_cids=False, force_not_quantize=False):
        """ddpm.py:761-819: z = 1/scale_factor * z; first_stage_model.decode(z)"""
        return self.first_stage_model.decode(z, scale_factor=self.scale_factor)

    def q_sample(self, x_start, t, noise=None):
        """ddpm.py:289-292: sqrt(ac_t) * x_start + sqrt(1 - ac_t) * noise on NCHW latents, natively (cb_q_sample between two layout
        changes). Inside training_step the same kernel runs fused in cb_trainer_step; this entry serves stand-alone callers."""
        dev = x_start.device
        ctx = context_for(dev)
        if ctx.dtype != _lib.DTYPE_F32:
            raise NotImplementedError("stand-alone q_sample runs on an fp32 context (public latents are fp32)")
        L = _lib.load()
        B, Cc, H, W = x_start.shape
        x = x_start.float().contiguous()
        nz = (torch.randn_like(x) if noise is None else noise.to(dev).float()).contiguous()
        tt = t.to(dev).to(torch.int64).contiguous()
        sa, sb = self.sqrt_alphas_cumprod.to(dev).contiguous(), self.sqrt_one_minus_alphas_cumprod.to(dev).contiguous()
        x_nhwc, xt_nhwc, out = torch.empty(B, H, W, Cc, device=dev), torch.empty(B, H, W, Cc, device=dev), torch.empty_like(x)
        _lib.check(L.cb_nchw_to_nhwc(ctx.h, _lib.ptr(x), _lib.ptr(x_nhwc), B, Cc, H, W))
        _lib.check(L.cb_q_sample(ctx.h, _lib.ptr(x_nhwc), _lib.ptr(nz), _lib.ptr(tt, torch.int64), _lib.ptr(sa), _lib.ptr(sb),
                                 _lib.ptr(xt_nhwc), B, Cc, H * W))
        _lib.check(L.cb_nhwc_to_nchw(ctx.h, _lib.ptr(xt_nhwc), _lib.ptr(out), B, Cc, H, W))
        return out

    def train(self, mode=True):
        self.training = mode
        if self.embedding_manager is not None:
            self.embedding_manager.train(mode)
        return self

    def eval(self):
        return self.train(False)

    # ---- native trainer
    def _state_for_trainer(self):
        em = self.embedding_manager
        sd = {}
        sd.update(self.model.diffusion_model.state_dict("model.diffusion_model."))
        sd.update(self.first_stage_model.state_dict("first_stage_model.", only=self.first_stage_model.ENC_KEYS))   # the step never decodes
        sd.update(self.cond_stage_model.state_dict("cond_stage_model."))
        sd.update(em.meta_id_net.id_model.state_dict("embedding_manager.meta_id_net.id_model."))
        sd["embedding_manager.meta_id_net.stylegan_mlp.net.0.weight"] = em.meta_id_net.weight.data
        sd["embedding_manager.meta_id_net.stylegan_mlp.net.0.bias"] = em.meta_id_net.bias.data
        if em.use_rm_mlp:
            sd["embedding_manager.meta_id_net.coef"] = em.meta_id_net.coef.data
        if self.cond_stage_model.celeb_embeddings is None:
            raise RuntimeError("the celeb basis is not loaded: cond_stage_model.load_celeb_basis(tensor (es,1+K,768))")
        sd["celeb_embeddings"] = self.cond_stage_model.celeb_embeddings
        sd["id_coefficients"] = torch.stack([c.reshape(-1).float() for c in em.id_coefficients])
        sd["id_embeddings"] = torch.stack([e.reshape(-1).float() for e in em.id_embeddings])
        return sd

    def trainer(self, device):
        if self._trainer is None:
            em = self.embedding_manager
            cfg = _lib.trainer_cfg_default()
            cfg.unet = _lib.unet_cfg(**self.model.diffusion_model.cfg)
            cfg.vae = _lib.vae_cfg(**self.first_stage_model.cfg)
            cfg.clip = _lib.clip_cfg(**self.cond_stage_model.cfg)
            fr = em.meta_id_net.id_model
            for i, v in enumerate(fr.layers):
                cfg.fr_layers[i] = v
            cfg.fr_width, cfg.fr_in = fr.width, fr.in_size
            cfg.image_size = self.image_size * 2 ** (len(self.first_stage_model.cfg["ch_mult"]) - 1)
            cfg.es, cfg.inner_dim, cfg.max_ids, cfg.momentum = em.num_es, em.inner_dim, em.max_ids, em.momentum
            cfg.scale_factor, cfg.linear_start, cfg.linear_end, cfg.timesteps = self.scale_factor, self.linear_start, self.linear_end, self.num_timesteps
            cfg.placeholder_token = em.string_to_token_dict[em.placeholder_strings[0]]
            cfg.use_rm_mlp = 1 if em.use_rm_mlp else 0
            self._trainer = _lib.Trainer(context_for(device), _lib.WeightTable(self._state_for_trainer()), cfg)
        return self._trainer

    def _first_stage_inputs(self, batch):
        """device tensors the native first stage reads (image, faces, posterior noise); cached on the batch so that a
        prefetch and the step that follows hand the very same buffers to the library"""
        prep = batch.get("_first_stage")
        if prep is None:
            image = batch[self.first_stage_key]
            dev, B, lat = image.device, image.shape[0], self.image_size
            post = batch.get("post_noise")
            post = torch.randn(B, self.channels, lat, lat, device=dev) if post is None else post.to(dev)
            prep = (image.float().contiguous(), batch["image_ori"]["faces"].to(dev).float().contiguous(), post.float().contiguous())
            batch["_first_stage"] = prep
        return prep

    def prefetch_first_stage(self, next_batch):
        """optional cross-step pipelining (cb_trainer_prefetch): call with batch i+1 before training_step(batch i)"""
        image, faces, post = self._first_stage_inputs(next_batch)
        self.trainer(image.device).prefetch(image, faces, post)

    def training_step(self, batch, batch_idx=0):
        """ddpm.py:357-370. batch keys as produced by FaceIdDatasetStyleGAN3 (face_id.py:598-644): 'image' (B,H,W,3),
        'caption' (strings or pre-tokenised ids) / 'tokens', 'image_ori' {'faces','ids','num_ids'}. Optional replay keys
        't', 'noise', 'post_noise' pin the RNG stream (otherwise drawn on the device like ddpm.py:927, 1070, distributions.py:36)."""
        image = batch[self.first_stage_key]
        dev = image.device
        tr = self.trainer(dev)
        B = image.shape[0]
        tokens = batch["tokens"] if "tokens" in batch else self.cond_stage_model.tokenize(batch[self.cond_stage_key])
        tokens = tokens.to(dev).to(torch.int64).contiguous()
        lat = self.image_size
        t = batch.get("t")
        t = torch.randint(0, self.num_timesteps, (B,), device=dev) if t is None else t.to(dev)
        noise = batch.get("noise")
        noise = torch.randn(B, self.channels, lat, lat, device=dev) if noise is None else noise.to(dev)
        image_c, faces_c, post = self._first_stage_inputs(batch)
        ori = batch["image_ori"]
        assert int(ori["num_ids"].max()) == 1, "training prompts carry one identity (dual_img is never set, face_id.py:612)"
        ids_c, t_c, noise_c = ori["ids"].to(dev).to(torch.int64).contiguous(), t.to(torch.int64).contiguous(), noise.float().contiguous()
        tr.step(image_c, faces_c, ids_c, tokens, t_c, noise_c, post, float(self.learning_rate))
        # the step is only ENQUEUED here; its reads are ordered on this context's stream (the side stream is joined inside the
        # call), so the temporaries above can be dropped: the caching allocator recycles them in stream order
        self.global_step += 1
        return tr.loss_tensor(dev)          # ddpm.py:357-370 returns `loss`: a 0-dim device tensor, copied in stream order (no sync)

    def last_loss(self):
        return float(self._trainer.get_state("loss")[0])

    def configure_optimizers(self):
        """ddpm.py:1442-1454: `torch.optim.AdamW(embedding_params, lr=lr)` over embedding_parameters() + trainable_parameters().
        The object is returned for callers that inspect it (param groups, lr); the update itself is fused into the native step
        (cb_trainer_step applies the same single-tensor AdamW with torch's defaults), so its .step() has nothing to do — the
        Parameters carry no .grad, exactly like the frozen FR-net's in the reference."""
        params = list(self.embedding_manager.embedding_parameters()) + list(self.embedding_manager.trainable_parameters())
        return torch.optim.AdamW(params, lr=self.learning_rate)

    def sync_manager_from_trainer(self):
        em, tr = self.embedding_manager, self._trainer
        if tr is None:
            return
        es, K, W = em.num_es, em.inner_dim, em.token_dim
        coef = tr.get_state("id_coefficients").reshape(em.max_ids, es, em.meta_heads, K)
        emb = tr.get_state("id_embeddings").reshape(em.max_ids, es * em.meta_heads, W)
        em.id_coefficients = [coef[i].clone() for i in range(em.max_ids)]
        em.id_embeddings = [emb[i].clone() for i in range(em.max_ids)]
        em.meta_id_net.weight.data = tr.get_state("mlp.weight").reshape(es * K, -1)       # same Parameter objects, refreshed
        em.meta_id_net.bias.data = tr.get_state("mlp.bias")
        if em.use_rm_mlp:
            em.meta_id_net.coef.data = tr.get_state("rm.coef").reshape(em.max_ids, es, em.meta_heads, K)

    def on_save_checkpoint(self, checkpoint, ckptdir="."):
        """ddpm.py:1519-1528: clear the (frozen) model checkpoint, write embeddings.pt + embeddings_gs-{step}.pt"""
        checkpoint.clear()
        self.sync_manager_from_trainer()
        os.makedirs(ckptdir, exist_ok=True)
        self.embedding_manager.save(os.path.join(ckptdir, "embeddings.pt"))
        self.embedding_manager.save(os.path.join(ckptdir, f"embeddings_gs-{self.global_step}.pt"))
