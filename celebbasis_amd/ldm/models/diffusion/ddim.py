"""Host-side mirror of the reference's `DDIMSampler` (ldm/models/diffusion/ddim.py:12-204), eps-parameterisation.

Drop-in at the API level: same constructor, the same attribute names after `make_schedule` and the same `sample(...)`
keywords, so `scripts/stable_txt2img.py:330-340` runs unchanged. What differs is where the work happens: the schedule is a
handful of host scalars, and the S-step loop — the batch-2n classifier-free-guidance UNet pass plus the x_{t-1} update — runs
inside libcelebbasis_hip.so (`cb_unet_ddim_sample`) without returning to Python between steps. Per-step hooks (`callback`,
`img_callback`, `score_corrector`, `noise_dropout`) use a step-by-step loop over the same native ops (`apply_model` +
`cb_ddim_update`). `quantize_x0` (VQ first stages) and inpainting masks are outside the CelebBasis path.
"""
import numpy as np
import torch

from .... import lib as _lib
from ...modules.frozen import context_for


def ddim_timesteps(method, n_ddim, n_ddpm):
    """the sub-sequence of DDPM steps visited (util.py:46-60): uniform stride or quadratic spacing, shifted by +1"""
    if method == "uniform":
        base = np.arange(0, n_ddpm, n_ddpm // n_ddim)
    elif method == "quad":
        base = (np.linspace(0, np.sqrt(0.8 * n_ddpm), n_ddim) ** 2).astype(int)
    else:
        raise NotImplementedError(f'There is no ddim discretization method called "{method}"')
    return base + 1


def ddim_tables(acp, steps, eta):
    """Per-step constants of the DDIM update (util.py:63-74 + ddim.py:51).

    acp: float32 tensor of cumulative alphas. The reference mixes precisions here and the mix is reproduced because the
    values end up in fp32 `torch.full` tensors: a_t is gathered in fp32; a_{t-1} is the same fp32 values widened to float64;
    sigma is evaluated in float64; sqrt(1 - a_t) is evaluated in fp32."""
    a_t = acp[torch.as_tensor(steps)]                                            # fp32 tensor
    a_prev = np.concatenate([[float(acp[0])], a_t[:-1].double().numpy()])        # float64 array of fp32 values
    a32 = a_t.numpy()
    # float64, except that `(1 - a_prev) / (1 - a_t)` is ndarray / fp32-tensor in the reference, which torch evaluates as
    # reciprocal(fp32) * other: the reciprocal is rounded to fp32 before it is widened
    recip32 = (np.float32(1.0) / (1 - a32)).astype(np.float32)
    sigma = eta * np.sqrt((1 - a_prev) * recip32 * (1 - a32 / a_prev))
    return dict(ddim_sigmas=sigma, ddim_alphas=a_t, ddim_alphas_prev=a_prev, ddim_sqrt_one_minus_alphas=torch.sqrt(1. - a_t))


class DDIMSampler(object):
    def __init__(self, model, schedule="linear", **kwargs):
        self.model, self.schedule = model, schedule
        self.ddpm_num_timesteps = model.num_timesteps

    def register_buffer(self, name, attr):
        setattr(self, name, attr)          # tables stay on the host: the native loop takes them as plain arrays

    def make_schedule(self, ddim_num_steps, ddim_discretize="uniform", ddim_eta=0., verbose=True):
        """registers the buffers of ddim.py:25-54 under the reference's names"""
        acp = torch.as_tensor(self.model.alphas_cumprod).detach().float().cpu()
        if acp.shape[0] != self.ddpm_num_timesteps:
            raise AssertionError("alphas have to be defined for each timestep")
        prev = torch.as_tensor(self.model.alphas_cumprod_prev).detach().float().cpu()
        self.ddim_timesteps = ddim_timesteps(ddim_discretize, ddim_num_steps, self.ddpm_num_timesteps)
        if int(self.ddim_timesteps.max()) >= self.ddpm_num_timesteps:
            # same failure as the reference (its `alphacums[ddim_timesteps]` raises too), with the reason spelled out
            raise IndexError(f"ddim_num_steps={ddim_num_steps} does not divide {self.ddpm_num_timesteps}: the '+1' shifted schedule "
                             f"reaches step {int(self.ddim_timesteps.max())}")
        full = dict(betas=torch.as_tensor(self.model.betas).detach().float().cpu(), alphas_cumprod=acp, alphas_cumprod_prev=prev,
                    sqrt_alphas_cumprod=acp.sqrt(), sqrt_one_minus_alphas_cumprod=(1. - acp).sqrt(),
                    log_one_minus_alphas_cumprod=(1. - acp).log(), sqrt_recip_alphas_cumprod=(1. / acp).sqrt(),
                    sqrt_recipm1_alphas_cumprod=(1. / acp - 1).sqrt(),
                    ddim_sigmas_for_original_num_steps=ddim_eta * torch.sqrt((1 - prev) / (1 - acp) * (1 - acp / prev)))
        full.update(ddim_tables(acp, self.ddim_timesteps, ddim_eta))
        for name, value in full.items():
            self.register_buffer(name, value)
        if verbose:
            print(f"DDIM schedule: {len(self.ddim_timesteps)} of {self.ddpm_num_timesteps} steps {self.ddim_timesteps}, eta {ddim_eta}")

    def _tables(self, use_original_steps=False):
        """(a_t, a_prev, sigma, sqrt(1-a_t)) as float32 arrays — the values the reference puts into torch.full (ddim.py:186-189)"""
        names = ("alphas_cumprod", "alphas_cumprod_prev", "ddim_sigmas_for_original_num_steps", "sqrt_one_minus_alphas_cumprod") \
            if use_original_steps else ("ddim_alphas", "ddim_alphas_prev", "ddim_sigmas", "ddim_sqrt_one_minus_alphas")
        return tuple(np.asarray(getattr(self, n)).astype(np.float32) for n in names)

    @staticmethod
    def _crossattn(c):
        if isinstance(c, dict):
            c = c["c_crossattn"]
        return c[0] if isinstance(c, (list, tuple)) else c

    @torch.no_grad()
    def sample(self, S, batch_size, shape, conditioning=None, callback=None, normals_sequence=None, img_callback=None,
               quantize_x0=False, eta=0., mask=None, x0=None, temperature=1., noise_dropout=0., score_corrector=None,
               corrector_kwargs=None, verbose=True, x_T=None, log_every_t=100, unconditional_guidance_scale=1.,
               unconditional_conditioning=None, **kwargs):
        """ddim.py:56-111 (same keywords)"""
        c = self._crossattn(conditioning) if conditioning is not None else None
        if c is not None and c.shape[0] != batch_size:
            print(f"Warning: Got {c.shape[0]} conditionings but batch-size is {batch_size}")
        self.make_schedule(ddim_num_steps=S, ddim_eta=eta, verbose=verbose)
        return self.ddim_sampling(conditioning, (batch_size,) + tuple(shape), x_T=x_T, callback=callback, img_callback=img_callback,
                                  quantize_denoised=quantize_x0, mask=mask, x0=x0, noise_dropout=noise_dropout,
                                  temperature=temperature, score_corrector=score_corrector, corrector_kwargs=corrector_kwargs,
                                  log_every_t=log_every_t, unconditional_guidance_scale=unconditional_guidance_scale,
                                  unconditional_conditioning=unconditional_conditioning)

    @torch.no_grad()
    def ddim_sampling(self, cond, shape, x_T=None, ddim_use_original_steps=False, callback=None, timesteps=None,
                      quantize_denoised=False, mask=None, x0=None, img_callback=None, log_every_t=100, temperature=1.,
                      noise_dropout=0., score_corrector=None, corrector_kwargs=None, unconditional_guidance_scale=1.,
                      unconditional_conditioning=None):
        """ddim.py:113-163"""
        if quantize_denoised:
            raise NotImplementedError("quantize_x0 needs a VQ first stage; the CelebBasis first stage is AutoencoderKL")
        if mask is not None:
            raise NotImplementedError("inpainting masks (mask / x0) are outside the CelebBasis path")
        device = self.model.device
        x = torch.randn(shape, device=device) if x_T is None else x_T.to(device).float()
        if ddim_use_original_steps:
            seq = np.arange(self.ddpm_num_timesteps if timesteps is None else timesteps)
        elif timesteps is None:
            seq = np.asarray(self.ddim_timesteps)
        else:   # truncated schedule (ddim.py:125-127)
            n_all = self.ddim_timesteps.shape[0]
            seq = np.asarray(self.ddim_timesteps[:int(min(timesteps / n_all, 1) * n_all) - 1])
        S = int(seq.shape[0])
        a, a_prev, sigma, s1m = (t[:S] for t in self._tables(ddim_use_original_steps))
        log = {"x_inter": [x], "pred_x0": [x]}
        needs_python_steps = any(v is not None for v in (callback, img_callback, mask, score_corrector)) or noise_dropout != 0.
        if not needs_python_steps:
            # whole loop on the device; the log then holds the start and the final state (with the default log_every_t = 100 > S
            # the reference logs just the first step and index 0)
            noise = torch.randn((S,) + tuple(shape), device=device) * temperature if np.any(sigma != 0) else None
            uc = None if unconditional_conditioning is None else self._crossattn(unconditional_conditioning).float().contiguous()
            unet = self.model.model.diffusion_model.native(device)
            c_ = self._crossattn(cond).float().contiguous()
            x, pred = unet.ddim_sample(x, c_, uc, seq, a, a_prev, sigma, s1m,
                                       scale=float(unconditional_guidance_scale), noise=noise, want_pred_x0=True)
            log["x_inter"].append(x); log["pred_x0"].append(pred)
            return x, log
        for i in range(S):
            index = S - 1 - i
            ts = torch.full((shape[0],), int(seq[index]), device=device, dtype=torch.long)
            x, pred = self.p_sample_ddim(x, cond, ts, index=index, use_original_steps=ddim_use_original_steps, temperature=temperature,
                                         noise_dropout=noise_dropout, score_corrector=score_corrector, corrector_kwargs=corrector_kwargs,
                                         unconditional_guidance_scale=unconditional_guidance_scale,
                                         unconditional_conditioning=unconditional_conditioning)
            if callback:
                callback(i)
            if img_callback:
                img_callback(pred, i)
            if index % log_every_t == 0 or index == S - 1:
                log["x_inter"].append(x); log["pred_x0"].append(pred)
        return x, log

    @torch.no_grad()
    def p_sample_ddim(self, x, c, t, index, repeat_noise=False, use_original_steps=False, quantize_denoised=False, temperature=1.,
                      noise_dropout=0., score_corrector=None, corrector_kwargs=None, unconditional_guidance_scale=1.,
                      unconditional_conditioning=None):
        """ddim.py:166-204: one step = native UNet pass (batch 2n under guidance) + the native update kernel"""
        scale = float(unconditional_guidance_scale)
        guided = unconditional_conditioning is not None and scale != 1.
        if guided:
            both = torch.cat([self._crossattn(unconditional_conditioning), self._crossattn(c)])
            eps = self.model.apply_model(torch.cat([x, x]), torch.cat([t, t]), both)          # rows: uncond first, then cond
        else:
            eps = self.model.apply_model(x, t, c)
        if score_corrector is not None:
            if guided:
                e_u, e_c = eps.chunk(2)
                eps, guided = e_u + scale * (e_c - e_u), False
            eps = score_corrector.modify_score(self.model, eps, x, t, c, **(corrector_kwargs or {}))
        a, a_prev, sigma, s1m = (float(tab[index]) for tab in self._tables(use_original_steps))
        noise = None
        if sigma != 0:
            shp = (1,) + tuple(x.shape[1:]) if repeat_noise else tuple(x.shape)
            noise = torch.randn(shp, device=x.device).expand(x.shape) * temperature
            if noise_dropout > 0.:
                noise = torch.nn.functional.dropout(noise, p=noise_dropout)
            noise = noise.float().contiguous()
        return _lib.ddim_update(context_for(x.device), x.float().contiguous().clone(), eps.contiguous(), a, a_prev, sigma, s1m,
                                scale=scale if guided else 1.0, noise=noise)
