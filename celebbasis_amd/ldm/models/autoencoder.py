"""Host-side mirror of ldm/models/autoencoder.py `AutoencoderKL` (reference :285-424): `encode` is used by every training
step (ddpm.py:881-919), `decode` by sampling (ddpm.py:761-819, stable_txt2img.py:342). The two halves are separate native
objects (cb_vae / cb_vae_dec); a checkpoint that only carries one half still serves that half."""
import torch

from ... import lib as _lib
from ..modules.frozen import FrozenModule


class DiagonalGaussianPosterior:
    """what AutoencoderKL.encode returns (distributions.py:24-62): holds moments; sample() draws on the device"""

    def __init__(self, owner, image_hwc):
        self._owner, self._image = owner, image_hwc

    def sample(self, noise=None, scale=1.0):
        b, h, w, _ = self._image.shape
        f = 2 ** (len(self._owner.cfg["ch_mult"]) - 1)
        if noise is None:
            noise = torch.randn(b, self._owner.cfg["z_channels"], h // f, w // f, device=self._image.device)
        _, z = self._owner.native(self._image.device).encode(self._image, noise.float().contiguous(), scale)
        return z

    def mode(self, scale=1.0):
        _, z = self._owner.native(self._image.device).encode(self._image, None, scale)
        return z


class AutoencoderKL(FrozenModule):
    def __init__(self, ddconfig, lossconfig=None, embed_dim=4, ckpt_path=None, ignore_keys=(), image_key="image",
                 colorize_nlabels=None, monitor=None):
        assert ddconfig["double_z"] and not ddconfig.get("attn_resolutions"), "SD v1-4 first stage only"
        assert embed_dim == ddconfig["z_channels"]
        self.cfg = dict(ch=ddconfig["ch"], ch_mult=tuple(int(v) for v in ddconfig["ch_mult"]),
                        num_res_blocks=ddconfig["num_res_blocks"], in_channels=ddconfig["in_channels"],
                        z_channels=ddconfig["z_channels"])
        self.image_key = image_key
        super().__init__(self.param_shapes(self.cfg))

    @staticmethod
    def param_shapes(cfg):
        S = {}

        def conv(p, ci, co, k):
            S[p + ".weight"] = (co, ci, k, k); S[p + ".bias"] = (co,)

        def affine(p, c):
            S[p + ".weight"] = (c,); S[p + ".bias"] = (c,)

        def res(p, ci, co):
            affine(p + ".norm1", ci); conv(p + ".conv1", ci, co, 3); affine(p + ".norm2", co); conv(p + ".conv2", co, co, 3)
            if ci != co:
                conv(p + ".nin_shortcut", ci, co, 1)

        conv("encoder.conv_in", cfg["in_channels"], cfg["ch"], 3)
        width = cfg["ch"]
        for level, mult in enumerate(cfg["ch_mult"]):
            for b in range(cfg["num_res_blocks"]):
                res(f"encoder.down.{level}.block.{b}", width, cfg["ch"] * mult); width = cfg["ch"] * mult
            if level < len(cfg["ch_mult"]) - 1:
                conv(f"encoder.down.{level}.downsample.conv", width, width, 3)
        res("encoder.mid.block_1", width, width)
        affine("encoder.mid.attn_1.norm", width)
        for name in ("q", "k", "v", "proj_out"):
            conv(f"encoder.mid.attn_1.{name}", width, width, 1)
        res("encoder.mid.block_2", width, width)
        affine("encoder.norm_out", width); conv("encoder.conv_out", width, 2 * cfg["z_channels"], 3)
        conv("quant_conv", 2 * cfg["z_channels"], 2 * cfg["z_channels"], 1)
        # decoder half (model.py:462-568) + post_quant_conv (autoencoder.py:303)
        conv("post_quant_conv", cfg["z_channels"], cfg["z_channels"], 1)
        width = cfg["ch"] * cfg["ch_mult"][-1]
        conv("decoder.conv_in", cfg["z_channels"], width, 3)
        res("decoder.mid.block_1", width, width)
        affine("decoder.mid.attn_1.norm", width)
        for name in ("q", "k", "v", "proj_out"):
            conv(f"decoder.mid.attn_1.{name}", width, width, 1)
        res("decoder.mid.block_2", width, width)
        for level in reversed(range(len(cfg["ch_mult"]))):
            out = cfg["ch"] * cfg["ch_mult"][level]
            for b in range(cfg["num_res_blocks"] + 1):
                res(f"decoder.up.{level}.block.{b}", width, out); width = out
            if level != 0:
                conv(f"decoder.up.{level}.upsample.conv", width, width, 3)
        affine("decoder.norm_out", width); conv("decoder.conv_out", width, cfg["in_channels"], 3)
        return S

    ENC_KEYS = ("encoder.", "quant_conv.")
    DEC_KEYS = ("decoder.", "post_quant_conv.")

    def load_state_dict(self, sd, strict=True, prefix=""):
        # loss.* (LPIPS / discriminator) keys of a full first-stage checkpoint never run
        keep = tuple(prefix + k for k in self.ENC_KEYS + self.DEC_KEYS)
        sd = {k: v for k, v in sd.items() if k.startswith(keep)}
        self._dec = None
        return super().load_state_dict(sd, strict=strict, prefix=prefix)

    def _build(self, ctx):
        return _lib.VAE(ctx, _lib.WeightTable(self.state_dict(only=self.ENC_KEYS)), _lib.vae_cfg(**self.cfg))

    def native_decoder(self, device):
        if getattr(self, "_dec", None) is None:
            from ..modules.frozen import context_for
            self._dec = _lib.VAEDecoder(context_for(device), _lib.WeightTable(self.state_dict(only=self.DEC_KEYS)),
                                        _lib.vae_cfg(**self.cfg))
        return self._dec

    def decode(self, z, scale_factor=1.0, to01=False):
        """autoencoder.py:330-333: post_quant_conv -> Decoder; z NCHW [B,zc,h,w] -> image NCHW [B,3,8h,8w]"""
        return self.native_decoder(z.device).decode(z.float().contiguous(), scale_factor, to01)

    def encode(self, x):
        """x: NCHW (reference layout) or NHWC image batch in [-1,1]; returns the posterior"""
        if x.shape[1] == self.cfg["in_channels"] and x.shape[-1] != self.cfg["in_channels"]:
            x = x.permute(0, 2, 3, 1)
        return DiagonalGaussianPosterior(self, x.float().contiguous())
