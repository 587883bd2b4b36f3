"""Host-side mirror of ldm/models/autoencoder.py `AutoencoderKL` (reference :285-424), encoder half only: the CelebBasis
training step uses `encode` every step (ddpm.py:881-919); the decoder belongs to sampling (SURVEY §8 f, next)."""
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
        return S

    def load_state_dict(self, sd, strict=True, prefix=""):
        # decoder.* / post_quant_conv.* / loss.* keys of a full SD checkpoint are not part of the training step
        sd = {k: v for k, v in sd.items() if k.startswith(prefix + "encoder.") or k.startswith(prefix + "quant_conv.")}
        return super().load_state_dict(sd, strict=strict, prefix=prefix)

    def _build(self, ctx):
        return _lib.VAE(ctx, _lib.WeightTable(self.state_dict()), _lib.vae_cfg(**self.cfg))

    def encode(self, x):
        """x: NCHW (reference layout) or NHWC image batch in [-1,1]; returns the posterior"""
        if x.shape[1] == self.cfg["in_channels"] and x.shape[-1] != self.cfg["in_channels"]:
            x = x.permute(0, 2, 3, 1)
        return DiagonalGaussianPosterior(self, x.float().contiguous())
