"""Host-side mirror of ldm/modules/diffusionmodules/openaimodel.py `UNetModel` (reference :413-742): same constructor
keywords (aigc_id.yaml:39-54) and state-dict keys; the arithmetic runs in libcelebbasis_hip.so (cb_unet_*)."""
import torch

from .... import lib as _lib
from ..frozen import FrozenModule


class UNetModel(FrozenModule):
    def __init__(self, image_size=32, in_channels=4, model_channels=320, out_channels=4, num_res_blocks=2,
                 attention_resolutions=(4, 2, 1), dropout=0, channel_mult=(1, 2, 4, 4), conv_resample=True, dims=2, num_classes=None,
                 use_checkpoint=False, use_fp16=False, num_heads=8, num_head_channels=-1, num_heads_upsample=-1,
                 use_scale_shift_norm=False, resblock_updown=False, use_new_attention_order=False, use_spatial_transformer=True,
                 transformer_depth=1, context_dim=768, n_embed=None, legacy=False):
        unsupported = dict(dropout=(dropout, 0), dims=(dims, 2), num_classes=(num_classes, None), use_fp16=(use_fp16, False),
                           use_scale_shift_norm=(use_scale_shift_norm, False), resblock_updown=(resblock_updown, False),
                           use_spatial_transformer=(use_spatial_transformer, True), transformer_depth=(transformer_depth, 1),
                           n_embed=(n_embed, None), legacy=(legacy, False), conv_resample=(conv_resample, True))
        for k, (got, want) in unsupported.items():
            if got != want:
                raise NotImplementedError(f"UNetModel({k}={got!r}) is outside the CelebBasis hot path (supported: {want!r})")
        if isinstance(context_dim, (list, tuple)):
            context_dim = context_dim[0]
        self.cfg = dict(in_channels=in_channels, out_channels=out_channels, model_channels=model_channels,
                        num_res_blocks=num_res_blocks, channel_mult=tuple(int(v) for v in channel_mult),
                        attention_resolutions=tuple(int(v) for v in attention_resolutions), num_heads=num_heads,
                        context_dim=int(context_dim))
        super().__init__(self.param_shapes(self.cfg))
        self.in_channels, self.out_channels, self.model_channels = in_channels, out_channels, model_channels

    @staticmethod
    def param_shapes(cfg):
        mc, ted, cd = cfg["model_channels"], 4 * cfg["model_channels"], cfg["context_dim"]
        S = {}
        P = lambda name, *shape: S.__setitem__(name, tuple(shape))

        def conv(p, ci, co, k):
            P(p + ".weight", co, ci, k, k); P(p + ".bias", co)

        def affine(p, c):
            P(p + ".weight", c); P(p + ".bias", c)

        def linear(p, ci, co, bias=True):
            P(p + ".weight", co, ci)
            if bias:
                P(p + ".bias", co)

        def res(p, ci, co):
            affine(p + ".in_layers.0", ci); conv(p + ".in_layers.2", ci, co, 3); linear(p + ".emb_layers.1", ted, co)
            affine(p + ".out_layers.0", co); conv(p + ".out_layers.3", co, co, 3)
            if ci != co:
                conv(p + ".skip_connection", ci, co, 1)

        def xf(p, ch):
            affine(p + ".norm", ch); conv(p + ".proj_in", ch, ch, 1); conv(p + ".proj_out", ch, ch, 1)
            t = p + ".transformer_blocks.0"
            for name, kdim in (("attn1", ch), ("attn2", cd)):
                linear(f"{t}.{name}.to_q", ch, ch, False); linear(f"{t}.{name}.to_k", kdim, ch, False)
                linear(f"{t}.{name}.to_v", kdim, ch, False); linear(f"{t}.{name}.to_out.0", ch, ch)
            linear(t + ".ff.net.0.proj", ch, 8 * ch); linear(t + ".ff.net.2", 4 * ch, ch)
            for n in (1, 2, 3):
                affine(f"{t}.norm{n}", ch)

        linear("time_embed.0", mc, ted); linear("time_embed.2", ted, ted)
        conv("input_blocks.0.0", cfg["in_channels"], mc, 3)
        widths, ch, ds, i = [mc], mc, 1, 1
        mults = cfg["channel_mult"]
        for level, mult in enumerate(mults):
            for _ in range(cfg["num_res_blocks"]):
                res(f"input_blocks.{i}.0", ch, mult * mc); ch = mult * mc
                if ds in cfg["attention_resolutions"]:
                    xf(f"input_blocks.{i}.1", ch)
                widths.append(ch); i += 1
            if level < len(mults) - 1:
                conv(f"input_blocks.{i}.0.op", ch, ch, 3); widths.append(ch); ds *= 2; i += 1
        res("middle_block.0", ch, ch); xf("middle_block.1", ch); res("middle_block.2", ch, ch)
        o = 0
        for level in reversed(range(len(mults))):
            for j in range(cfg["num_res_blocks"] + 1):
                res(f"output_blocks.{o}.0", ch + widths.pop(), mc * mults[level]); ch = mc * mults[level]
                has_xf = ds in cfg["attention_resolutions"]
                if has_xf:
                    xf(f"output_blocks.{o}.1", ch)
                if level > 0 and j == cfg["num_res_blocks"]:
                    conv(f"output_blocks.{o}.{2 if has_xf else 1}.conv", ch, ch, 3); ds //= 2
                o += 1
        affine("out.0", ch); conv("out.2", mc, cfg["out_channels"], 3)
        return S

    # ---- native execution
    def _build(self, ctx):
        return _lib.UNet(ctx, _lib.WeightTable(self.state_dict()), _lib.unet_cfg(**self.cfg))

    def forward(self, x, timesteps=None, context=None, y=None, **kwargs):
        """UNetModel.forward(x, timesteps, context) (reference :710-742); x NCHW, returns eps NCHW"""
        assert y is None, "class-conditional UNet is outside the CelebBasis hot path"
        h = self.native(x.device)
        return h.forward(x.float().contiguous(), timesteps.to(torch.int64).contiguous(), context.float().contiguous(),
                         need_grad=bool(kwargs.get("need_grad", False)))

    __call__ = forward

    def backward_context(self, deps):
        return self.native(deps.device).backward_context(deps.float().contiguous())
