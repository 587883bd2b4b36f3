"""Host-side mirror of ldm/modules/id_embedding/iresnet.py (reference :67-235): frozen, eval-mode face-recognition
backbone; compute in libcelebbasis_hip.so (cb_frnet_*)."""
from .... import lib as _lib
from ..frozen import FrozenModule


class IResNet(FrozenModule):
    def __init__(self, layers=(3, 13, 30, 3), num_features=512, width=64, in_size=112):
        self.layers, self.num_features, self.width, self.in_size = tuple(layers), num_features, width, in_size
        super().__init__(self.param_shapes(self.layers, num_features, width, in_size))

    @staticmethod
    def param_shapes(layers, feat=512, width=64, in_size=112):
        S = {}

        def bn(p, c):
            for k in ("weight", "bias", "running_mean", "running_var"):
                S[f"{p}.{k}"] = (c,)

        S["conv1.weight"] = (width, 3, 3, 3); bn("bn1", width); S["prelu.weight"] = (width,)
        cin = width
        for stage, n in enumerate(layers):
            planes = width << stage
            for i in range(n):
                p = f"layer{stage + 1}.{i}"
                bn(p + ".bn1", cin); S[p + ".conv1.weight"] = (planes, cin, 3, 3); bn(p + ".bn2", planes)
                S[p + ".prelu.weight"] = (planes,); S[p + ".conv2.weight"] = (planes, planes, 3, 3); bn(p + ".bn3", planes)
                if i == 0:
                    S[p + ".downsample.0.weight"] = (planes, cin, 1, 1); bn(p + ".downsample.1", planes)
                cin = planes
        bn("bn2", width * 8)
        side = in_size // 16
        S["fc.weight"] = (feat, width * 8 * side * side); S["fc.bias"] = (feat,)
        bn("features", feat)
        return S

    def load_state_dict(self, sd, strict=True, prefix=""):
        sd = {k: v for k, v in sd.items() if "num_batches_tracked" not in k}
        return super().load_state_dict(sd, strict=strict, prefix=prefix)

    def _build(self, ctx):
        return _lib.FrNet(ctx, _lib.WeightTable(self.state_dict()), self.layers, self.width, self.in_size, feat=self.num_features)

    def forward(self, x):
        """x: [N,3,S,S] NCHW (reference) -> [N,512]"""
        return self.native(x.device).forward(x.permute(0, 2, 3, 1).float().contiguous())

    __call__ = forward


def iresnet100(pretrained=False, progress=True, **kwargs):
    return IResNet((3, 13, 30, 3), **kwargs)
