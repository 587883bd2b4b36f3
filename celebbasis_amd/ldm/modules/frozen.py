"""Base of the host-side mirror classes: a frozen parameter container with the reference's state-dict keys whose
compute lives in libcelebbasis_hip.so. It deliberately has no torch-op forward: without the native library it raises."""
import torch

from ... import lib as _lib

_CTX = {}
_PRECISION = "f32"


def set_precision(precision):
    """element type of the frozen UNet / VAE / CLIP built from now on: 'f32' (the reference's arithmetic, BASELINE config 2) or
    'bf16' (config 3). The coefficient head, its gradient and AdamW are fp32 either way; public tensors stay fp32."""
    global _PRECISION
    _lib.dtype_code(precision)      # validates
    _PRECISION = precision


def get_precision():
    return _PRECISION


def context_for(device):
    """one cb_ctx per (device, stream, precision) the caller is on"""
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError("celebbasis_amd runs on an MI355X (gfx950) only: move inputs to the GPU "
                           "(there is no CPU / PyTorch-eager fallback for the hot path)")
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    key = (idx, torch.cuda.current_stream(idx).cuda_stream, _PRECISION)
    if key not in _CTX:
        with torch.cuda.device(idx):
            _CTX[key] = _lib.Context(idx, dtype=_PRECISION)
    return _CTX[key]


class FrozenModule:
    """parameters are plain tensors keyed by the reference state-dict names; all frozen (ddpm.py:491-500)"""

    def __init__(self, shapes):
        self._shapes = dict(shapes)
        self._params = {}
        self._native = None
        self.training = False

    # -- torch.nn.Module-like surface used by the reference's callers
    def state_dict(self, prefix="", only=None):
        """`only`: tuple of key prefixes restricting the dict to one half of the module (e.g. the VAE encoder)"""
        keys = [k for k in self._shapes if only is None or k.startswith(tuple(only))]
        missing = [k for k in keys if k not in self._params]
        if missing:
            raise RuntimeError(f"{type(self).__name__}: {len(missing)} parameters were never loaded (e.g. {missing[0]})")
        return {prefix + k: self._params[k] for k in keys}

    def load_state_dict(self, sd, strict=True, prefix=""):
        unexpected, loaded = [], 0
        for k, v in sd.items():
            if not k.startswith(prefix):
                continue
            name = k[len(prefix):]
            if name not in self._shapes:
                unexpected.append(k)
                continue
            if tuple(v.shape) != tuple(self._shapes[name]):
                raise RuntimeError(f"size mismatch for {k}: checkpoint {tuple(v.shape)} vs model {self._shapes[name]}")
            self._params[name] = v.detach().to(torch.float32).contiguous()
            loaded += 1
        missing = [prefix + k for k in self._shapes if k not in self._params]
        if strict and (missing or unexpected):
            raise RuntimeError(f"load_state_dict: missing {missing[:3]}… unexpected {unexpected[:3]}…")
        self._native = None
        return missing, unexpected

    def parameters(self):
        return iter(())          # nothing here is trainable

    def eval(self):
        self.training = False
        return self

    def train(self, mode=True):
        self.training = mode
        return self

    def to(self, *a, **k):
        return self

    def native(self, device):
        if self._native is None:
            self._native = self._build(context_for(device))
        return self._native

    def _build(self, ctx):
        raise NotImplementedError
