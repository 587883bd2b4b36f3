"""ctypes binding of libcelebbasis_hip.so (include/celebbasis_hip.h).

PyTorch is plumbing here: it owns the device buffers and the stream that are handed to the C ABI as raw
pointers. The product path has NO fallback: if the shared library is missing or the device is not gfx950 the
calls raise.
"""
import ctypes as C
import os

import torch  # imported BEFORE the .so so that the process holds a single HIP runtime (torch's libamdhip64.so.7)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcelebbasis_hip.so")
_lib = None


# cb_dtype (include/celebbasis_hip.h): element type of a context's activations and frozen UNet / VAE / CLIP weights
DTYPE_F32, DTYPE_BF16 = 0, 1
_DTYPES = {"f32": DTYPE_F32, "fp32": DTYPE_F32, "float32": DTYPE_F32, torch.float32: DTYPE_F32, DTYPE_F32: DTYPE_F32,
           "bf16": DTYPE_BF16, "bfloat16": DTYPE_BF16, torch.bfloat16: DTYPE_BF16}


def dtype_code(dtype):
    """'f32' | 'bf16' | torch.float32 | torch.bfloat16 | cb_dtype int -> cb_dtype int"""
    if dtype is None:
        return DTYPE_F32
    if dtype == DTYPE_BF16 and not isinstance(dtype, bool):
        return DTYPE_BF16
    try:
        return _DTYPES[dtype]
    except (KeyError, TypeError):
        raise ValueError(f"unknown dtype {dtype!r}: use 'f32' or 'bf16'")


def torch_dtype(code):
    return torch.bfloat16 if code == DTYPE_BF16 else torch.float32


class CBError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libcelebbasis_hip error {code}: {msg}")
        self.code = code


# ---------------------------------------------------------------- C structs (mirror include/celebbasis_hip.h)
ABI_VERSION = 303        # CB_ABI_VERSION of include/celebbasis_hip.h these bindings mirror


class _Sized(C.Structure):
    """descriptor whose first member is `struct_size` (include/celebbasis_hip.h): filled on construction"""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.struct_size = C.sizeof(self)


class GemmDesc(_Sized):
    _fields_ = [("struct_size", C.c_int), ("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p),
                ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
                ("lda", C.c_int64), ("ldb", C.c_int64), ("ldc", C.c_int64),
                ("a_layout", C.c_int), ("b_layout", C.c_int), ("batch", C.c_int),
                ("strideA", C.c_int64), ("strideB", C.c_int64), ("strideC", C.c_int64),
                ("bias", C.c_void_p), ("R", C.c_void_p), ("ldr", C.c_int64), ("strideR", C.c_int64),
                ("alpha", C.c_float), ("beta", C.c_float), ("dtype", C.c_int)]


class AttentionDesc(_Sized):
    _fields_ = [("struct_size", C.c_int), ("qbuf", C.c_void_p), ("ldq", C.c_int64), ("q_off", C.c_int),
                ("kvbuf", C.c_void_p), ("ldkv", C.c_int64), ("k_off", C.c_int), ("v_off", C.c_int),
                ("B", C.c_int), ("Nq", C.c_int), ("Nk", C.c_int), ("C", C.c_int), ("heads", C.c_int),
                ("scale", C.c_float), ("causal", C.c_int),
                ("out", C.c_void_p), ("dout", C.c_void_p), ("dqbuf", C.c_void_p), ("dkvbuf", C.c_void_p), ("dtype", C.c_int)]


class UNetCfg(C.Structure):
    _fields_ = [("in_channels", C.c_int), ("out_channels", C.c_int), ("model_channels", C.c_int), ("num_res_blocks", C.c_int),
                ("channel_mult", C.c_int * 8), ("n_mult", C.c_int),
                ("attention_resolutions", C.c_int * 8), ("n_attn", C.c_int),
                ("num_heads", C.c_int), ("context_dim", C.c_int)]


class VaeCfg(C.Structure):
    _fields_ = [("ch", C.c_int), ("ch_mult", C.c_int * 8), ("n_mult", C.c_int), ("num_res_blocks", C.c_int),
                ("in_channels", C.c_int), ("z_channels", C.c_int)]


class ClipCfg(C.Structure):
    _fields_ = [("vocab", C.c_int), ("width", C.c_int), ("layers", C.c_int), ("heads", C.c_int), ("ctx_len", C.c_int),
                ("mlp", C.c_int)]


class TrainerCfg(_Sized):
    _fields_ = [("struct_size", C.c_int), ("unet", UNetCfg), ("vae", VaeCfg), ("clip", ClipCfg),
                ("fr_layers", C.c_int * 4), ("fr_width", C.c_int), ("fr_in", C.c_int),
                ("image_size", C.c_int), ("es", C.c_int), ("inner_dim", C.c_int), ("max_ids", C.c_int),
                ("momentum", C.c_float), ("scale_factor", C.c_float), ("linear_start", C.c_float), ("linear_end", C.c_float),
                ("timesteps", C.c_int), ("placeholder_token", C.c_int64),
                ("adam_b1", C.c_float), ("adam_b2", C.c_float), ("adam_eps", C.c_float), ("weight_decay", C.c_float),
                ("dtype", C.c_int), ("use_rm_mlp", C.c_int)]


class Batch(C.Structure):
    _fields_ = [("image", C.c_void_p), ("faces", C.c_void_p), ("ids", C.c_void_p), ("tokens", C.c_void_p), ("t", C.c_void_p),
                ("noise", C.c_void_p), ("post_noise", C.c_void_p), ("B", C.c_int)]


# every symbol include/celebbasis_hip.h declares (tests/test_abi.py checks the .so exports all of them)
ABI_SYMBOLS = [
    "cb_last_error", "cb_version", "cb_ctx_create", "cb_ctx_create_dtype", "cb_ctx_dtype", "cb_ctx_aux_stream", "cb_ctx_destroy", "cb_ctx_sync", "cb_ctx_stats", "cb_ctx_reset_stats",
    "cb_ctx_set_debug_taps", "cb_ctx_tap", "cb_ctx_profile_begin", "cb_ctx_profile_end", "cb_wtab_create", "cb_wtab_add", "cb_wtab_destroy", "cb_gemm", "cb_debug_glds_safe", "cb_debug_splitk_inlaunch", "cb_trainer_set_tail_overlap",
    "cb_pack_conv_weight", "cb_conv2d_fwd", "cb_conv2d_dgrad", "cb_groupnorm_silu_fwd", "cb_groupnorm_silu_bwd",
    "cb_layernorm_fwd", "cb_layernorm_bwd", "cb_geglu_fwd", "cb_geglu_bwd", "cb_attention", "cb_timestep_embedding",
    "cb_q_sample", "cb_mse_loss", "cb_ctx_set_warp_grid", "cb_face_warp_resize", "cb_row_surgery", "cb_adamw_step", "cb_nchw_to_nhwc",
    "cb_nhwc_to_nchw", "cb_unet_create", "cb_unet_destroy", "cb_unet_forward", "cb_unet_backward_context", "cb_vae_create",
    "cb_vae_destroy", "cb_vae_encode", "cb_clip_create", "cb_clip_destroy", "cb_clip_embed_tokens", "cb_clip_forward",
    "cb_clip_backward_embeds", "cb_frnet_create", "cb_frnet_destroy", "cb_frnet_forward", "cb_trainer_cfg_default",
    "cb_trainer_create", "cb_trainer_destroy", "cb_trainer_step", "cb_trainer_get_state", "cb_trainer_set_state",
    "cb_debug_force_plan", "cb_trainer_prefetch", "cb_trainer_set_graph_replay", "cb_trainer_graph_stats", "cb_trainer_prefetch_stream", "cb_trainer_set_prefetch_stream", "cb_trainer_loss_to", "cb_celeb_head_forward", "cb_face_augment", "cb_paste_on_canvas", "cb_row_surgery_multi", "cb_trainer_share_unet", "cb_trainer_share_clip", "cb_unet_ddim_sample", "cb_ddim_update", "cb_vae_decoder_create", "cb_vae_decoder_destroy", "cb_vae_decode",
]


_P, _I, _L, _F = C.c_void_p, C.c_int, C.c_int64, C.c_float
_ARGTYPES = {
    "cb_last_error": [], "cb_version": [],
    "cb_ctx_create": [_I, _P, _P], "cb_ctx_create_dtype": [_I, _P, _I, _P], "cb_ctx_dtype": [_P, _P], "cb_ctx_aux_stream": [_P, _P], "cb_ctx_destroy": [_P], "cb_ctx_sync": [_P], "cb_ctx_stats": [_P, _P, _P, _P, _P],
    "cb_ctx_reset_stats": [_P], "cb_ctx_set_debug_taps": [_P, _I], "cb_ctx_tap": [_P, C.c_char_p, _P, _L, _P],
    "cb_debug_glds_safe": [_I], "cb_debug_splitk_inlaunch": [_I], "cb_trainer_set_tail_overlap": [_P, _I], "cb_ctx_profile_begin": [_P], "cb_ctx_profile_end": [_P, C.c_char_p, _L],
    "cb_wtab_create": [_P], "cb_wtab_add": [_P, C.c_char_p, _P, _I, _I, _P], "cb_wtab_destroy": [_P],
    "cb_gemm": [_P, _P], "cb_pack_conv_weight": [_P, _P, _P, _I, _I, _I, _I],
    "cb_conv2d_fwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I],
    "cb_conv2d_dgrad": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F],
    "cb_groupnorm_silu_fwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _I],
    "cb_groupnorm_silu_bwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F],
    "cb_layernorm_fwd": [_P, _P, _P, _P, _P, _P, _L, _I, _F],
    "cb_layernorm_bwd": [_P, _P, _P, _P, _P, _P, _L, _I, _F],
    "cb_geglu_fwd": [_P, _P, _P, _L, _I], "cb_geglu_bwd": [_P, _P, _P, _P, _L, _I],
    "cb_attention": [_P, _P],
    "cb_timestep_embedding": [_P, _P, _P, _I, _I],
    "cb_q_sample": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I],
    "cb_mse_loss": [_P, _P, _P, _P, _P, _I, _I, _I],
    "cb_face_warp_resize": [_P, _P, _P, _P, _I, _I, _I, _I], "cb_ctx_set_warp_grid": [_P, _P, _I],
    "cb_row_surgery": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _L],
    "cb_adamw_step": [_P, _P, _P, _P, _P, _L, _P, _F, _F, _F, _F, _F],
    "cb_nchw_to_nhwc": [_P, _P, _P, _I, _I, _I, _I], "cb_nhwc_to_nchw": [_P, _P, _P, _I, _I, _I, _I],
    "cb_unet_create": [_P, _P, C.c_char_p, _P, _P], "cb_unet_destroy": [_P],
    "cb_unet_forward": [_P, _P, _P, _P, _I, _I, _I, _I, _P, _I], "cb_unet_backward_context": [_P, _P, _P],
    "cb_vae_create": [_P, _P, C.c_char_p, _P, _P], "cb_vae_destroy": [_P],
    "cb_vae_encode": [_P, _P, _P, _I, _I, _I, _F, _P, _P],
    "cb_clip_create": [_P, _P, C.c_char_p, _P, _P], "cb_clip_destroy": [_P],
    "cb_clip_embed_tokens": [_P, _P, _I, _I, _P], "cb_clip_forward": [_P, _P, _I, _I, _P, _I],
    "cb_clip_backward_embeds": [_P, _P, _P],
    "cb_frnet_create": [_P, _P, C.c_char_p, _P, _I, _I, _P], "cb_frnet_destroy": [_P], "cb_frnet_forward": [_P, _P, _I, _P],
    "cb_trainer_cfg_default": [_P], "cb_trainer_create": [_P, _P, _P, _P], "cb_trainer_destroy": [_P],
    "cb_trainer_step": [_P, _P, _F], "cb_trainer_set_graph_replay": [_P, _I], "cb_trainer_graph_stats": [_P, _P, _P], "cb_trainer_prefetch": [_P, _P], "cb_trainer_prefetch_stream": [_P, _P], "cb_debug_force_plan": [_I, _I, _I], "cb_trainer_set_prefetch_stream": [_P, _P], "cb_trainer_get_state": [_P, C.c_char_p, _P, _L, _P],
    "cb_trainer_set_state": [_P, C.c_char_p, _P, _L], "cb_trainer_loss_to": [_P, _P],
    "cb_row_surgery_multi": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _I],
    "cb_celeb_head_forward": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P],
    "cb_face_augment": [_P, _P, _P, _P, _I, _I], "cb_paste_on_canvas": [_P, _P, _P, _P, _I, _I, _I],
    "cb_trainer_share_unet": [_P, _P], "cb_trainer_share_clip": [_P, _P],
    "cb_unet_ddim_sample": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _F, _P, _P],
    "cb_ddim_update": [_P, _P, _P, _P, _I, _I, _I, _I, _F, _F, _F, _F, _F, _P],
    "cb_vae_decoder_create": [_P, _P, C.c_char_p, _P, _P], "cb_vae_decoder_destroy": [_P],
    "cb_vae_decode": [_P, _P, _I, _I, _I, _F, _I, _P],
}


def load():
    """dlopen the in-tree library. Raises if it has not been built (python -m celebbasis_amd.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: build it with `python -m celebbasis_amd.build` "
                           "(there is no CPU or PyTorch fallback for the CelebBasis hot path)")
    lib = C.CDLL(LIB_PATH)
    lib.cb_last_error.restype = C.c_char_p
    for name in ABI_SYMBOLS:
        fn = getattr(lib, name)
        if name != "cb_last_error":
            fn.restype = C.c_int
        if name in _ARGTYPES:
            fn.argtypes = _ARGTYPES[name]
    if lib.cb_version() != ABI_VERSION:
        raise RuntimeError(f"{LIB_PATH} implements ABI {lib.cb_version()}, these bindings were written for {ABI_VERSION}: rebuild "
                           "(python -m celebbasis_amd.build --force)")
    _lib = lib
    return lib


def check(code):
    if code != 0:
        raise CBError(code, load().cb_last_error().decode("utf-8", "replace"))


def ptr(t, dtype=torch.float32):
    """raw pointer of a contiguous tensor (or None); `dtype` may be a tuple of accepted types"""
    if t is None:
        return None
    if not isinstance(t, torch.Tensor):
        raise TypeError("expected a torch.Tensor")
    if (t.dtype not in dtype) if isinstance(dtype, tuple) else (t.dtype != dtype):
        raise TypeError(f"expected dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError("tensor must be contiguous")
    return C.c_void_p(t.data_ptr())


# meta_net.py:131-142 `trans_matrix` (the '/512' is the reference's own): FFHQ-aligned crop -> ArcFace-aligned crop
WARP_TRANS_MATRIX = [[1.07695457, -0.03625215, -1.56352194 / 512], [0.03625215, 1.07695457, -5.32134629 / 512]]


class Context:
    """cb_ctx wrapper. device=None -> current torch device + current stream; device=-1 -> dry context (no GPU).
    dtype: 'f32' (BASELINE config 2, the reference's arithmetic) or 'bf16' (config 3: bf16 storage + bf16 MFMA for the frozen
    UNet / VAE / CLIP; statistics, the coefficient head, its gradient and AdamW stay fp32)."""

    def __init__(self, device=None, dtype="f32"):
        lib = load()
        self.dtype = dtype_code(dtype)
        self.act_dtype = torch_dtype(self.dtype)      # torch dtype of op-level activation tensors on this context
        self.dry = device is not None and device < 0
        if self.dry:
            dev, stream = -1, None
        else:
            if not torch.cuda.is_available():
                raise RuntimeError("celebbasis_amd needs an MI355X (gfx950) GPU; none is visible. "
                                   "Use Context(device=-1) only for dry (no-compute) host-logic checks.")
            dev = torch.cuda.current_device() if device is None else device
            stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        self.device = dev
        self.h = C.c_void_p()
        check(lib.cb_ctx_create_dtype(C.c_int(dev), stream, C.c_int(self.dtype), C.byref(self.h)))

    def sync(self):
        check(load().cb_ctx_sync(self.h))

    def ensure_warp_grid(self, S):
        """Hand the library the face warp's sampling grid as a constant table (cb_ctx_set_warp_grid). The reference evaluates
        F.affine_grid(trans_matrix.repeat(B,1,1), size, align_corners=True) on every step (meta_net.py:253-257) — from a constant
        matrix (:131-142) at a constant size, so it is a table like the beta schedule: built here ONCE, on the host, by the very
        call the reference makes, and uploaded. The kernel then samples at the reference's own fp32 coordinates instead of
        re-deriving them in another evaluation order (two ulps of a 512-pixel coordinate = 3e-5 on a white-noise face)."""
        S = int(S)
        if getattr(self, "_warp_S", None) == S:
            return
        M = torch.tensor([WARP_TRANS_MATRIX], dtype=torch.float32)
        self.set_warp_grid(torch.nn.functional.affine_grid(M, size=[1, 3, S, S], align_corners=True)[0])

    def set_warp_grid(self, grid):
        """install a given [S, S, 2] fp32 sampling grid (host tensor). torch's CPU affine_grid is linspace + a K = 3 bmm whose rounding is
        HOST dependent (two x86 hosts with the same torch build differ in 14 % of the elements by one ulp,
        tools/probes/probe_affine_grid_host.py): a caller that must reproduce another host's run — the parity tests against oracle
        trajectories generated elsewhere — hands over that host's table."""
        grid = grid.detach().to(torch.float32).cpu().contiguous()
        S = int(grid.shape[0])
        assert tuple(grid.shape) == (S, S, 2), grid.shape
        check(load().cb_ctx_set_warp_grid(self.h, C.c_void_p(grid.data_ptr()), C.c_int(S)))
        self._warp_S = S
        self.warp_grid_host = grid

    def aux_stream(self):
        """the library's side stream as a torch stream (for Tensor.record_stream); None on a dry context"""
        if self.dry:
            return None
        p = C.c_void_p()
        check(load().cb_ctx_aux_stream(self.h, C.byref(p)))
        return torch.cuda.ExternalStream(p.value, device=torch.device("cuda", self.device)) if p.value else None

    def stats(self):
        f = C.c_double(); l = C.c_int64(); a = C.c_int64(); p = C.c_int64()
        check(load().cb_ctx_stats(self.h, C.byref(f), C.byref(l), C.byref(a), C.byref(p)))
        return {"gemm_flops": f.value, "launches": l.value, "arena_peak_bytes": a.value, "pool_bytes": p.value}

    def reset_stats(self):
        check(load().cb_ctx_reset_stats(self.h))

    def set_debug_taps(self, on=True):
        check(load().cb_ctx_set_debug_taps(self.h, C.c_int(1 if on else 0)))

    def tap(self, name):
        n = C.c_int64()
        check(load().cb_ctx_tap(self.h, name.encode(), None, C.c_int64(0), C.byref(n)))
        out = torch.empty(n.value, dtype=torch.float32)
        check(load().cb_ctx_tap(self.h, name.encode(), C.c_void_p(out.data_ptr()), C.c_int64(n.value), C.byref(n)))
        return out

    def profile_begin(self):
        check(load().cb_ctx_profile_begin(self.h))

    def profile_end(self):
        import json
        buf = C.create_string_buffer(1 << 16)
        check(load().cb_ctx_profile_end(self.h, buf, C.c_int64(len(buf))))
        return json.loads(buf.value.decode())

    def close(self):
        if getattr(self, "h", None) is not None and self.h.value:
            load().cb_ctx_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class WeightTable:
    """cb_wtab wrapper built from a {state-dict key: tensor} mapping. Shape-only entries (dry) pass torch.Size."""

    def __init__(self, state=None):
        self.h = C.c_void_p()
        self._keep = []
        check(load().cb_wtab_create(C.byref(self.h)))
        if state:
            for k, v in state.items():
                self.add(k, v)

    def add(self, name, value):
        if isinstance(value, torch.Tensor):
            if value.dtype != torch.float32:
                return  # integer buffers (position_ids, num_batches_tracked) are not used by the engine
            t = value.detach().contiguous()
            self._keep.append(t)
            dims = (C.c_int64 * max(1, t.dim()))(*t.shape)
            check(load().cb_wtab_add(self.h, name.encode(), C.c_void_p(t.data_ptr()), C.c_int(1 if t.is_cuda else 0),
                                     C.c_int(t.dim()), dims))
        else:  # shape only
            shape = tuple(value)
            dims = (C.c_int64 * max(1, len(shape)))(*shape)
            check(load().cb_wtab_add(self.h, name.encode(), None, C.c_int(0), C.c_int(len(shape)), dims))

    def close(self):
        if getattr(self, "h", None) is not None and self.h.value:
            load().cb_wtab_destroy(self.h)
            self.h = C.c_void_p()
        self._keep = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def unet_cfg(in_channels=4, out_channels=4, model_channels=320, num_res_blocks=2, channel_mult=(1, 2, 4, 4),
             attention_resolutions=(4, 2, 1), num_heads=8, context_dim=768):
    c = UNetCfg()
    c.in_channels, c.out_channels, c.model_channels, c.num_res_blocks = in_channels, out_channels, model_channels, num_res_blocks
    for i, v in enumerate(channel_mult):
        c.channel_mult[i] = int(v)
    c.n_mult = len(channel_mult)
    for i, v in enumerate(attention_resolutions):
        c.attention_resolutions[i] = int(v)
    c.n_attn = len(attention_resolutions)
    c.num_heads, c.context_dim = num_heads, context_dim
    return c


def vae_cfg(ch=128, ch_mult=(1, 2, 4, 4), num_res_blocks=2, in_channels=3, z_channels=4):
    c = VaeCfg()
    c.ch = ch
    for i, v in enumerate(ch_mult):
        c.ch_mult[i] = int(v)
    c.n_mult = len(ch_mult)
    c.num_res_blocks, c.in_channels, c.z_channels = num_res_blocks, in_channels, z_channels
    return c


def clip_cfg(vocab=49408, width=768, layers=12, heads=12, ctx_len=77, mlp=3072):
    c = ClipCfg()
    c.vocab, c.width, c.layers, c.heads, c.ctx_len, c.mlp = vocab, width, layers, heads, ctx_len, mlp
    return c


def trainer_cfg_default():
    c = TrainerCfg()
    check(load().cb_trainer_cfg_default(C.byref(c)))
    return c


class _Handle:
    _destroy = None

    def close(self):
        if getattr(self, "h", None) is not None and self.h.value:
            getattr(load(), self._destroy)(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class UNet(_Handle):
    _destroy = "cb_unet_destroy"

    def __init__(self, ctx, wtab, cfg, prefix=""):
        self.ctx, self.cfg = ctx, cfg
        self.h = C.c_void_p()
        check(load().cb_unet_create(ctx.h, wtab.h, prefix.encode(), C.byref(cfg), C.byref(self.h)))

    def forward(self, x, t, context, need_grad=False):
        """x [B,C,H,W] float32, t [B] int64, context [B,L,D] -> eps [B,Cout,H,W] (openaimodel.py:710-742)"""
        B, _, H, W = x.shape
        L = context.shape[1]
        out = torch.empty((B, self.cfg.out_channels, H, W), dtype=torch.float32, device=x.device)
        self._ctx_shape = context.shape
        dry = self.ctx.dry
        check(load().cb_unet_forward(self.h, None if dry else ptr(x), None if dry else ptr(t, torch.int64),
                                     None if dry else ptr(context), B, H, W, L, None if dry else ptr(out),
                                     C.c_int(1 if need_grad else 0)))
        return out

    def backward_context(self, deps):
        dc = torch.empty(self._ctx_shape, dtype=torch.float32, device=deps.device)
        dry = self.ctx.dry
        check(load().cb_unet_backward_context(self.h, None if dry else ptr(deps), None if dry else ptr(dc)))
        return dc

    def ddim_sample(self, x_T, cond, uncond, timesteps, alphas, alphas_prev, sigmas, sqrt_one_minus_alphas, scale=1.0,
                    noise=None, want_pred_x0=False):
        """the whole DDIM loop (ddim.py:113-204) on the device. x_T [n,C,H,W]; cond/uncond [n,L,D]; the five tables are the
        S-entry buffers of DDIMSampler.make_schedule (host, any array-like). Returns (x_0 sample, last pred_x0 | None)."""
        import numpy as np
        n, _, H, W = x_T.shape
        L = cond.shape[1]
        S = len(timesteps)
        ts = np.ascontiguousarray(np.asarray(timesteps, dtype=np.int64))
        tabs = [np.ascontiguousarray(np.asarray(torch.as_tensor(v).cpu() if isinstance(v, torch.Tensor) else v, dtype=np.float32))
                for v in (alphas, alphas_prev, sigmas, sqrt_one_minus_alphas)]
        assert all(t.shape == (S,) for t in tabs), "DDIM tables must have one entry per DDIM step"
        x = x_T.detach().to(torch.float32).contiguous().clone()
        p0 = torch.empty_like(x) if want_pred_x0 else None
        dry = self.ctx.dry
        hp = lambda a: C.c_void_p(a.ctypes.data)
        check(load().cb_unet_ddim_sample(self.h, None if dry else ptr(x), None if dry else ptr(cond),
                                         None if (dry or uncond is None) else ptr(uncond), n, H, W, L, S, hp(ts), hp(tabs[0]),
                                         hp(tabs[1]), hp(tabs[2]), hp(tabs[3]), C.c_float(scale),
                                         None if (dry or noise is None) else ptr(noise), None if (dry or p0 is None) else ptr(p0)))
        return x, p0


class VAE(_Handle):
    _destroy = "cb_vae_destroy"

    def __init__(self, ctx, wtab, cfg, prefix=""):
        self.ctx, self.cfg = ctx, cfg
        self.h = C.c_void_p()
        check(load().cb_vae_create(ctx.h, wtab.h, prefix.encode(), C.byref(cfg), C.byref(self.h)))

    def encode(self, image_hwc, post_noise=None, scale=0.18215):
        """image [B,H,W,3] -> (moments NHWC [B,H/8,W/8,2z], z NCHW [B,z,H/8,W/8])"""
        B, H, W, _ = image_hwc.shape
        f = 2 ** (self.cfg.n_mult - 1)
        zc = self.cfg.z_channels
        mom = torch.empty((B, H // f, W // f, 2 * zc), dtype=torch.float32, device=image_hwc.device)
        z = torch.empty((B, zc, H // f, W // f), dtype=torch.float32, device=image_hwc.device)
        dry = self.ctx.dry
        check(load().cb_vae_encode(self.h, None if dry else ptr(image_hwc), None if dry else ptr(post_noise), B, H, W,
                                   C.c_float(scale), None if dry else ptr(mom), None if dry else ptr(z)))
        return mom, z


class VAEDecoder(_Handle):
    _destroy = "cb_vae_decoder_destroy"

    def __init__(self, ctx, wtab, cfg, prefix=""):
        self.ctx, self.cfg = ctx, cfg
        self.h = C.c_void_p()
        check(load().cb_vae_decoder_create(ctx.h, wtab.h, prefix.encode(), C.byref(cfg), C.byref(self.h)))

    def decode(self, z, scale_factor=1.0, to01=False):
        """z [B,zc,h,w] NCHW -> image [B,3,8h,8w] NCHW: decode_first_stage (ddpm.py:761-819) = decode(z / scale_factor)"""
        B, _, h, w = z.shape
        f = 2 ** (self.cfg.n_mult - 1)
        out = torch.empty((B, self.cfg.in_channels, h * f, w * f), dtype=torch.float32, device=z.device)
        dry = self.ctx.dry
        # images are independent: decode in chunks so that every activation stays below 2 GiB (the 16-byte buffer-load
        # kernels address an operand with 32-bit offsets; beyond that the scalar loaders would take over)
        per = max(1, int((2 ** 31 - 2 ** 20) // max(1, h * f * w * f * self.cfg.ch * 2 * 4)))
        for s0 in range(0, B, per):
            n = min(per, B - s0)
            zc = z[s0:s0 + n].contiguous()
            check(load().cb_vae_decode(self.h, None if dry else ptr(zc), n, h, w, C.c_float(scale_factor),
                                       C.c_int(1 if to01 else 0), None if dry else C.c_void_p(out[s0:s0 + n].data_ptr())))
        return out


def ddim_update(ctx, x, eps, a_t, a_prev, sigma_t, sqrt_one_minus_at, scale=1.0, noise=None, want_pred_x0=True):
    """one p_sample_ddim update (ddim.py:181-199) in place on x [n,C,H,W]; eps is [n,...] or, with guidance, [2n,...] (uncond first)"""
    n, Cc, H, W = x.shape
    cfg_rows = 1 if eps.shape[0] == 2 * n else 0
    p0 = torch.empty_like(x) if want_pred_x0 else None
    check(load().cb_ddim_update(ctx.h, ptr(x), ptr(eps), None if noise is None else ptr(noise), n, Cc, H * W, cfg_rows,
                                C.c_float(scale), C.c_float(a_t), C.c_float(a_prev), C.c_float(sigma_t), C.c_float(sqrt_one_minus_at),
                                None if p0 is None else ptr(p0)))
    return x, p0


def face_augment(ctx, faces, params):
    """flip + colour jitter of HWC faces [B,S,S,3] in [-1,1]; params [B,5] = flip, brightness, contrast, saturation, hue (device)"""
    B, S = faces.shape[0], faces.shape[1]
    out = torch.empty_like(faces)
    check(load().cb_face_augment(ctx.h, ptr(faces), ptr(params), ptr(out), B, S))
    return out


def paste_on_canvas(ctx, img, geom):
    """`_add_bg`: rescale to (rh, rw) and paste at (pos_h, pos_w) on a -1 canvas; geom [B,4] int32 (device)"""
    B, S, _, Cc = img.shape
    out = torch.empty_like(img)
    check(load().cb_paste_on_canvas(ctx.h, ptr(img), ptr(geom, torch.int32), ptr(out), B, S, Cc))
    return out


class Clip(_Handle):
    _destroy = "cb_clip_destroy"

    def __init__(self, ctx, wtab, cfg, prefix=""):
        self.ctx, self.cfg = ctx, cfg
        self.h = C.c_void_p()
        check(load().cb_clip_create(ctx.h, wtab.h, prefix.encode(), C.byref(cfg), C.byref(self.h)))

    def embed_tokens(self, tokens):
        B, L = tokens.shape
        out = torch.empty((B, L, self.cfg.width), dtype=torch.float32, device=tokens.device)
        check(load().cb_clip_embed_tokens(self.h, ptr(tokens, torch.int64), B, L, ptr(out)))
        return out

    def forward(self, embedded, need_grad=False):
        B, L, _ = embedded.shape
        out = torch.empty_like(embedded)
        self._shape = embedded.shape
        dry = self.ctx.dry
        check(load().cb_clip_forward(self.h, None if dry else ptr(embedded), B, L, None if dry else ptr(out),
                                     C.c_int(1 if need_grad else 0)))
        return out

    def backward_embeds(self, dhidden):
        de = torch.empty(self._shape, dtype=torch.float32, device=dhidden.device)
        dry = self.ctx.dry
        check(load().cb_clip_backward_embeds(self.h, None if dry else ptr(dhidden), None if dry else ptr(de)))
        return de


class FrNet(_Handle):
    _destroy = "cb_frnet_destroy"

    def __init__(self, ctx, wtab, layers=(3, 13, 30, 3), width=64, in_size=112, prefix="", feat=512):
        self.ctx, self.in_size, self.feat = ctx, in_size, feat
        self.h = C.c_void_p()
        arr = (C.c_int * 4)(*layers)
        check(load().cb_frnet_create(ctx.h, wtab.h, prefix.encode(), arr, width, in_size, C.byref(self.h)))

    def forward(self, img_nhwc):
        N = img_nhwc.shape[0]
        out = torch.empty((N, self.feat), dtype=torch.float32, device=img_nhwc.device)
        dry = self.ctx.dry
        check(load().cb_frnet_forward(self.h, None if dry else ptr(img_nhwc), N, None if dry else ptr(out)))
        return out


class Trainer(_Handle):
    _destroy = "cb_trainer_destroy"

    def __init__(self, ctx, wtab, cfg=None):
        self.ctx = ctx
        self.cfg = cfg if cfg is not None else trainer_cfg_default()
        self.cfg.dtype = ctx.dtype          # the frozen networks follow the context's element type
        self.h = C.c_void_p()
        check(load().cb_trainer_create(ctx.h, wtab.h, C.byref(self.cfg), C.byref(self.h)))
        ctx.ensure_warp_grid(self.cfg.image_size)

    def step(self, image, faces, ids, tokens, t, noise, post_noise, lr):
        b = Batch()
        b.B = int(tokens.shape[0]) if tokens is not None else int(image)
        if not self.ctx.dry:
            b.image, b.faces = ptr(image), ptr(faces)
            b.ids, b.tokens, b.t = ptr(ids, torch.int64), ptr(tokens, torch.int64), ptr(t, torch.int64)
            b.noise, b.post_noise = ptr(noise), ptr(post_noise)
        check(load().cb_trainer_step(self.h, C.byref(b), C.c_float(lr)))

    def set_graph_replay(self, on=True):
        """whole-step HIP-graph replay (cb_trainer_set_graph_replay); opt-in (or CB_STEP_GRAPH=1)"""
        check(load().cb_trainer_set_graph_replay(self.h, C.c_int(1 if on else 0)))

    def set_tail_overlap(self, on=True):
        """deferred step tail (cb_trainer_set_tail_overlap): default on; off = the CLIP gradient chain / head backward / AdamW are
        joined into the context stream before `step` returns (A/B measurements, bitwise comparison)"""
        check(load().cb_trainer_set_tail_overlap(self.h, C.c_int(1 if on else 0)))

    def graph_stats(self):
        c, r = C.c_int64(), C.c_int64()
        check(load().cb_trainer_graph_stats(self.h, C.byref(c), C.byref(r)))
        return dict(captures=c.value, replays=r.value)

    def loss_tensor(self, device):
        """the last step's loss as a 0-dim DEVICE tensor, copied in stream order (no host sync)"""
        out = torch.empty((), dtype=torch.float32, device=device)
        check(load().cb_trainer_loss_to(self.h, None if self.ctx.dry else ptr(out)))
        return out

    def prefetch(self, image, faces, post_noise):
        """enqueue the first stage (VAE encode, FR backbone) of the NEXT batch so that it overlaps the step enqueued after this
        call; `step` on the same tensors then consumes it (identical results). Order: prefetch(batch[i+1]); step(batch[i])."""
        b = Batch()
        b.B = int(image.shape[0])
        if not self.ctx.dry:
            b.image, b.faces, b.post_noise = ptr(image), ptr(faces), ptr(post_noise)
            if getattr(self, "_pf_stream", None) is None:
                # the prefetch stream is TORCH's (never destroyed): the one call whose reads of caller memory are not ordered on
                # the context stream when it returns (celebbasis_hip.h), so the caching allocator is told about the second reader
                # HIGH priority (-1): measured on MI355X (profiles/README.md, round 3) — the bf16 bs=4 step runs 34.2 ms pipelined with
                # the prefetch stream at priority -1 against 37.8 / 38.0 ms at +1 / 0 (and 35.5 ms unpipelined): the big first-stage
                # convolutions have to get onto the CUs AHEAD of the step's many small kernels to overlap with them at all; the fp32
                # bs=1 step is indifferent (38.3 - 38.7 ms). CB_PF_PRIORITY overrides the value.
                lo = int(os.environ.get("CB_PF_PRIORITY", "-1"))
                try:
                    self._pf_stream = torch.cuda.Stream(device=torch.device("cuda", self.ctx.device), priority=lo)
                except Exception:
                    self._pf_stream = torch.cuda.Stream(device=torch.device("cuda", self.ctx.device))
                check(load().cb_trainer_set_prefetch_stream(self.h, C.c_void_p(self._pf_stream.cuda_stream)))
        check(load().cb_trainer_prefetch(self.h, C.byref(b)))
        if not self.ctx.dry:
            for t_ in (image, faces, post_noise):
                t_.record_stream(self._pf_stream)

    def unet(self):
        """UNet view sharing the trainer's device weights (sampling after personalisation without a second copy)"""
        u = UNet.__new__(UNet)
        u.ctx, u.cfg, u.h = self.ctx, self.cfg.unet, C.c_void_p()
        check(load().cb_trainer_share_unet(self.h, C.byref(u.h)))
        return u

    def clip(self):
        c = Clip.__new__(Clip)
        c.ctx, c.cfg, c.h = self.ctx, self.cfg.clip, C.c_void_p()
        check(load().cb_trainer_share_clip(self.h, C.byref(c.h)))
        return c

    def get_state(self, name):
        n = C.c_int64()
        check(load().cb_trainer_get_state(self.h, name.encode(), None, C.c_int64(0), C.byref(n)))
        out = torch.zeros(n.value, dtype=torch.float32)
        check(load().cb_trainer_get_state(self.h, name.encode(), C.c_void_p(out.data_ptr()), C.c_int64(n.value), C.byref(n)))
        return out

    def set_state(self, name, value):
        v = value.detach().to(torch.float32).contiguous().cpu().reshape(-1)
        check(load().cb_trainer_set_state(self.h, name.encode(), C.c_void_p(v.data_ptr()), C.c_int64(v.numel())))
