"""GPU parity tests of the bf16 kernels (BASELINE config 3), op level, THROUGH THE C ABI on a bf16 context.

Reference = the same op in torch fp32 on the CPU, evaluated on the bf16-ROUNDED inputs (so the comparison isolates the
kernel: fp32 accumulation of bf16 operands + one bf16 rounding of the output). Tolerance (stated per test):
max|hip - ref| <= tol * max|ref| with tol = 1e-2, i.e. ~2.5 bf16 ulps (2^-8 = 3.9e-3) of the output's largest magnitude —
one rounding of the result plus the roundings of bf16 intermediates (probabilities, dS) inside the fused kernels."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from test_gpu_ops import MODES, close, rnd, torch_attention, torch_conv

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
TOL = 1e-2


def _dev():
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def cbf(built_lib):
    from celebbasis_amd import lib
    ctx = lib.Context(dtype="bf16")
    yield lib, ctx
    ctx.sync()


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def r16(t):
    """round to bf16 and back: the values the kernel actually sees"""
    return t.to(BF).float()


def d16(t):
    return t.to(BF).contiguous().to(_dev())


def d32(t):
    return t.float().contiguous().to(_dev())


# ------------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("M,N,K,batch,extra", [
    (256, 192, 320, 1, "bias"),
    (77, 40, 96, 3, "none"),            # ragged M/N, small K
    (4096, 320, 320, 1, "bias+res"),
    (64, 1280, 11520, 1, "bias"),       # split-K path (deep 8x8 conv shape)
    (300, 77, 40, 8, "alpha"),          # QK^T-like: K = 40 (one ragged k-tile)
    (16384, 2560, 320, 1, "bias"),      # GEGLU projection at bs=4
    (1, 1280, 1280, 1, "bias+beta"),
])
def test_gemm_bf16(cbf, M, N, K, batch, extra):
    lib, ctx = cbf
    A = r16(rnd(batch, M, K, seed=1)); B = r16(rnd(batch, N, K, seed=2))
    bias = rnd(N, seed=3) if "bias" in extra else None
    R = r16(rnd(batch, M, N, seed=4)) if "res" in extra else None
    C0 = r16(rnd(batch, M, N, seed=5))
    alpha = 0.37 if "alpha" in extra else 1.0
    beta = 1.0 if "beta" in extra else 0.0
    ref = alpha * torch.matmul(A, B.transpose(1, 2))
    if bias is not None:
        ref = ref + bias
    if R is not None:
        ref = ref + R
    ref = ref + beta * C0
    Ad, Bd, Cd = d16(A), d16(B), d16(C0)
    d = lib.GemmDesc()
    d.A, d.B, d.C = Ad.data_ptr(), Bd.data_ptr(), Cd.data_ptr()
    d.M, d.N, d.K = M, N, K
    d.lda = d.ldb = K; d.ldc = N
    d.a_layout = d.b_layout = 0; d.batch = batch
    d.strideA, d.strideB, d.strideC = M * K, N * K, M * N
    if bias is not None:
        bd = d32(bias); d.bias = bd.data_ptr()
    if R is not None:
        Rd = d16(R); d.R = Rd.data_ptr(); d.ldr = N; d.strideR = M * N
    d.alpha, d.beta, d.dtype = alpha, beta, lib.DTYPE_BF16
    lib.check(lib.load().cb_gemm(ctx.h, C.byref(d)))
    ctx.sync()
    close(f"bf16.gemm[{M}x{N}x{K},batch{batch},{extra}]", Cd, ref, TOL)


@pytest.fixture
def tile256(cbf):
    """pin the planner to the 8-wave 256x128 LDS-DMA tile for one test (cb_debug_force_plan is process-global)"""
    lib, _ = cbf
    L = lib.load()

    def pin(split):
        lib.check(L.cb_debug_force_plan(256, 128, split))
    yield pin
    lib.check(L.cb_debug_force_plan(0, 0, 0))


@pytest.mark.parametrize("M,N,K,split,extra", [(1024, 256, 640, 1, "bias+res"), (777, 200, 328, 1, "bias"), (512, 320, 2880, 3, "bias+beta")])
def test_gemm_bf16_tile256(cbf, tile256, M, N, K, split, extra):
    """the 256x128 tile (three-stage ring, 512 threads): full tiles, ragged M / N / K tails, split-K + reduce"""
    tile256(split)
    test_gemm_bf16(cbf, M, N, K, 1, extra)


@pytest.mark.parametrize("N,H,W,Cin,Cout,ks,mode", [(2, 32, 32, 64, 160, 3, "same"), (1, 24, 20, 72, 96, 3, "same"), (2, 33, 31, 64, 128, 3, "down"),
                                                     (2, 16, 16, 64, 128, 3, "up")])
def test_conv_bf16_tile256(cbf, tile256, N, H, W, Cin, Cout, ks, mode):
    """implicit-GEMM gathers (fast path, general path, stride 2, fused upsample) and their input gradients on the 256x128 tile"""
    tile256(1)
    test_conv_bf16_fwd_dgrad(cbf, N, H, W, Cin, Cout, ks, mode, "bias")


# ---- the phase-pipelined 256 x 256 tile (cb_gemm_bf16_p8.hip): 8 waves in two groups one barrier apart, four phases per k-tile,
# LDS-DMA half-tiles in flight across the barriers (one counted vmcnt(4) per k-tile). Pinned with cb_debug_force_plan(256, 256, split);
# a problem the tile cannot run (K % 64, general conv gathers) silently keeps the planner's choice, so the tests assert the kernel name.
def _p8_ran(ctx, fn):
    ctx.profile_begin()
    try:
        fn()
    finally:
        ran = [r["kernel"] for r in ctx.profile_end()]
    return ran


@pytest.mark.parametrize("M,N,K,split,extra", [
    (1024, 512, 640, 1, "bias+res"),        # full tiles, 10 k-tiles
    (777, 300, 320, 1, "bias"),             # ragged M / N: zero-filled rows, masked stores
    (512, 320, 2880, 3, "bias+beta"),       # split-K (15 k-tiles per split) + in-launch reduction
    (300, 200, 64, 1, "none"),              # ONE k-tile: prologue + drained tail only
    (256, 256, 128, 1, "alpha"),            # two k-tiles: both ring buffers once
    (4096, 1280, 1280, 1, "bias"),          # 16 x 5 tiles, 20 k-tiles (the 32x32-level Linear at bs = 4)
    (130, 88, 192, 2, "res"),               # split-K with an odd number of k-tiles per split (3 -> 2 + 1)
])
@pytest.mark.parametrize("tile", [(256, 256), (512, 128)])
def test_gemm_bf16_tile_p8(cbf, tile, M, N, K, split, extra):
    lib, ctx = cbf
    L = lib.load()
    lib.check(L.cb_debug_force_plan(tile[0], tile[1], split))
    try:
        ran = _p8_ran(ctx, lambda: test_gemm_bf16(cbf, M, N, K, 1, extra))
    finally:
        lib.check(L.cb_debug_force_plan(0, 0, 0))
    assert any(k.startswith(f"gemm_bf16_p8_kernel<bf16,{tile[0]},{tile[1]},ROW>") for k in ran), ran


@pytest.mark.parametrize("N,H,W,Cin,Cout,ks,mode", [(2, 32, 32, 64, 160, 3, "same"), (2, 33, 31, 64, 128, 3, "down"), (1, 16, 16, 128, 256, 1, "same"),
                                                     (4, 64, 64, 320, 320, 3, "same"), (1, 24, 20, 192, 96, 3, "same")])
@pytest.mark.parametrize("tile", [(256, 256), (512, 128)])
def test_conv_bf16_tile_p8(cbf, tile, N, H, W, Cin, Cout, ks, mode):
    """implicit-GEMM gathers on the phase-pipelined tile: the scalar (tap, channel) walk with per-lane validity re-derived at a tap change,
    stride 2, 1x1, three k-tiles per tap; forward and — where the gather is of the plain kind — the input gradient"""
    lib, ctx = cbf
    L = lib.load()
    lib.check(L.cb_debug_force_plan(tile[0], tile[1], 1))
    try:
        ran = _p8_ran(ctx, lambda: test_conv_bf16_fwd_dgrad(cbf, N, H, W, Cin, Cout, ks, mode, "bias"))
    finally:
        lib.check(L.cb_debug_force_plan(0, 0, 0))
    assert any(k.startswith(f"gemm_bf16_p8_kernel<bf16,{tile[0]},{tile[1]},CONVF>") for k in ran), ran


@pytest.mark.parametrize("tile", [(256, 256), (512, 128)])
def test_p8_counted_wait_matches_safe_mode_and_repeats(cbf, tile):
    """cb_debug_glds_safe(1) drains the DMA queue (vmcnt(0)) where the pipeline waits for vmcnt(4): a half-tile read before it landed, or
    re-staged before its last reader was done, shows up as a difference — BIT FOR BIT equality on step-sized GEMMs and convolutions
    (one / several waves of blocks, split-K, ragged tails), and 8 back-to-back repeats of the counted pipeline must all agree (race screen)."""
    lib, ctx = cbf
    L = lib.load()
    shapes = [(4096, 320, 2880, 1), (1024, 640, 5760, 2), (16384, 320, 320, 1), (1024, 1280, 11520, 4), (777, 300, 320, 1), (16384, 1280, 1280, 1)]
    outs = {}
    try:
        for safe, reps in ((1, 1), (0, 8)):
            lib.check(L.cb_debug_glds_safe(safe))
            for (M, N, K, split) in shapes:
                lib.check(L.cb_debug_force_plan(tile[0], tile[1], split))
                Ad, Bd = d16(r16(rnd(1, M, K, seed=M + 1))), d16(r16(rnd(1, N, K, seed=N + 2)))
                for r in range(reps):
                    Cd = torch.zeros(M, N, device=Ad.device, dtype=Ad.dtype)
                    d = lib.GemmDesc()
                    d.A, d.B, d.C = Ad.data_ptr(), Bd.data_ptr(), Cd.data_ptr()
                    d.M, d.N, d.K = M, N, K
                    d.lda = d.ldb = K; d.ldc = N; d.batch = 1; d.alpha = 1.0; d.dtype = lib.DTYPE_BF16
                    lib.check(L.cb_gemm(ctx.h, C.byref(d)))
                    outs[(safe, M, N, K, r)] = Cd
                ctx.sync()
    finally:
        lib.check(L.cb_debug_glds_safe(0)); lib.check(L.cb_debug_force_plan(0, 0, 0))
    for (M, N, K, split) in shapes:
        ref = outs[(1, M, N, K, 0)].cpu()
        assert torch.isfinite(ref.float()).all() and ref.float().abs().max() > 0
        for r in range(8):
            got = outs[(0, M, N, K, r)].cpu()
            assert torch.equal(got, ref), f"tile {tile} {M}x{N}x{K}/{split} repeat {r}: counted-wait and drained pipelines differ " \
                                          f"(max {(got.float() - ref.float()).abs().max().item():.3e}, {(got != ref).sum().item()} elements)"


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
def test_glds_counted_wait_matches_safe_mode(cbf, dtype):
    """the LDS-DMA kernel's tile ring is synchronised by a hand-counted `s_waitcnt vmcnt(N)` + raw `s_barrier`
    (cb_gemm_glds.hip); cb_debug_glds_safe(1) replaces it by vmcnt(0) + a full barrier. On step-sized problems (one wave of
    blocks, several waves, split-K, ragged tails; both element types, i.e. the 2-stage rings of every tile shape the planner
    picks) the two pipelines must agree BIT FOR BIT — a tile read before it landed would show up here"""
    lib, ctx = cbf
    L = lib.load()
    bf = dtype == "bf16"
    shapes = [(4096, 320, 2880), (1024, 640, 5760), (16384, 320, 320), (256, 1280, 11520), (777, 200, 328), (64, 1280, 2560)]
    outs = {}
    for safe in (0, 1):
        lib.check(L.cb_debug_glds_safe(safe))
        try:
            for (M, N, K) in shapes:
                A = rnd(1, M, K, seed=M + 1); B = rnd(1, N, K, seed=N + 2)
                Ad, Bd = (d16(r16(A)), d16(r16(B))) if bf else (d32(A), d32(B))
                Cd = torch.zeros(M, N, device=Ad.device, dtype=Ad.dtype)
                d = lib.GemmDesc()
                d.A, d.B, d.C = Ad.data_ptr(), Bd.data_ptr(), Cd.data_ptr()
                d.M, d.N, d.K = M, N, K
                d.lda = d.ldb = K; d.ldc = N; d.batch = 1; d.alpha = 1.0
                d.dtype = lib.DTYPE_BF16 if bf else lib.DTYPE_F32
                lib.check(L.cb_gemm(ctx.h, C.byref(d)))
                ctx.sync()
                outs[(safe, M, N, K)] = Cd.clone()
        finally:
            lib.check(L.cb_debug_glds_safe(0))
    for (M, N, K) in shapes:
        a, b = outs[(0, M, N, K)], outs[(1, M, N, K)]
        assert torch.isfinite(a.float()).all() and a.float().abs().max() > 0
        assert torch.equal(a, b), f"{dtype} {M}x{N}x{K}: counted-wait and safe pipelines differ (max {(a.float() - b.float()).abs().max().item():.3e})"


def test_descriptor_struct_size_guard(cbf):
    """a caller compiled against another header revision passes a descriptor of another size: rejected, not read"""
    lib, ctx = cbf
    d = lib.GemmDesc()
    assert d.struct_size == C.sizeof(lib.GemmDesc) and lib.load().cb_version() == lib.ABI_VERSION == 303
    d.struct_size -= 8
    assert lib.load().cb_gemm(ctx.h, C.byref(d)) != 0 and b"struct_size" in lib.load().cb_last_error()
    ad = lib.AttentionDesc(); ad.struct_size = 0
    assert lib.load().cb_attention(ctx.h, C.byref(ad)) != 0 and b"struct_size" in lib.load().cb_last_error()


def test_gemm_bf16_rejects_unsupported_layouts(cbf):
    """the bf16 GEMM takes k-contiguous, 16-byte-chunked operands only: anything else is an error, never a silent fallback"""
    lib, ctx = cbf
    A = d16(rnd(64, 64)); B = d16(rnd(64, 64)); Cd = d16(rnd(64, 64))
    d = lib.GemmDesc()
    d.A, d.B, d.C = A.data_ptr(), B.data_ptr(), Cd.data_ptr()
    d.M = d.N = d.K = 64; d.lda = d.ldb = d.ldc = 64; d.batch = 1; d.alpha = 1.0; d.dtype = lib.DTYPE_BF16
    d.b_layout = 1
    assert lib.load().cb_gemm(ctx.h, C.byref(d)) != 0 and b"k-contiguous" in lib.load().cb_last_error()
    d.b_layout = 0; d.K = 60
    assert lib.load().cb_gemm(ctx.h, C.byref(d)) != 0 and b"multiple of 8" in lib.load().cb_last_error()


# ------------------------------------------------------------------------------------------------ conv
@pytest.mark.parametrize("N,H,W,Cin,Cout,ks,mode,extra", [
    (1, 16, 16, 64, 96, 3, "same", "bias+rowbias+res"),     # fast gather (C % 32 == 0)
    (2, 16, 16, 72, 40, 3, "same", "bias"),                 # general gather (C % 8 == 0 only)
    (2, 17, 15, 32, 64, 3, "down", "bias"),
    (1, 16, 16, 32, 32, 3, "down_asym", "bias"),
    (2, 8, 8, 64, 32, 3, "up", "bias"),
    (1, 12, 12, 64, 48, 1, "same", "bias"),
    (2, 16, 16, 3, 32, 3, "same", "bias"),                  # RGB input: channel axis padded 3 -> 8
    (1, 16, 16, 64, 4, 3, "same", "bias"),                  # latent output: dgrad pads 4 -> 8
    (4, 64, 64, 320, 320, 3, "same", "bias"),               # UNet level-0 ResBlock conv at bs=4
])
def test_conv_bf16_fwd_dgrad(cbf, N, H, W, Cin, Cout, ks, mode, extra):
    lib, ctx = cbf
    L = lib.load()
    x = r16(rnd(N, Cin, H, W, seed=1)); w = r16(rnd(Cout, Cin, ks, ks, seed=2, scale=(Cin * ks * ks) ** -0.5)); b = rnd(Cout, seed=3)
    xr = x.clone().requires_grad_(True)
    y_ref = torch_conv(xr, w, b, mode)
    Ho, Wo = y_ref.shape[2], y_ref.shape[3]
    rb = r16(rnd(N, Cout, seed=4)) if "rowbias" in extra else None
    rs = r16(rnd(N, Cout, Ho, Wo, seed=5)) if "res" in extra else None
    ref = y_ref + (rb[:, :, None, None] if rb is not None else 0) + (rs if rs is not None else 0)
    cin_p, cout_p = (Cin + 7) // 8 * 8, (Cout + 7) // 8 * 8
    wp = torch.empty(Cout * ks * ks * cin_p, dtype=BF, device=_dev())
    wpd = torch.empty(Cin * ks * ks * cout_p, dtype=BF, device=_dev())
    wd = d32(w)
    lib.check(L.cb_pack_conv_weight(ctx.h, lib.ptr(wd), lib.ptr(wp, BF), Cout, Cin, ks, 0))
    lib.check(L.cb_pack_conv_weight(ctx.h, lib.ptr(wd), lib.ptr(wpd, BF), Cout, Cin, ks, 1))
    xd = d16(nhwc(x)); bd = d32(b)
    rbd = d16(rb) if rb is not None else None
    rsd = d16(nhwc(rs)) if rs is not None else None
    y = torch.empty(N, Ho, Wo, Cout, dtype=BF, device=_dev())
    lib.check(L.cb_conv2d_fwd(ctx.h, lib.ptr(xd, BF), lib.ptr(wp, BF), lib.ptr(bd), lib.ptr(rbd, BF), lib.ptr(rsd, BF), lib.ptr(y, BF),
                              N, H, W, Cin, Cout, ks, MODES[mode]))
    ctx.sync()
    tag = f"bf16.conv[{N}x{H}x{W},{Cin}->{Cout},k{ks},{mode}]"
    close(tag + ".fwd", y.float().permute(0, 3, 1, 2), ref, TOL)
    if mode in ("same", "down", "up"):
        dy = r16(rnd(N, Cout, Ho, Wo, seed=6))
        (dx_ref,) = torch.autograd.grad((y_ref * dy).sum(), xr)
        dx0 = r16(rnd(N, H, W, Cin, seed=7))
        for beta in (0.0, 1.0):
            dx = d16(dx0)
            dyd = d16(nhwc(dy))
            lib.check(L.cb_conv2d_dgrad(ctx.h, lib.ptr(dyd, BF), lib.ptr(wpd, BF), lib.ptr(dx, BF), N, H, W, Cin, Cout, ks, MODES[mode], beta))
            ctx.sync()
            close(tag + f".dgrad[beta{beta}]", dx.float().permute(0, 3, 1, 2), dx_ref + beta * dx0.permute(0, 3, 1, 2), TOL)
        if mode == "up":       # the combined 4x4 stride-2 gradient conv (dgrad = 2 pack)
            wpu = torch.empty(Cin * 16 * cout_p, dtype=BF, device=_dev())
            lib.check(L.cb_pack_conv_weight(ctx.h, lib.ptr(wd), lib.ptr(wpu, BF), Cout, Cin, ks, 2))
            dx = d16(dx0)
            dyd = d16(nhwc(dy))
            lib.check(L.cb_conv2d_dgrad(ctx.h, lib.ptr(dyd, BF), lib.ptr(wpu, BF), lib.ptr(dx, BF), N, H, W, Cin, Cout, ks, 5, 1.0))
            ctx.sync()
            close(tag + ".dgrad4x4[beta1.0]", dx.float().permute(0, 3, 1, 2), dx_ref + dx0.permute(0, 3, 1, 2), TOL)


# ------------------------------------------------------------------------------------------------ norms / GEGLU
@pytest.mark.parametrize("N,HW,C,silu,eps", [(2, 256, 320, 1, 1e-5), (1, 1024, 64, 0, 1e-6), (4, 64, 1280, 1, 1e-5), (1, 4096, 128, 1, 1e-6),
                                             (1, 131072, 128, 1, 1e-6)])     # both GroupNorm forms (one launch per group below 48 KB per slab, three chip-wide kernels above)
def test_groupnorm_bf16(cbf, N, HW, C, silu, eps):
    lib, ctx = cbf
    L = lib.load()
    x = r16(rnd(N, HW, C, seed=1) * 2 + 0.5); g = rnd(C, seed=2) * 0.3 + 1; b = rnd(C, seed=3) * 0.2; dy = r16(rnd(N, HW, C, seed=4))
    xr = x.clone().requires_grad_(True)
    t = F.group_norm(xr.permute(0, 2, 1), 32, g, b, eps).permute(0, 2, 1)
    ref = F.silu(t) if silu else t
    (dx_ref,) = torch.autograd.grad((ref * dy).sum(), xr)
    xd, dyd, gd, bd = d16(x), d16(dy), d32(g), d32(b)
    y = torch.empty_like(xd); dx = torch.empty_like(xd); stats = torch.empty(N, 32, 2, device=_dev())
    lib.check(L.cb_groupnorm_silu_fwd(ctx.h, lib.ptr(xd, BF), lib.ptr(gd), lib.ptr(bd), lib.ptr(y, BF), lib.ptr(stats), N, HW, C, eps, silu))
    lib.check(L.cb_groupnorm_silu_bwd(ctx.h, lib.ptr(xd, BF), lib.ptr(dyd, BF), lib.ptr(gd), lib.ptr(bd), lib.ptr(stats), lib.ptr(dx, BF), N, HW, C, silu, 0.0))
    ctx.sync()
    tag = f"bf16.groupnorm[{N}x{HW}x{C},silu{silu}]"
    close(tag + ".fwd", y, ref, TOL)
    close(tag + ".bwd", dx, dx_ref, TOL)
    # statistics are fp32 in both modes
    xg = x.reshape(N, HW, 32, C // 32).permute(0, 2, 1, 3).reshape(N, 32, -1)
    close(tag + ".mean", stats[..., 0], xg.mean(-1), 1e-5)


@pytest.mark.parametrize("rows,C", [(4096, 320), (77, 768), (256, 1280)])
def test_layernorm_bf16(cbf, rows, C):
    lib, ctx = cbf
    L = lib.load()
    x = r16(rnd(rows, C, seed=1) * 1.5 + 0.3); g = rnd(C, seed=2) * 0.3 + 1; b = rnd(C, seed=3) * 0.2; dy = r16(rnd(rows, C, seed=4))
    xr = x.clone().requires_grad_(True)
    ref = F.layer_norm(xr, (C,), g, b, 1e-5)
    (dx_ref,) = torch.autograd.grad((ref * dy).sum(), xr)
    xd, dyd, gd, bd = d16(x), d16(dy), d32(g), d32(b)
    y = torch.empty_like(xd); dx = torch.empty_like(xd); stats = torch.empty(rows, 2, device=_dev())
    lib.check(L.cb_layernorm_fwd(ctx.h, lib.ptr(xd, BF), lib.ptr(gd), lib.ptr(bd), lib.ptr(y, BF), lib.ptr(stats), rows, C, 1e-5))
    lib.check(L.cb_layernorm_bwd(ctx.h, lib.ptr(xd, BF), lib.ptr(dyd, BF), lib.ptr(gd), lib.ptr(stats), lib.ptr(dx, BF), rows, C, 0.0))
    ctx.sync()
    close(f"bf16.layernorm[{rows}x{C}].fwd", y, ref, TOL)
    close(f"bf16.layernorm[{rows}x{C}].bwd", dx, dx_ref, TOL)


def test_geglu_bf16(cbf):
    lib, ctx = cbf
    L = lib.load()
    M, D = 300, 640
    x = r16(rnd(M, 2 * D, seed=1)); dy = r16(rnd(M, D, seed=2))
    xr = x.clone().requires_grad_(True)
    a, gate = xr.chunk(2, dim=-1)
    ref = a * F.gelu(gate)
    (dx_ref,) = torch.autograd.grad((ref * dy).sum(), xr)
    xd, dyd = d16(x), d16(dy)
    y = torch.empty(M, D, dtype=BF, device=_dev()); dx = torch.empty_like(xd)
    lib.check(L.cb_geglu_fwd(ctx.h, lib.ptr(xd, BF), lib.ptr(y, BF), M, D))
    lib.check(L.cb_geglu_bwd(ctx.h, lib.ptr(xd, BF), lib.ptr(dyd, BF), lib.ptr(dx, BF), M, D))
    ctx.sync()
    close("bf16.geglu.fwd", y, ref, TOL)
    close("bf16.geglu.bwd", dx, dx_ref, TOL)


# ------------------------------------------------------------------------------------------------ attention
@pytest.mark.parametrize("B,Nq,Nk,Cc,heads,causal,fused", [
    (2, 256, 256, 320, 8, 0, 1),     # self-attention d=40 (padded to 64 in the QK^T contraction), fused QKV buffer
    (1, 1024, 77, 640, 8, 0, 0),     # cross-attention, Nk=77 (ragged key tile, query-split dK/dV), d=80
    (2, 77, 77, 768, 12, 1, 1),      # CLIP causal, d=64
    (1, 64, 77, 1280, 8, 0, 0),      # d=160
    (1, 200, 136, 256, 2, 0, 0),     # d=128 (KS=4), ragged both ways
    (2, 100, 100, 64, 4, 1, 1),      # d=16 causal (mini configs)
    (4, 1024, 1024, 640, 8, 0, 1),   # UNet level-1 self-attention at bs=4
    # second-generation kernels (cb_flash_bf16_v2.hip: LDS-DMA ring, 16*QB rows per wave, pre-transposed second operands), 32 < d <= 80
    # (the (4, 1024, 1024, 640, 8) case above is the UNet's 32x32 level on them too):
    (1, 4096, 4096, 320, 8, 0, 1),   # the shape that carries the attention FLOPs: 64x64 level, d = 40 (QB = 1 at 8 heads)
    (4, 1024, 1024, 320, 8, 0, 1),   # d = 40, 32 heads: QB = 2 forward / dQ
    (8, 2048, 2048, 320, 8, 0, 1),   # d = 40, 64 heads: QB = 4 forward / dQ, QB = 2 dK/dV
    (1, 1000, 520, 512, 8, 0, 0),    # d = 64 (four output tiles), ragged query / key counts, separate q and kv buffers
    (2, 264, 1288, 384, 8, 0, 0),    # d = 48, key tail of 8
    (1, 520, 1000, 576, 8, 0, 0),    # d = 72: three 32-wide k groups (192-byte tile rows), ones row, ragged both ways
    (2, 1032, 264, 640, 8, 0, 0),    # d = 80 (five output tiles, no spare row), ragged query tile, separate q / kv buffers
])
def test_attention_bf16(cbf, B, Nq, Nk, Cc, heads, causal, fused):
    lib, ctx = cbf
    L = lib.load()
    d = Cc // heads
    scale = d ** -0.5
    q = r16(rnd(B, Nq, Cc, seed=1)); k = r16(rnd(B, Nk, Cc, seed=2)); v = r16(rnd(B, Nk, Cc, seed=3)); do = r16(rnd(B, Nq, Cc, seed=4))
    qr, kr, vr = (t.clone().requires_grad_(True) for t in (q, k, v))
    oref = torch_attention(qr, kr, vr, heads, scale, causal)
    dq, dk, dv = torch.autograd.grad((oref * do).sum(), [qr, kr, vr])
    desc = lib.AttentionDesc()
    nan = float("nan")
    if fused:
        qkv = d16(torch.cat([q, k, v], dim=-1))
        dqkv = torch.full_like(qkv, nan)
        desc.qbuf = desc.kvbuf = qkv.data_ptr(); desc.ldq = desc.ldkv = 3 * Cc
        desc.q_off, desc.k_off, desc.v_off = 0, Cc, 2 * Cc
        desc.dqbuf = desc.dkvbuf = dqkv.data_ptr()
    else:
        qd = d16(q); kv = d16(torch.cat([k, v], dim=-1))
        dqd = torch.full_like(qd, nan); dkv = torch.full_like(kv, nan)
        desc.qbuf = qd.data_ptr(); desc.ldq = Cc; desc.q_off = 0
        desc.kvbuf = kv.data_ptr(); desc.ldkv = 2 * Cc; desc.k_off, desc.v_off = 0, Cc
        desc.dqbuf = dqd.data_ptr(); desc.dkvbuf = dkv.data_ptr()
    desc.B, desc.Nq, desc.Nk, desc.C, desc.heads = B, Nq, Nk, Cc, heads
    desc.scale, desc.causal, desc.dtype = scale, causal, lib.DTYPE_BF16
    out = torch.empty(B, Nq, Cc, dtype=BF, device=_dev()); dod = d16(do)
    desc.out = out.data_ptr(); desc.dout = dod.data_ptr()
    lib.check(L.cb_attention(ctx.h, C.byref(desc)))
    ctx.sync()
    tag = f"bf16.attention[B{B},{Nq}x{Nk},C{Cc},h{heads},causal{causal}]"
    close(tag + ".out", out, oref, TOL)
    if fused:
        gq, gk, gv = dqkv[..., :Cc], dqkv[..., Cc:2 * Cc], dqkv[..., 2 * Cc:]
    else:
        gq, gk, gv = dqd, dkv[..., :Cc], dkv[..., Cc:]
    # the backward recomputes P from bf16 q, k and the fp32 lse, and contracts bf16-rounded P / dS: two more roundings
    close(tag + ".dq", gq, dq, 2 * TOL)
    close(tag + ".dk", gk, dk, 2 * TOL)
    close(tag + ".dv", gv, dv, 2 * TOL)


def test_attention_bf16_materialised_d512(cbf):
    """VAE AttnBlock (model.py:176-202): one head of 512 -> materialised scores, V transposed for the k-contiguous GEMM"""
    lib, ctx = cbf
    B, N, Cc = 1, 304, 512
    q = r16(rnd(B, N, Cc, seed=1)); k = r16(rnd(B, N, Cc, seed=2)); v = r16(rnd(B, N, Cc, seed=3))
    ref = torch_attention(q, k, v, 1, Cc ** -0.5, 0)
    qkv = d16(torch.cat([q, k, v], dim=-1))
    desc = lib.AttentionDesc()
    desc.qbuf = desc.kvbuf = qkv.data_ptr(); desc.ldq = desc.ldkv = 3 * Cc
    desc.q_off, desc.k_off, desc.v_off = 0, Cc, 2 * Cc
    desc.B, desc.Nq, desc.Nk, desc.C, desc.heads = B, N, N, Cc, 1
    desc.scale, desc.causal, desc.dtype = Cc ** -0.5, 0, lib.DTYPE_BF16
    out = torch.empty(B, N, Cc, dtype=BF, device=_dev())
    desc.out = out.data_ptr()
    lib.check(lib.load().cb_attention(ctx.h, C.byref(desc)))
    ctx.sync()
    close("bf16.attention.materialised[d512]", out, ref, 2 * TOL)
