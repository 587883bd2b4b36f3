"""GPU parity tests, op level: every HIP kernel family called THROUGH THE C ABI (celebbasis_amd.lib -> libcelebbasis_hip.so)
against plain torch fp32 ops evaluated on the CPU. Tolerance: fp32 round-off class, stated per test as
max|hip - ref| <= tol * max|ref| (the reductions differ in order from torch's, nothing else)."""
import ctypes as C
import json
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

REPORT = {}


def _dev():
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def cb(built_lib):
    from celebbasis_amd import lib
    ctx = lib.Context()
    yield lib, ctx
    ctx.sync()
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/ops_report.json", "w") as f:
        json.dump(REPORT, f, indent=1, sort_keys=True)


def close(name, got, ref, tol):
    got = got.detach().float().cpu(); ref = ref.detach().float().cpu()
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    assert torch.isfinite(got).all(), f"{name}: non-finite output"
    err = (got - ref).abs().max().item()
    scale = max(ref.abs().max().item(), 1e-30)
    REPORT[name] = dict(max_abs_err=err, ref_max=scale, rel=err / scale, tol=tol)
    assert err <= tol * scale, f"{name}: max abs err {err:.3e} vs ref max {scale:.3e} (rel {err / scale:.3e} > {tol})"


def dv(t):
    """device copy on the context's stream. Temporaries may be dropped right after a native call: every entry point orders its
    reads on the context stream (cb_trainer_prefetch is the exception, handled by record_stream in lib.Trainer.prefetch)"""
    return t.contiguous().to(_dev())


class _Hold(list):
    """device temporaries whose raw pointer goes straight into a native call: `lib.ptr(dv(x))` alone would drop the tensor before
    the call is even made (a Python lifetime, not a stream-order question). Cleared by the `cb` fixture's sync at module end."""

    def __call__(self, t, dtype=torch.float32):
        from celebbasis_amd import lib
        t = dv(t)
        self.append(t)
        if len(self) > 64:
            del self[:32]
        return lib.ptr(t, dtype)


HOLD = _Hold()


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


# ------------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("M,N,K,al,bl,batch,extra", [
    (256, 192, 320, 0, 0, 1, "bias"),
    (77, 40, 96, 0, 0, 3, "none"),          # ragged M/N, small K
    (4096, 320, 320, 0, 0, 1, "bias+res"),
    (64, 1280, 11520, 0, 0, 1, "bias"),     # split-K path (deep 8x8 conv shape)
    (300, 77, 40, 0, 0, 8, "alpha"),        # QK^T-like: K=40 (not a multiple of 16)
    (300, 40, 77, 0, 1, 8, "none"),         # PV-like: B is [K][N]
    (77, 40, 300, 1, 1, 8, "beta"),         # dK/dV-like: both operands K-strided, K=300
    (130, 66, 50, 1, 0, 2, "none"),
    (1, 1280, 1280, 0, 0, 1, "bias"),       # single row through the GEMM kernel
])
def test_gemm(cb, M, N, K, al, bl, batch, extra):
    lib, ctx = cb
    A = rnd(batch, M, K, seed=1); B = rnd(batch, K, N, seed=2)
    bias = rnd(N, seed=3) if "bias" in extra else None
    R = rnd(batch, M, N, seed=4) if "res" in extra else None
    C0 = rnd(batch, M, N, seed=5)
    alpha = 0.37 if "alpha" in extra else 1.0
    beta = 1.0 if "beta" in extra else 0.0
    ref = alpha * torch.matmul(A, B)
    if bias is not None:
        ref = ref + bias
    if R is not None:
        ref = ref + R
    ref = ref + beta * C0
    # memory layouts: ROW = [rows][K] (K contiguous); COL = [K][rows]
    Ad = (A if al == 0 else A.transpose(1, 2)).contiguous().to(_dev())
    Bd = (B.transpose(1, 2) if bl == 0 else B).contiguous().to(_dev())
    Cd = C0.clone().to(_dev())
    d = lib.GemmDesc()
    d.A, d.B, d.C = Ad.data_ptr(), Bd.data_ptr(), Cd.data_ptr()
    d.M, d.N, d.K = M, N, K
    d.lda = K if al == 0 else M
    d.ldb = K if bl == 0 else N
    d.ldc = N
    d.a_layout, d.b_layout, d.batch = al, bl, batch
    d.strideA, d.strideB, d.strideC = M * K, K * N, M * N
    if bias is not None:
        bd = bias.to(_dev()); d.bias = bd.data_ptr()
    if R is not None:
        Rd = R.to(_dev()); d.R = Rd.data_ptr(); d.ldr = N; d.strideR = M * N
    d.alpha, d.beta = alpha, beta
    lib.check(lib.load().cb_gemm(ctx.h, C.byref(d)))
    ctx.sync()
    close(f"gemm[{M}x{N}x{K},a{al}b{bl},batch{batch},{extra}]", Cd, ref, 2e-5)


# ------------------------------------------------------------------------------------------------ conv
MODES = {"same": 0, "down": 1, "down_asym": 2, "up": 3, "s2_1x1": 4}


def torch_conv(x_nchw, w, b, mode):
    if mode == "same":
        return F.conv2d(x_nchw, w, b, 1, w.shape[-1] // 2)
    if mode == "down":
        return F.conv2d(x_nchw, w, b, 2, 1)
    if mode == "down_asym":
        return F.conv2d(F.pad(x_nchw, (0, 1, 0, 1)), w, b, 2, 0)
    if mode == "up":
        return F.conv2d(F.interpolate(x_nchw, scale_factor=2, mode="nearest"), w, b, 1, 1)
    if mode == "s2_1x1":
        return F.conv2d(x_nchw, w, b, 2, 0)
    raise ValueError(mode)


@pytest.mark.parametrize("N,H,W,Cin,Cout,ks,mode,extra", [
    (2, 16, 16, 32, 64, 3, "same", "rowbias+res"),
    (1, 64, 64, 320, 320, 3, "same", "none"),       # level-0 ResBlock conv
    (1, 8, 8, 1280, 1280, 3, "same", "none"),       # level-3 conv: split-K
    (2, 12, 10, 4, 32, 3, "same", "none"),          # conv_in: Cin = 4
    (2, 12, 10, 3, 16, 3, "same", "none"),          # VAE / FR conv_in: Cin = 3 (scalar gather path)
    (2, 12, 10, 32, 4, 3, "same", "none"),          # UNet out: Cout = 4
    (2, 16, 16, 32, 48, 1, "same", "none"),         # 1x1
    (2, 16, 16, 32, 32, 3, "down", "none"),
    (2, 16, 16, 32, 32, 3, "down_asym", "none"),
    (2, 8, 8, 32, 32, 3, "up", "none"),
    (1, 9, 7, 36, 20, 3, "up", "none"),             # odd sizes, channels % 32 != 0: the general gather of the 4x4 gradient conv
    (2, 14, 14, 32, 64, 1, "s2_1x1", "none"),
])
def test_conv_fwd_dgrad(cb, N, H, W, Cin, Cout, ks, mode, extra):
    lib, ctx = cb
    L = lib.load()
    x = rnd(N, Cin, H, W, seed=1); w = rnd(Cout, Cin, ks, ks, seed=2, scale=(Cin * ks * ks) ** -0.5); b = rnd(Cout, seed=3)
    xr = x.clone().requires_grad_(True)
    ref = torch_conv(xr, w, b, mode)
    Ho, Wo = ref.shape[2], ref.shape[3]
    rowbias = rnd(N, Cout, seed=4) if "rowbias" in extra else None
    res = rnd(N, Cout, Ho, Wo, seed=5) if "res" in extra else None
    if rowbias is not None:
        ref = ref + rowbias[:, :, None, None]
    if res is not None:
        ref = ref + res
    dy = rnd(N, Cout, Ho, Wo, seed=6)
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().to(_dev())
    xd = nhwc(x); wd = w.to(_dev()); bd = b.to(_dev())
    wp = torch.empty(w.numel(), device=_dev()); wpd = torch.empty(w.numel(), device=_dev())
    lib.check(L.cb_pack_conv_weight(ctx.h, lib.ptr(wd), lib.ptr(wp), Cout, Cin, ks, 0))
    lib.check(L.cb_pack_conv_weight(ctx.h, lib.ptr(wd), lib.ptr(wpd), Cout, Cin, ks, 1))
    y = torch.empty(N, Ho, Wo, Cout, device=_dev())
    rbd = rowbias.to(_dev()) if rowbias is not None else None
    rsd = nhwc(res) if res is not None else None
    lib.check(L.cb_conv2d_fwd(ctx.h, lib.ptr(xd), lib.ptr(wp), lib.ptr(bd), lib.ptr(rbd), lib.ptr(rsd), lib.ptr(y), N, H, W, Cin,
                              Cout, ks, MODES[mode]))
    ctx.sync()
    tag = f"conv[{mode},{N}x{H}x{W},{Cin}->{Cout},k{ks},{extra}]"
    close(tag + ".fwd", y.permute(0, 3, 1, 2), ref, 2e-5)
    if mode in ("same", "down", "up"):
        (dx_ref,) = torch.autograd.grad((ref * dy).sum(), [xr])
        dx0 = rnd(N, H, W, Cin, seed=7)
        for beta in (0.0, 1.0):
            dx = dx0.clone().to(_dev())
            lib.check(L.cb_conv2d_dgrad(ctx.h, HOLD(nhwc(dy)), lib.ptr(wpd), lib.ptr(dx), N, H, W, Cin, Cout, ks, MODES[mode], beta))
            ctx.sync()
            close(tag + f".dgrad(beta={beta})", dx.permute(0, 3, 1, 2), dx_ref + beta * dx0.permute(0, 3, 1, 2), 2e-5)
        if mode == "up":       # the same gradient as ONE 4x4 stride-2 conv over dy (dgrad = 2 pack, what the UNet's Upsample layers use)
            wpu = torch.empty(Cin * 16 * Cout, device=_dev())
            lib.check(L.cb_pack_conv_weight(ctx.h, lib.ptr(wd), lib.ptr(wpu), Cout, Cin, ks, 2))
            for beta in (0.0, 1.0):
                dx = dx0.clone().to(_dev())
                lib.check(L.cb_conv2d_dgrad(ctx.h, HOLD(nhwc(dy)), lib.ptr(wpu), lib.ptr(dx), N, H, W, Cin, Cout, ks, 5, beta))
                ctx.sync()
                close(tag + f".dgrad4x4(beta={beta})", dx.permute(0, 3, 1, 2), dx_ref + beta * dx0.permute(0, 3, 1, 2), 2e-5)


# ------------------------------------------------------------------------------------------------ the 80-column tiles
# cb_gemm_glds.hip::gemm_glds80_kernel (BM = 64 / 32 / 16 rows x 80 columns, 1 / 2 / 4 k-groups inside the block), pinned through
# cb_debug_force_plan; `split` > 1 adds the cross-block split-K on top. Same references and tolerance as the planner's own tiles.
@pytest.mark.parametrize("bm", [64, 32, 16])
@pytest.mark.parametrize("M,N,K,extra,split", [
    (4096, 320, 320, "bias+res", 1),       # the 64x64-level Linear: 64 x 4 = 256 blocks at BM = 64
    (256, 1280, 1280, "bias", 1),          # the 16x16 level: 16 x 16 blocks at BM = 16
    (77, 200, 96, "alpha", 1),             # ragged rows / columns (200 = 2.5 tiles), 3 k-tiles: fewer than k-groups at BM = 16
    (130, 88, 52, "beta", 1),              # K = 52: one full and one partial k-tile
    (64, 1280, 2560, "bias", 4),           # 8x8 level: k-groups inside the block AND a split-K over blocks
    (300, 84, 4, "none", 1),               # a single 16-byte chunk of K
])
def test_gemm_n80_tiles(cb, bm, M, N, K, extra, split):
    lib, ctx = cb
    lib.check(lib.load().cb_debug_force_plan(bm, 80, split))
    ctx.profile_begin()
    try:
        test_gemm(cb, M, N, K, 0, 0, 1, extra)
    finally:
        lib.check(lib.load().cb_debug_force_plan(0, 0, 0))
        prof = ctx.profile_end()
    ran = [r["kernel"] for r in prof]
    # a pinned plan the problem is not eligible for is dropped silently by the planner (cb_gemm_tiles.h): the test must see the kernel it names
    assert any(k.startswith(f"gemm_glds80_kernel<float,{bm},80,") for k in ran), f"the {bm}x80 kernel did not run: {ran}"
    # split-K folds its partials inside the launch (cb_debug_splitk_inlaunch, on by default): no reduce kernel follows
    assert not any(k.startswith("splitk_reduce") for k in ran), (split, ran)
    key = f"gemm[{M}x{N}x{K},a0b0,batch1,{extra}]"
    REPORT[f"n80.{bm}/{split}." + key] = REPORT.pop(key)


@pytest.mark.parametrize("bm", [64, 32, 16])
@pytest.mark.parametrize("N,H,W,Cin,Cout,ks,mode,extra,split", [
    (1, 64, 64, 320, 320, 3, "same", "none", 1),        # level-0 ResBlock conv: k-groups start in the middle of a tap at BM = 16 / 32
    (2, 16, 16, 32, 64, 3, "same", "rowbias+res", 1),   # one k-tile per tap, Cout = 64 < 80
    (1, 8, 8, 1280, 1280, 3, "same", "none", 3),        # 8x8 level with a split-K over blocks on top
    (2, 16, 16, 32, 160, 3, "down", "none", 1),         # stride 2
    (2, 16, 16, 64, 96, 1, "same", "none", 1),          # 1x1
])
def test_conv_n80_tiles(cb, bm, N, H, W, Cin, Cout, ks, mode, extra, split):
    lib, ctx = cb
    lib.check(lib.load().cb_debug_force_plan(bm, 80, split))
    ctx.profile_begin()
    try:
        test_conv_fwd_dgrad(cb, N, H, W, Cin, Cout, ks, mode, extra)
    finally:
        lib.check(lib.load().cb_debug_force_plan(0, 0, 0))
        prof = ctx.profile_end()
    ran = [r["kernel"] for r in prof]
    assert any(k.startswith(f"gemm_glds80_kernel<float,{bm},80,CONV") for k in ran), f"the {bm}x80 conv kernel did not run: {ran}"


def test_n80_fast_pieces_match_safe_mode(cb):
    """the 80-column kernel's FAST form (scalar-offset LDS-DMA pieces, ring slot = tile parity) against its per-lane form: cb_debug_glds_safe(1)
    forces FAST off and adds a full barrier per k-tile; results must be the same BIT FOR BIT. (The two-stage ring drains with vmcnt(0) in both
    modes — the counted wait belongs to the three-stage rings of gemm_glds_kernel, tested in test_glds_counted_wait_matches_safe_mode.)"""
    lib, ctx = cb
    L = lib.load()
    outs = {}
    try:
        for safe in (0, 1):
            lib.check(L.cb_debug_glds_safe(safe))
            for bm in (64, 32, 16):
                lib.check(L.cb_debug_force_plan(bm, 80, 1))
                for (M, N, K) in [(4096, 320, 2560), (1024, 640, 640), (256, 1280, 5120), (77, 200, 96)]:
                    A = rnd(1, M, K, seed=M + 1).to(_dev()); B = rnd(1, N, K, seed=N + 2).to(_dev())
                    Cd = torch.zeros(M, N, device=_dev())
                    d = lib.GemmDesc()
                    d.A, d.B, d.C = A.data_ptr(), B.data_ptr(), Cd.data_ptr()
                    d.M, d.N, d.K = M, N, K
                    d.lda = d.ldb = K; d.ldc = N; d.batch = 1; d.alpha = 1.0
                    lib.check(L.cb_gemm(ctx.h, C.byref(d)))
                    ctx.sync()
                    outs[(safe, bm, M, N, K)] = Cd.cpu()
    finally:
        lib.check(L.cb_debug_glds_safe(0)); lib.check(L.cb_debug_force_plan(0, 0, 0))
    for (safe, bm, M, N, K), v in outs.items():
        if safe == 0:
            assert v.abs().max() > 0 and torch.equal(v, outs[(1, bm, M, N, K)]), (bm, M, N, K)


@pytest.mark.parametrize("plan", [(64, 80, 4), (32, 80, 8), (16, 80, 4), (16, 80, 2), (64, 128, 3), (128, 128, 4), (128, 64, 2), (64, 64, 6)])
def test_splitk_inlaunch_matches_reduce_kernel(cb, plan):
    """Split-K launches fold their fp32 partial slabs INSIDE the launch: the tile's last-arriving block (agent-scope release / ticket /
    acquire, cb_gemm_tiles.h::splitk_arrive) sums all slabs in split order and runs the epilogue. cb_debug_splitk_inlaunch(0) restores
    partials + splitk_reduce_kernel, which sums in the same order: BIT FOR BIT the same C, whichever block arrives last — each case is
    launched 12 times in a row (other launches in flight on the stream, L1-warm reducers) and every repeat must match, every word."""
    lib, ctx = cb
    L = lib.load()
    bm, bn, split = plan
    cases = [(64, 1280, 2560, "bias+res"), (77, 768, 3072, "bias"), (256, 1280, 5120, "none"), (130, 88, 520, "beta"), (1024, 640, 1280, "res")]
    try:
        lib.check(L.cb_debug_force_plan(bm, bn, split))
        for ci, (M, N, K, extra) in enumerate(cases):
            A = rnd(1, M, K, seed=ci + 1).to(_dev()); B = rnd(1, N, K, seed=ci + 20).to(_dev())
            bias = rnd(N, seed=ci + 40).to(_dev()) if "bias" in extra else None
            R = rnd(M, N, seed=ci + 60).to(_dev()) if "res" in extra else None
            C0 = rnd(M, N, seed=ci + 80).to(_dev())
            outs = {}
            for mode, reps in ((0, 1), (1, 12)):
                lib.check(L.cb_debug_splitk_inlaunch(mode))
                for r in range(reps):
                    Cd = C0.clone()
                    d = lib.GemmDesc()
                    d.A, d.B, d.C = A.data_ptr(), B.data_ptr(), Cd.data_ptr()
                    d.M, d.N, d.K = M, N, K
                    d.lda = d.ldb = K; d.ldc = N; d.batch = 1; d.alpha = 1.0
                    d.beta = 1.0 if "beta" in extra else 0.0
                    if bias is not None: d.bias = bias.data_ptr()
                    if R is not None: d.R = R.data_ptr(); d.ldr = N
                    lib.check(L.cb_gemm(ctx.h, C.byref(d)))
                    outs[(mode, r)] = Cd
            ctx.sync()
            ref = outs[(0, 0)].cpu()
            assert ref.abs().max() > 0 and torch.isfinite(ref).all()
            for r in range(12):
                got = outs[(1, r)].cpu()
                assert torch.equal(got, ref), f"plan {plan}, case {cases[ci]}, repeat {r}: in-launch reduction differs from splitk_reduce_kernel " \
                                              f"(max {(got - ref).abs().max().item():.3e}, {(got != ref).sum().item()} words)"
    finally:
        lib.check(L.cb_debug_splitk_inlaunch(1)); lib.check(L.cb_debug_force_plan(0, 0, 0))


@pytest.mark.parametrize("plan", [(64, 80), (32, 80), (16, 80), (64, 128), (64, 64)])
def test_conv_fast_pieces_match_safe_mode(cb, plan):
    """the conv gather's FAST form (scalar-offset pieces; per-lane validity and (tap, channel) offset re-derived at a tap change only;
    A-side descriptor bias for negative row bases) against the per-lane form that cb_debug_glds_safe(1) forces: stride 1 / 2, pad 1,
    1x1, a split-K over taps, on the 80-column tiles and on gemm_glds_kernel's — BIT FOR BIT (round-4 advisor finding: only ROW
    GEMMs were compared before)."""
    lib, ctx = cb
    L = lib.load()
    bm, bn = plan
    cases = [(1, 64, 64, 320, 320, 3, "same", 1), (2, 33, 31, 64, 160, 3, "down", 1), (1, 8, 8, 1280, 1280, 3, "same", 3),
             (2, 16, 16, 64, 96, 1, "same", 1), (1, 16, 16, 640, 1280, 3, "same", 2)]
    outs = {}
    try:
        for safe in (0, 1):
            lib.check(L.cb_debug_glds_safe(safe))
            for ci, (N, H, W, Cin, Cout, ks, mode, split) in enumerate(cases):
                lib.check(L.cb_debug_force_plan(bm, bn, split))
                x = rnd(N, H, W, Cin, seed=ci + 1).to(_dev()); w = rnd(Cout, Cin, ks, ks, seed=ci + 20, scale=(Cin * ks * ks) ** -0.5).to(_dev())
                b = rnd(Cout, seed=ci + 40).to(_dev())
                wp = torch.empty(w.numel(), device=_dev())
                lib.check(L.cb_pack_conv_weight(ctx.h, lib.ptr(w), lib.ptr(wp), Cout, Cin, ks, 0))
                Ho, Wo = ((H + 1) // 2, (W + 1) // 2) if mode == "down" else (H, W)
                y = torch.zeros(N, Ho, Wo, Cout, device=_dev())
                lib.check(L.cb_conv2d_fwd(ctx.h, lib.ptr(x), lib.ptr(wp), lib.ptr(b), None, None, lib.ptr(y), N, H, W, Cin, Cout, ks, MODES[mode]))
                ctx.sync()
                outs[(safe, ci)] = y.cpu()
    finally:
        lib.check(L.cb_debug_glds_safe(0)); lib.check(L.cb_debug_force_plan(0, 0, 0))
    for ci in range(len(cases)):
        a, b_ = outs[(0, ci)], outs[(1, ci)]
        assert a.abs().max() > 0 and torch.isfinite(a).all()
        assert torch.equal(a, b_), f"plan {plan}, conv case {cases[ci]}: FAST and per-lane forms differ (max {(a - b_).abs().max().item():.3e})"


# ------------------------------------------------------------------------------------------------ norms
@pytest.mark.parametrize("N,HW,C,silu,eps", [(2, 64, 320, 1, 1e-5), (1, 4096, 320, 1, 1e-5), (2, 100, 64, 0, 1e-6),
                                             (1, 64, 2560, 1, 1e-5), (1, 5000, 128, 1, 1e-6), (2, 256, 960, 0, 1e-6),
                                             (1, 65536, 128, 1, 1e-6)])     # slabs > 48 KB per group take the chip-wide three-kernel path, smaller ones the one-launch kernel
def test_groupnorm(cb, N, HW, C, silu, eps):
    lib, ctx = cb
    L = lib.load()
    x = rnd(N, HW, C, seed=1) * 2.0 + 3.0   # non-zero mean stresses the variance computation
    gam = 1 + 0.1 * rnd(C, seed=2); bet = 0.1 * rnd(C, seed=3); dy = rnd(N, HW, C, seed=4)
    xr = x.clone().requires_grad_(True)
    yr = F.group_norm(xr.permute(0, 2, 1), 32, gam, bet, eps)
    if silu:
        yr = F.silu(yr)
    yr = yr.permute(0, 2, 1)
    (dxr,) = torch.autograd.grad((yr * dy).sum(), [xr])
    xd, gd, bd, dyd = x.to(_dev()), gam.to(_dev()), bet.to(_dev()), dy.to(_dev())
    y = torch.empty_like(xd); stats = torch.empty(N * 32 * 2, device=_dev()); dx = torch.empty_like(xd)
    lib.check(L.cb_groupnorm_silu_fwd(ctx.h, lib.ptr(xd), lib.ptr(gd), lib.ptr(bd), lib.ptr(y), lib.ptr(stats), N, HW, C, eps, silu))
    lib.check(L.cb_groupnorm_silu_bwd(ctx.h, lib.ptr(xd), lib.ptr(dyd), lib.ptr(gd), lib.ptr(bd), lib.ptr(stats), lib.ptr(dx), N, HW, C, silu, 0.0))
    ctx.sync()
    tag = f"groupnorm[{N}x{HW}x{C},silu{silu}]"
    close(tag + ".fwd", y, yr, 1e-5)
    close(tag + ".bwd", dx, dxr, 5e-5)


@pytest.mark.parametrize("rows,C", [(77, 768), (4096, 320), (256, 1280), (9, 64)])
def test_layernorm(cb, rows, C):
    lib, ctx = cb
    L = lib.load()
    x = rnd(rows, C, seed=1) + 0.5; gam = 1 + 0.1 * rnd(C, seed=2); bet = 0.1 * rnd(C, seed=3); dy = rnd(rows, C, seed=4)
    xr = x.clone().requires_grad_(True)
    yr = F.layer_norm(xr, (C,), gam, bet, 1e-5)
    (dxr,) = torch.autograd.grad((yr * dy).sum(), [xr])
    xd, gd, bd, dyd = x.to(_dev()), gam.to(_dev()), bet.to(_dev()), dy.to(_dev())
    y = torch.empty_like(xd); stats = torch.empty(rows * 2, device=_dev()); dx = torch.empty_like(xd)
    lib.check(L.cb_layernorm_fwd(ctx.h, lib.ptr(xd), lib.ptr(gd), lib.ptr(bd), lib.ptr(y), lib.ptr(stats), rows, C, 1e-5))
    lib.check(L.cb_layernorm_bwd(ctx.h, lib.ptr(xd), lib.ptr(dyd), lib.ptr(gd), lib.ptr(stats), lib.ptr(dx), rows, C, 0.0))
    ctx.sync()
    close(f"layernorm[{rows}x{C}].fwd", y, yr, 1e-5)
    close(f"layernorm[{rows}x{C}].bwd", dx, dxr, 2e-5)


def test_geglu(cb):
    lib, ctx = cb
    L = lib.load()
    M, D = 300, 160
    x = rnd(M, 2 * D, seed=1) * 1.5; dy = rnd(M, D, seed=2)
    xr = x.clone().requires_grad_(True)
    a, g = xr.chunk(2, dim=-1)
    yr = a * F.gelu(g)
    (dxr,) = torch.autograd.grad((yr * dy).sum(), [xr])
    xd, dyd = x.to(_dev()), dy.to(_dev())
    y = torch.empty(M, D, device=_dev()); dx = torch.empty_like(xd)
    lib.check(L.cb_geglu_fwd(ctx.h, lib.ptr(xd), lib.ptr(y), M, D))
    lib.check(L.cb_geglu_bwd(ctx.h, lib.ptr(xd), lib.ptr(dyd), lib.ptr(dx), M, D))
    ctx.sync()
    close("geglu.fwd", y, yr, 1e-5)
    close("geglu.bwd", dx, dxr, 1e-5)


# ------------------------------------------------------------------------------------------------ attention
def torch_attention(q, k, v, heads, scale, causal):
    B, Nq, Cc = q.shape
    d = Cc // heads
    sp = lambda t: t.reshape(B, t.shape[1], heads, d).permute(0, 2, 1, 3)
    s = torch.matmul(sp(q), sp(k).transpose(-1, -2)) * scale
    if causal:
        s = s + torch.full((Nq, k.shape[1]), torch.finfo(torch.float32).min).triu_(1)
    return torch.matmul(s.softmax(-1), sp(v)).permute(0, 2, 1, 3).reshape(B, Nq, Cc)


@pytest.mark.parametrize("B,Nq,Nk,Cc,heads,causal,fused", [
    (2, 256, 256, 320, 8, 0, 1),     # self-attention d=40, fused QKV buffer
    (1, 1024, 77, 640, 8, 0, 0),     # cross-attention, Nk=77, d=80
    (2, 77, 77, 768, 12, 1, 1),      # CLIP causal
    (1, 300, 300, 512, 1, 0, 1),     # VAE AttnBlock: one head, d=512
    (1, 64, 77, 1280, 8, 0, 0),      # d=160
    (1, 1024, 1024, 640, 8, 0, 1),   # 32x32 self-attention d=80: key-split forward / dQ (4 ranges) + query-split dK/dV
    (1, 256, 256, 1280, 8, 0, 1),    # 16x16 self-attention d=160 (32-key tiles): key-split
    (1, 320, 600, 320, 8, 0, 0),     # key-split with a ragged last range (600 keys = 192 + 192 + 192 + 24), d=40
])
def test_attention(cb, B, Nq, Nk, Cc, heads, causal, fused):
    lib, ctx = cb
    L = lib.load()
    d = Cc // heads
    scale = d ** -0.5
    q = rnd(B, Nq, Cc, seed=1); k = rnd(B, Nk, Cc, seed=2); v = rnd(B, Nk, Cc, seed=3); do = rnd(B, Nq, Cc, seed=4)
    qr, kr, vr = (t.clone().requires_grad_(True) for t in (q, k, v))
    oref = torch_attention(qr, kr, vr, heads, scale, causal)
    dq, dk, dv = torch.autograd.grad((oref * do).sum(), [qr, kr, vr])
    desc = lib.AttentionDesc()
    if fused:
        qkv = torch.cat([q, k, v], dim=-1).contiguous().to(_dev())
        dqkv = torch.full_like(qkv, float("nan"))
        desc.qbuf = desc.kvbuf = qkv.data_ptr(); desc.ldq = desc.ldkv = 3 * Cc
        desc.q_off, desc.k_off, desc.v_off = 0, Cc, 2 * Cc
        desc.dqbuf = desc.dkvbuf = dqkv.data_ptr()
    else:
        qd = q.to(_dev()); kv = torch.cat([k, v], dim=-1).contiguous().to(_dev())
        dqd = torch.full_like(qd, float("nan")); dkv = torch.full_like(kv, float("nan"))
        desc.qbuf = qd.data_ptr(); desc.ldq = Cc; desc.q_off = 0
        desc.kvbuf = kv.data_ptr(); desc.ldkv = 2 * Cc; desc.k_off, desc.v_off = 0, Cc
        desc.dqbuf = dqd.data_ptr(); desc.dkvbuf = dkv.data_ptr()
    desc.B, desc.Nq, desc.Nk, desc.C, desc.heads = B, Nq, Nk, Cc, heads
    desc.scale, desc.causal = scale, causal
    out = torch.empty(B, Nq, Cc, device=_dev()); dod = do.to(_dev())
    desc.out = out.data_ptr(); desc.dout = dod.data_ptr()
    lib.check(L.cb_attention(ctx.h, C.byref(desc)))
    ctx.sync()
    tag = f"attention[B{B},{Nq}x{Nk},C{Cc},h{heads},causal{causal}]"
    close(tag + ".out", out, oref, 2e-5)
    if fused:
        gq, gk, gv = dqkv[..., :Cc], dqkv[..., Cc:2 * Cc], dqkv[..., 2 * Cc:]
    else:
        gq, gk, gv = dqd, dkv[..., :Cc], dkv[..., Cc:]
    close(tag + ".dq", gq, dq, 5e-5)
    close(tag + ".dk", gk, dk, 5e-5)
    close(tag + ".dv", gv, dv, 5e-5)


# ------------------------------------------------------------------------------------------------ step glue vs the oracle
def test_glue_ops_vs_oracle(cb, golden_dir):
    from oracle import cb_oracle as O
    lib, ctx = cb
    L = lib.load()
    fx = torch.load(os.path.join(golden_dir, "schedule.pt"), weights_only=False)
    # timestep embedding
    t = fx["temb_t"].to(_dev()); out = torch.empty(t.shape[0], 320, device=_dev())
    lib.check(L.cb_timestep_embedding(ctx.h, lib.ptr(t, torch.int64), lib.ptr(out), t.shape[0], 320))
    close("timestep_embedding", out, fx["temb"], 2e-4)   # cos/sin of arguments up to ~1e3: device libm vs host libm
    # q_sample + mse
    x0, eps, tt = fx["q_x0"], fx["q_eps"], fx["q_t"]
    B, Cc, H, W = x0.shape
    x0d = x0.permute(0, 2, 3, 1).contiguous().to(_dev()); xt = torch.empty_like(x0d)
    sa, sb = fx["sqrt_alphas_cumprod"].to(_dev()), fx["sqrt_one_minus_alphas_cumprod"].to(_dev())
    lib.check(L.cb_q_sample(ctx.h, lib.ptr(x0d), HOLD(eps), HOLD(tt, torch.int64), lib.ptr(sa), lib.ptr(sb),
                            lib.ptr(xt), B, Cc, H * W))
    close("q_sample", xt.permute(0, 3, 1, 2), fx["q_xt"], 1e-6)
    pred = rnd(B, Cc, H, W, seed=9)
    pr = pred.clone().requires_grad_(True)
    lref = F.mse_loss(eps, pr, reduction="none").mean([1, 2, 3]).mean()
    (gref,) = torch.autograd.grad(lref, [pr])
    loss = torch.zeros(1, device=_dev()); dp = torch.empty_like(x0d)
    lib.check(L.cb_mse_loss(ctx.h, HOLD(pred.permute(0, 2, 3, 1)), HOLD(eps), lib.ptr(loss),
                            lib.ptr(dp), B, Cc, H * W))
    close("mse.loss", loss, lref.reshape(1), 1e-5)
    close("mse.grad", dp.permute(0, 3, 1, 2), gref, 1e-5)
    # face warp + resize
    Bf, S = 2, 96
    faces = torch.rand(Bf, S, S, 6, generator=torch.Generator().manual_seed(5)) * 2 - 1
    cat = torch.cat(faces.chunk(2, -1), 0)
    wref, sref = O.face_warp_resize(cat, 112)
    warped = torch.empty(2 * Bf, S, S, 3, device=_dev()); small = torch.empty(2 * Bf, 112, 112, 3, device=_dev())
    lib.check(L.cb_face_warp_resize(ctx.h, HOLD(faces), lib.ptr(warped), lib.ptr(small), Bf, S, 2, 112))
    close("face.warp", warped.permute(0, 3, 1, 2), wref, 1e-4)   # analytic grid: coordinates within two fp32 ulps of torch's
    close("face.resize", small.permute(0, 3, 1, 2), sref, 1e-4)
    # with the reference's own sampling grid as a constant table (cb_ctx_set_warp_grid; meta_net.py:253-257) the warp samples at
    # torch's coordinates and follows grid_sample's / interpolate's operation order: BIT-EXACT, at the real 512 x 512 size too
    import hashlib
    for S in (96, 512):
        faces = torch.rand(Bf, S, S, 6, generator=torch.Generator().manual_seed(5)) * 2 - 1
        wref, sref = O.face_warp_resize(torch.cat(faces.chunk(2, -1), 0), 112)
        ctx.ensure_warp_grid(S)
        warped = torch.empty(2 * Bf, S, S, 3, device=_dev()); small = torch.empty(2 * Bf, 112, 112, 3, device=_dev())
        lib.check(L.cb_face_warp_resize(ctx.h, HOLD(faces), lib.ptr(warped), lib.ptr(small), Bf, S, 2, 112))
        ctx.sync()
        dw = (warped.permute(0, 3, 1, 2).cpu() - wref).abs().max().item(); ds = (small.permute(0, 3, 1, 2).cpu() - sref).abs().max().item()
        print(f"[warp grid table] S={S}: warp max diff {dw:.3g}, resize max diff {ds:.3g}; host grid sha1 "
              f"{hashlib.sha1(ctx.warp_grid_host.numpy().tobytes()).hexdigest()[:12]} (cpu capability {torch.backends.cpu.get_cpu_capability()})")
        assert torch.equal(warped.permute(0, 3, 1, 2).cpu(), wref), f"warp with the reference's grid table is not bit-exact at S={S} ({dw:.3g})"
        assert torch.equal(small.permute(0, 3, 1, 2).cpu(), sref), f"resize after the table warp is not bit-exact at S={S} ({ds:.3g})"
    L.cb_ctx_set_warp_grid(ctx.h, None, 0); ctx._warp_S = None
    # row surgery: bit-exact against the reference golden cases (copies only)
    for i, case in enumerate(torch.load(os.path.join(golden_dir, "shift_rows.pt"), weights_only=False)):
        toks = case["tokens"][None].contiguous().to(_dev())
        T, Cw = case["embedded"].shape
        o = torch.empty(1, T, Cw, device=_dev()); mp = torch.empty(T, dtype=torch.int32, device=_dev())
        lib.check(L.cb_row_surgery(ctx.h, HOLD(case["embedded"][None]), HOLD(case["z"][None]),
                                   lib.ptr(toks, torch.int64), lib.ptr(o), C.c_void_p(mp.data_ptr()), 1, T, Cw, case["reps"], case["placeholder"]))
        ctx.sync()
        assert torch.equal(o[0].cpu(), case["out"]), f"row surgery case {i} is not bit-exact"
    # several placeholder tokens per prompt (two / three persons), incl. the reference's own helpers.py:44-54 example
    for i, case in enumerate(torch.load(os.path.join(golden_dir, "shift_rows_multi.pt"), weights_only=False)):
        toks = case["tokens"][None].contiguous().to(_dev())
        T, Cw = case["embedded"].shape
        nph = len(case["placeholders"])
        o = torch.empty(1, T, Cw, device=_dev()); mp = torch.empty(T, dtype=torch.int32, device=_dev())
        ph = torch.tensor([case["placeholders"]], dtype=torch.int64, device=_dev())
        zcat = dv(torch.cat(case["z"], dim=0)[None])                       # [1][nph*reps][C]
        lib.check(L.cb_row_surgery_multi(ctx.h, HOLD(case["embedded"][None]), lib.ptr(zcat), lib.ptr(toks, torch.int64), lib.ptr(o),
                                         C.c_void_p(mp.data_ptr()), 1, T, Cw, case["reps"], lib.ptr(ph, torch.int64), nph))
        ctx.sync()
        assert torch.equal(o[0].cpu(), case["out"]), f"multi-placeholder row surgery case {i} is not bit-exact"
    # AdamW, 5 steps
    g = torch.Generator().manual_seed(0)
    p = torch.randn(1000, generator=g); m = torch.zeros(1000); v = torch.zeros(1000)
    pd, md, vd = p.to(_dev()), m.to(_dev()), v.to(_dev()); step = torch.zeros(1, dtype=torch.int32, device=_dev())
    for s in range(1, 6):
        gr = torch.randn(1000, generator=g)
        p, m, v = O.adamw_step(p, gr, m, v, s, 1e-2)
        lib.check(L.cb_adamw_step(ctx.h, lib.ptr(pd), lib.ptr(md), lib.ptr(vd), HOLD(gr), 1000, C.c_void_p(step.data_ptr()),
                                  1e-2, 0.9, 0.999, 1e-8, 1e-2))
    close("adamw.p", pd, p, 1e-5)
    assert int(step.item()) == 5


def test_feeder_augmentations_vs_oracle(cb):
    """cb_face_augment / cb_paste_on_canvas against the oracle's restatement of face_id.py:526-532 (torchvision tensor
    formulas) and :712-730 (`_add_bg`), plus the batch-level DeviceAugmenter contract (layouts, dtypes, ranges)"""
    from oracle import cb_oracle as O
    lib, ctx = cb
    g = torch.Generator().manual_seed(5)
    B, S = 3, 96
    low = torch.rand(B, 3, 12, 12, generator=g) * 2 - 1
    faces = torch.nn.functional.interpolate(low, size=S, mode="bilinear", align_corners=True).permute(0, 2, 3, 1).contiguous()
    params = torch.tensor([[1.0, 1.13, 0.87, 1.19, 0.009], [0.0, 0.81, 1.2, 0.8, -0.01], [1.0, 1.0, 1.0, 1.0, 0.0]])
    out = lib.face_augment(ctx, dv(faces), dv(params))
    ctx.sync()
    ref = torch.stack([O.color_jitter_flip(faces[b], bool(params[b, 0]), *[float(v) for v in params[b, 1:]]) for b in range(B)])
    close("feeder.jitter", out, ref, 2e-5)
    assert torch.equal(out[2].cpu(), faces[2].flip(1)) or (out[2].cpu() - faces[2].flip(1)).abs().max() < 1e-6   # identity jitter = flip only
    # ColorJitter's per-call op order (fn_idx): the feeder runs four single-op passes, pass k = the k-th op of each sample's own permutation
    from celebbasis_amd.ldm.data.face_id import DeviceAugmenter
    orders = [[3, 1, 0, 2], [1, 2, 3, 0], [0, 1, 2, 3]]
    rows = torch.tensor([DeviceAugmenter.passes([float(v) for v in params[b]], orders[b]) for b in range(B)])
    chain = dv(faces)
    for k in range(4):
        chain = lib.face_augment(ctx, chain, dv(rows[:, 5 * k:5 * k + 5].contiguous()))
    ctx.sync()
    refo = torch.stack([O.color_jitter_flip(faces[b], bool(params[b, 0]), *[float(v) for v in params[b, 1:]], order=orders[b]) for b in range(B)])
    close("feeder.jitter.order", chain, refo, 2e-5)
    assert float((refo[0] - ref[0]).abs().max()) > 1e-2         # the permutation is not a no-op on this image
    geom = torch.tensor([[40, 43, 10, 7], [96, 96, 0, 0], [1, 1, 95, 95]], dtype=torch.int32)
    pasted = lib.paste_on_canvas(ctx, out, geom.to(_dev()))
    ctx.sync()
    refp = torch.stack([O.add_bg(ref[b], *[int(v) for v in geom[b]]) for b in range(B)])
    close("feeder.add_bg", pasted, refp, 2e-5)
    # batch-level feeder
    from celebbasis_amd.ldm.data.face_id import DeviceAugmenter, FaceIdDatasetOneShot
    ds = FaceIdDatasetOneShot(pickle_path=None, num_ids=2, size=64, seed=3, diff_cnt=0, synthetic=True)
    b = DeviceAugmenter(ds, _dev()).batch([0, 1, 2])
    ctx.sync()
    assert tuple(b["image"].shape) == (3, 64, 64, 3) and tuple(b["image_ori"]["faces"].shape) == (3, 64, 64, 6)
    assert b["image_ori"]["ids"].dtype == torch.int64 and b["image_ori"]["ids"].tolist() == [[0, 0], [1, 1], [0, 0]]
    # (bilinear weights sum to 1 only to fp32 round-off: 1 + 2.4e-7 has been seen)
    assert float(b["image"].min()) >= -1.0 - 1e-6 and float(b["image"].max()) <= 1.0 + 1e-6 and torch.isfinite(b["image_ori"]["faces"]).all()
    assert all("sks" in c for c in b["caption"])


def test_device_feeder_matches_cpu_feeder(cb):
    """the product's default feeder (celebbasis_amd.main -> DeviceFeeder -> DeviceAugmenter, pixels on the GPU) against the
    reference-shaped CPU path (`FaceIdDatasetOneShot.__getitem__` + DataModuleFromConfig.collate, face_id.py:598-644) on the
    SAME generator state: same draws in the same order => the same batch (images to fp32 round-off, ids / captions exactly)"""
    from celebbasis_amd.ldm.data.face_id import DeviceAugmenter, DeviceFeeder, FaceIdDatasetOneShot
    from celebbasis_amd.main import DataModuleFromConfig
    lib, ctx = cb
    mk = lambda: FaceIdDatasetOneShot(pickle_path=None, num_ids=3, size=64, seed=9, diff_cnt=0, synthetic=True, repeats=4)
    ds_cpu, ds_dev = mk(), mk()
    idx = [0, 1, 2, 4, 8]
    ref = DataModuleFromConfig.collate([ds_cpu[i] for i in idx])
    got = DeviceAugmenter(ds_dev, _dev()).batch(idx)
    torch.cuda.synchronize()
    close("feeder.batch.image", got["image"], ref["image"], 2e-5)
    close("feeder.batch.faces", got["image_ori"]["faces"], ref["image_ori"]["faces"], 2e-5)
    assert got["image_ori"]["ids"].cpu().tolist() == ref["image_ori"]["ids"].tolist() and got["caption"] == ref["caption"]
    assert got["id_idx"].tolist() == ref["id_idx"].tolist() and got["img_idx"].tolist() == ref["img_idx"].tolist()
    assert ds_cpu.rng.random() == ds_dev.rng.random()        # both paths consumed exactly the same number of draws
    # one epoch of the feeder: every sample once, last (short) batch kept, tokens ready on the device and cached per caption
    calls = []
    def tok(caps):
        calls.append(tuple(caps))
        return torch.full((len(caps), 77), 7, dtype=torch.int64)
    torch.manual_seed(0)
    feeder = DeviceFeeder(mk(), batch_size=5, device=_dev(), tokenize=tok)
    batches = list(feeder)
    assert len(batches) == len(feeder) == 3 and [b["image"].shape[0] for b in batches] == [5, 5, 2]
    assert sorted(int(i) for b in batches for i in b["img_idx"]) == list(range(12))
    assert all(b["tokens"].is_cuda and tuple(b["tokens"].shape) == (b["image"].shape[0], 77) for b in batches)
    assert len(calls) == len(set(calls)) <= 12       # one tokenizer call per DISTINCT caption


def test_celeb_head_forward_vs_oracle(cb):
    """cb_celeb_head_forward (the `test_mode: image` head, meta_net.py:262-289) against the oracle's restatement"""
    from oracle import cb_oracle as O
    lib, ctx = cb
    L = lib.load()
    g = torch.Generator().manual_seed(11)
    N, I, es, K, Wd = 3, 512, 2, 512, 768
    feat = torch.randn(N, I, generator=g) * 3.0
    W = {"embedding_manager.meta_id_net.stylegan_mlp.net.0.weight": torch.randn(es * K, I, generator=g),
         "embedding_manager.meta_id_net.stylegan_mlp.net.0.bias": torch.randn(es * K, generator=g) * 0.1,
         "celeb_embeddings": torch.randn(es, 1 + K, Wd, generator=g) * 0.05}
    coef = torch.empty(N, es * K, device=_dev()); z = torch.empty(N, es * Wd, device=_dev())
    lib.check(L.cb_celeb_head_forward(ctx.h, HOLD(feat), HOLD(W["embedding_manager.meta_id_net.stylegan_mlp.net.0.weight"]),
                                      HOLD(W["embedding_manager.meta_id_net.stylegan_mlp.net.0.bias"]), HOLD(W["celeb_embeddings"]),
                                      N, I, es, K, Wd, lib.ptr(coef), lib.ptr(z)))
    ctx.sync()
    cref, zref = O.celeb_head(W, torch.nn.functional.normalize(feat, dim=-1, p=2), es, K)
    close("celeb_head.coef", coef.reshape(N, es, K), cref, 2e-5)
    close("celeb_head.z", z.reshape(N, es, Wd), zref, 2e-5)
