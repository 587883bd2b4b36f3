#!/usr/bin/env python
"""Per-shape kernel micro-benchmarks through the C ABI (HIP events on the launch stream). Shapes are the SD v1-4 worklist
(SURVEY.md Appendix A). Prints one line per case and writes gpurun_out/kernels_bench.json.
    python tools/bench_kernels.py [--iters 20]"""
import argparse
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from celebbasis_amd import lib  # noqa: E402

DEV = "cuda:0"


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    L = lib.load()
    ctx = lib.Context()
    res = []

    def rec(kind, name, ms, flops=None, bytes_=None):
        r = dict(kind=kind, name=name, ms=round(ms, 4))
        if flops:
            r["tflops"] = round(flops / ms / 1e9, 2)
        if bytes_:
            r["gbps"] = round(bytes_ / ms / 1e6, 1)
        res.append(r)
        print(json.dumps(r), flush=True)

    # ---- conv 3x3 (N,H,W,Cin,Cout,mode) : UNet A.1 + VAE A.6
    convs = [(1, 512, 512, 128, 128, 0), (1, 256, 256, 256, 256, 0), (1, 128, 128, 512, 512, 0), (1, 64, 64, 512, 512, 0),
             (1, 64, 64, 320, 320, 0), (1, 64, 64, 640, 320, 0), (1, 32, 32, 640, 640, 0), (1, 32, 32, 1280, 640, 0),
             (1, 16, 16, 1280, 1280, 0), (1, 16, 16, 2560, 1280, 0), (1, 8, 8, 1280, 1280, 0), (1, 8, 8, 2560, 1280, 0),
             (1, 512, 512, 3, 128, 0), (1, 512, 512, 128, 128, 2), (1, 32, 32, 640, 640, 3), (1, 64, 64, 320, 320, 1)]
    for (N, H, W, Ci, Co, mode) in convs:
        x = torch.randn(N, H, W, Ci, device=DEV); w = torch.randn(Co, 9 * Ci, device=DEV) * 0.01; b = torch.zeros(Co, device=DEV)
        Ho, Wo = {0: (H, W), 1: (H // 2, W // 2), 2: (H // 2, W // 2), 3: (2 * H, 2 * W)}[mode]
        y = torch.empty(N, Ho, Wo, Co, device=DEV)
        fn = lambda: lib.check(L.cb_conv2d_fwd(ctx.h, lib.ptr(x), lib.ptr(w), lib.ptr(b), None, None, lib.ptr(y), N, H, W, Ci, Co, 3, mode))
        ms = timeit(fn, args.iters)
        rec("conv3x3", f"{N}x{H}x{W} {Ci}->{Co} mode{mode}", ms, flops=2.0 * N * Ho * Wo * Co * 9 * Ci)
    # ---- linear / 1x1 (rows, in, out)
    lins = [(4096, 320, 320), (4096, 320, 2560), (4096, 1280, 320), (1024, 640, 5120), (1024, 2560, 640), (256, 1280, 10240),
            (256, 5120, 1280), (64, 1280, 1280), (77, 768, 2560), (77, 768, 3072), (4096, 320, 960), (262144, 128, 128)]
    for (M, K, N) in lins:
        A = torch.randn(M, K, device=DEV); B = torch.randn(N, K, device=DEV) * 0.01; Cc = torch.empty(M, N, device=DEV)
        d = lib.GemmDesc(); d.A, d.B, d.C = A.data_ptr(), B.data_ptr(), Cc.data_ptr(); d.M, d.N, d.K = M, N, K
        d.lda, d.ldb, d.ldc = K, K, N; d.batch = 1; d.alpha = 1.0
        ms = timeit(lambda: lib.check(L.cb_gemm(ctx.h, C.byref(d))), args.iters)
        rec("linear", f"{M}x{K}->{N}", ms, flops=2.0 * M * N * K)
        d2 = lib.GemmDesc(); d2.A, d2.B = Cc.data_ptr(), B.data_ptr(); dx = torch.empty(M, K, device=DEV); d2.C = dx.data_ptr()
        d2.M, d2.N, d2.K = M, K, N; d2.lda, d2.ldb, d2.ldc = N, K, K; d2.b_layout = 1; d2.batch = 1; d2.alpha = 1.0
        ms = timeit(lambda: lib.check(L.cb_gemm(ctx.h, C.byref(d2))), args.iters)
        rec("linear_dgrad", f"{M}x{N}->{K}", ms, flops=2.0 * M * N * K)
    # ---- attention fwd+bwd (B, Nq, Nk, C, heads)
    for (B, Nq, Nk, Cc, h) in [(1, 4096, 4096, 320, 8), (1, 4096, 77, 320, 8), (1, 1024, 1024, 640, 8), (1, 256, 256, 1280, 8), (1, 4096, 4096, 512, 1)]:
        fused = Nq == Nk
        dsc = lib.AttentionDesc()
        if fused:
            qkv = torch.randn(B, Nq, 3 * Cc, device=DEV); dqkv = torch.empty_like(qkv)
            dsc.qbuf = dsc.kvbuf = qkv.data_ptr(); dsc.ldq = dsc.ldkv = 3 * Cc; dsc.q_off, dsc.k_off, dsc.v_off = 0, Cc, 2 * Cc
            dsc.dqbuf = dsc.dkvbuf = dqkv.data_ptr()
        else:
            q = torch.randn(B, Nq, Cc, device=DEV); kv = torch.randn(B, Nk, 2 * Cc, device=DEV); dq = torch.empty_like(q); dkv = torch.empty_like(kv)
            dsc.qbuf = q.data_ptr(); dsc.ldq = Cc; dsc.kvbuf = kv.data_ptr(); dsc.ldkv = 2 * Cc; dsc.k_off, dsc.v_off = 0, Cc
            dsc.dqbuf = dq.data_ptr(); dsc.dkvbuf = dkv.data_ptr()
        dsc.B, dsc.Nq, dsc.Nk, dsc.C, dsc.heads = B, Nq, Nk, Cc, h
        dsc.scale = (Cc // h) ** -0.5
        out = torch.empty(B, Nq, Cc, device=DEV); do = torch.randn(B, Nq, Cc, device=DEV); dsc.out = out.data_ptr()
        ms_f = timeit(lambda: lib.check(L.cb_attention(ctx.h, C.byref(dsc))), max(3, args.iters // 4))
        rec("attention_fwd", f"B{B} {Nq}x{Nk} C{Cc} h{h}", ms_f, flops=4.0 * B * Nq * Nk * Cc)
        dsc.dout = do.data_ptr()
        ms_fb = timeit(lambda: lib.check(L.cb_attention(ctx.h, C.byref(dsc))), max(3, args.iters // 4))
        rec("attention_fwd+bwd", f"B{B} {Nq}x{Nk} C{Cc} h{h}", ms_fb, flops=(12.0 if Nq == Nk else 12.0) * B * Nq * Nk * Cc)
    # ---- GroupNorm+SiLU / LayerNorm (HBM bound)
    for (N, HW, Cc) in [(1, 262144, 128), (1, 65536, 256), (1, 16384, 512), (1, 4096, 512), (1, 4096, 320), (1, 4096, 960), (1, 1024, 640), (1, 256, 1280), (1, 64, 2560)]:
        x = torch.randn(N, HW, Cc, device=DEV); g = torch.ones(Cc, device=DEV); b = torch.zeros(Cc, device=DEV); y = torch.empty_like(x)
        st = torch.empty(N * 64, device=DEV); dy = torch.randn_like(x); dx = torch.empty_like(x)
        ms = timeit(lambda: lib.check(L.cb_groupnorm_silu_fwd(ctx.h, lib.ptr(x), lib.ptr(g), lib.ptr(b), lib.ptr(y), lib.ptr(st), N, HW, Cc, 1e-5, 1)), args.iters)
        rec("groupnorm_silu_fwd", f"{N}x{HW}x{Cc}", ms, bytes_=3.0 * x.numel() * 4)
        ms = timeit(lambda: lib.check(L.cb_groupnorm_silu_bwd(ctx.h, lib.ptr(x), lib.ptr(dy), lib.ptr(g), lib.ptr(b), lib.ptr(st), lib.ptr(dx), N, HW, Cc, 1, 0.0)), args.iters)
        rec("groupnorm_silu_bwd", f"{N}x{HW}x{Cc}", ms, bytes_=5.0 * x.numel() * 4)
    for (rows, Cc) in [(4096, 320), (1024, 640), (256, 1280), (77, 768)]:
        x = torch.randn(rows, Cc, device=DEV); g = torch.ones(Cc, device=DEV); b = torch.zeros(Cc, device=DEV); y = torch.empty_like(x)
        st = torch.empty(rows * 2, device=DEV)
        ms = timeit(lambda: lib.check(L.cb_layernorm_fwd(ctx.h, lib.ptr(x), lib.ptr(g), lib.ptr(b), lib.ptr(y), lib.ptr(st), rows, Cc, 1e-5)), args.iters)
        rec("layernorm_fwd", f"{rows}x{Cc}", ms, bytes_=2.0 * x.numel() * 4)
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/kernels_bench.json", "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
