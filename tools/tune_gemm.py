#!/usr/bin/env python
"""tools/tune_gemm.py — planner calibration: for every GEMM / implicit-GEMM conv shape of the training step, time every
(tile, split-K) plan (cb_debug_force_plan) through the C ABI and compare the best with the planner's own choice.

    python tools/tune_gemm.py [--dtype f32|bf16] [--batch 1|4] [--reps 10] [--quick]     -> gpurun_out/tune_gemm_<dtype>_b<B>.json
"""
import argparse
import ctypes as C
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from celebbasis_amd import lib  # noqa: E402

DEV = torch.device("cuda:0")
# (H, W, Cin, Cout, launches per step forward + input gradient) at batch 1 — UNet levels, then the VAE encoder
CONVS = [(64, 64, 320, 320, 14), (64, 64, 640, 320, 4), (64, 64, 960, 320, 2), (32, 32, 320, 640, 2), (32, 32, 640, 640, 12),
         (32, 32, 1280, 640, 2), (32, 32, 1920, 640, 2), (16, 16, 640, 1280, 2), (16, 16, 1280, 1280, 14), (16, 16, 2560, 1280, 4),
         (8, 8, 1280, 1280, 22), (8, 8, 2560, 1280, 6),
         (512, 512, 128, 128, 4), (256, 256, 256, 256, 3), (128, 128, 512, 512, 3), (64, 64, 512, 512, 8)]
# (rows per image, N, K, launches)
LINS = [(4096, 320, 320, 30), (4096, 2560, 320, 10), (4096, 320, 1280, 10), (1024, 640, 640, 30), (1024, 5120, 640, 10),
        (1024, 640, 2560, 10), (256, 1280, 1280, 30), (256, 10240, 1280, 10), (256, 1280, 5120, 10), (77, 320, 768, 10), (77, 1280, 768, 12)]
TILES = [(128, 128), (128, 64), (64, 128), (64, 64), (256, 128), (64, 80), (32, 80), (16, 80)]   # 256x128: bf16 only; x80: fp32 only (round 4)
SPLITS = [1, 2, 3, 4, 6, 8, 12, 16]


def timeit(fn, reps):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--quick", action="store_true", help="planner's own choice only")
    a = ap.parse_args()
    tdt = lib.torch_dtype(a.dtype)
    ctx = lib.Context(dtype=a.dtype)
    L = lib.load()
    g = torch.Generator().manual_seed(0)
    B = a.batch
    ksz = 64 if a.dtype == "bf16" else 32
    out = []
    problems = []
    for (H, W, Ci, Co, cnt) in CONVS:
        x = torch.randn(B, H, W, Ci, generator=g).to(tdt).to(DEV)
        w = (torch.randn(Co, Ci, 3, 3, generator=g) * (Ci * 9) ** -0.5).to(DEV)
        wp = torch.empty(Co * 9 * Ci, dtype=tdt, device=DEV)
        lib.check(L.cb_pack_conv_weight(ctx.h, lib.ptr(w), lib.ptr(wp, tdt), Co, Ci, 3, 0))
        b = torch.zeros(Co, device=DEV)
        y = torch.empty(B, H, W, Co, dtype=tdt, device=DEV)
        keep = (x, wp, b, y)
        fn = (lambda x=x, wp=wp, b=b, y=y, H=H, W=W, Ci=Ci, Co=Co:
              lib.check(L.cb_conv2d_fwd(ctx.h, lib.ptr(x, tdt), lib.ptr(wp, tdt), lib.ptr(b), None, None, lib.ptr(y, tdt), B, H, W, Ci, Co, 3, 0)))
        problems.append((f"conv {B}x{H}x{W} {Ci}->{Co}", B * H * W, Co, 9 * Ci, cnt, fn, keep))
    for (M1, Nn, K, cnt) in LINS:
        M = M1 * B
        A = torch.randn(M, K, generator=g).to(tdt).to(DEV); Bm = (torch.randn(Nn, K, generator=g) * K ** -0.5).to(tdt).to(DEV)
        Cc = torch.empty(M, Nn, dtype=tdt, device=DEV)
        d = lib.GemmDesc()
        d.A, d.B, d.C = A.data_ptr(), Bm.data_ptr(), Cc.data_ptr()
        d.M, d.N, d.K = M, Nn, K
        d.lda = d.ldb = K; d.ldc = Nn; d.batch = 1; d.alpha = 1.0; d.dtype = lib.dtype_code(a.dtype)
        fn = (lambda d=d: lib.check(L.cb_gemm(ctx.h, C.byref(d))))
        problems.append((f"gemm {M}x{Nn}x{K}", M, Nn, K, cnt, fn, (A, Bm, Cc, d)))
    tot_plan = tot_best = 0.0
    for name, M, N, K, cnt, fn, keep in problems:
        fl = 2.0 * M * N * K
        lib.check(L.cb_debug_force_plan(0, 0, 0))
        t_plan = timeit(fn, a.reps)
        row = dict(op=name, M=M, N=N, K=K, launches=cnt, planner_us=round(t_plan * 1e6, 1), planner_tflops=round(fl / t_plan / 1e12, 1))
        best = (t_plan, "planner")
        if not a.quick:
            sweep = {}
            for (bm, bn) in TILES:
                if (bm >= 128 and M <= 64) or (bn == 128 and N <= 64) or (bm == 256 and (a.dtype != "bf16" or M <= 128)) or (bn == 80 and a.dtype != "f32"):
                    continue
                for sp in SPLITS:
                    if sp > 1 and (K // ksz) // sp < 2:
                        break
                    blocks = -(-M // bm) * -(-N // bn) * sp
                    if blocks > 4096 and sp > 1:
                        break
                    lib.check(L.cb_debug_force_plan(bm, bn, sp))
                    t = timeit(fn, a.reps)
                    sweep[f"{bm}x{bn}/{sp}"] = round(t * 1e6, 1)
                    if t < best[0]:
                        best = (t, f"{bm}x{bn}/{sp}")
            row.update(best=best[1], best_us=round(best[0] * 1e6, 1), best_tflops=round(fl / best[0] / 1e12, 1), sweep=sweep)
        out.append(row)
        tot_plan += t_plan * cnt; tot_best += best[0] * cnt
        print(f"{name:30s} planner {row['planner_us']:8.1f} us {row['planner_tflops']:6.1f} TF" +
              ("" if a.quick else f"   best {row['best']:>11s} {row['best_us']:8.1f} us {row['best_tflops']:6.1f} TF"), flush=True)
    lib.check(L.cb_debug_force_plan(0, 0, 0))
    print(f"launch-weighted: planner {tot_plan * 1e3:.3f} ms, best plans {tot_best * 1e3:.3f} ms")
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", f"tune_gemm_{a.dtype}_b{B}.json"), "w") as f:
        json.dump(dict(dtype=a.dtype, batch=B, planner_ms=round(tot_plan * 1e3, 3), best_ms=round(tot_best * 1e3, 3), rows=out), f, indent=1)
    del ctx


if __name__ == "__main__":
    main()
