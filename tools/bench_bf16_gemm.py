#!/usr/bin/env python
"""bench_bf16_gemm.py — per-shape throughput of the bf16 GEMM / implicit-GEMM conv kernels at the BASELINE config-3 shapes
(bs = 4) through the C ABI. The kernel variant is chosen by the environment (read once per process):
    CB_BF16_GEMM=regs|64x2|64x3|64x4|128x2|128x3|128x4 python tools/bench_bf16_gemm.py
Prints one line per shape and a FLOP-weighted summary; appends a JSON record to gpurun_out/bf16_gemm_variants.jsonl."""
import ctypes as C
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from celebbasis_amd import lib  # noqa: E402

DEV = torch.device("cuda:0")
BF = torch.bfloat16
CONVS = [  # (N, H, W, Cin, Cout, count per step fwd+dgrad)  UNet (bs=4) then VAE encoder
    (4, 64, 64, 320, 320, 14), (4, 64, 64, 640, 320, 4), (4, 32, 32, 640, 640, 12), (4, 32, 32, 1280, 640, 2),
    (4, 16, 16, 1280, 1280, 14), (4, 16, 16, 2560, 1280, 4), (4, 8, 8, 1280, 1280, 22), (4, 8, 8, 2560, 1280, 6),
    (4, 512, 512, 128, 128, 4), (4, 256, 256, 256, 256, 3), (4, 128, 128, 512, 512, 3), (4, 64, 64, 512, 512, 8)]
LINS = [  # (M, N, K, count)
    (16384, 320, 320, 30), (16384, 960, 320, 10), (16384, 2560, 320, 10), (16384, 320, 1280, 10),
    (4096, 640, 640, 30), (4096, 1920, 640, 10), (4096, 5120, 640, 10), (4096, 640, 2560, 10),
    (1024, 1280, 1280, 30), (1024, 3840, 1280, 10), (1024, 10240, 1280, 10), (1024, 1280, 5120, 10),
    (308, 24960, 768, 2)]


def timeit(fn, reps=20):
    for _ in range(3):
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
    variant = os.environ.get("CB_BF16_GEMM", "default")
    ctx = lib.Context(dtype="bf16")
    L = lib.load()
    if len(sys.argv) > 2 and sys.argv[1] == "--plan":          # round 6: --plan 256x256x1 pins a (tile, split) for every eligible shape
        bm, bn, sp = (int(v) for v in sys.argv[2].split("x"))
        lib.check(L.cb_debug_force_plan(bm, bn, sp))
        variant = f"plan{sys.argv[2]}"
    g = torch.Generator().manual_seed(0)
    rows, tot_fl, tot_t = [], 0.0, 0.0
    for (N, H, W, Ci, Co, cnt) in CONVS:
        x = torch.randn(N, H, W, Ci, generator=g).to(BF).to(DEV)
        w = (torch.randn(Co, Ci, 3, 3, generator=g) * (Ci * 9) ** -0.5).to(DEV)
        wp = torch.empty(Co * 9 * Ci, dtype=BF, device=DEV)
        lib.check(L.cb_pack_conv_weight(ctx.h, lib.ptr(w), lib.ptr(wp, BF), Co, Ci, 3, 0))
        b = torch.zeros(Co, device=DEV)
        y = torch.empty(N, H, W, Co, dtype=BF, device=DEV)
        fn = lambda: lib.check(L.cb_conv2d_fwd(ctx.h, lib.ptr(x, BF), lib.ptr(wp, BF), lib.ptr(b), None, None, lib.ptr(y, BF), N, H, W, Ci, Co, 3, 0))
        t = timeit(fn)
        fl = 2.0 * N * H * W * Co * Ci * 9
        rows.append(dict(op=f"conv {N}x{H}x{W} {Ci}->{Co}", us=round(t * 1e6, 1), tflops=round(fl / t / 1e12, 1), cnt=cnt))
        tot_fl += fl * cnt; tot_t += t * cnt
    for (M, Nn, K, cnt) in LINS:
        A = torch.randn(M, K, generator=g).to(BF).to(DEV); B = (torch.randn(Nn, K, generator=g) * K ** -0.5).to(BF).to(DEV)
        Cc = torch.empty(M, Nn, dtype=BF, device=DEV)
        d = lib.GemmDesc()
        d.A, d.B, d.C = A.data_ptr(), B.data_ptr(), Cc.data_ptr()
        d.M, d.N, d.K = M, Nn, K
        d.lda = d.ldb = K; d.ldc = Nn; d.batch = 1; d.alpha = 1.0; d.dtype = lib.DTYPE_BF16
        fn = lambda: lib.check(L.cb_gemm(ctx.h, C.byref(d)))
        t = timeit(fn)
        fl = 2.0 * M * Nn * K
        rows.append(dict(op=f"gemm {M}x{Nn}x{K}", us=round(t * 1e6, 1), tflops=round(fl / t / 1e12, 1), cnt=cnt))
        tot_fl += fl * cnt; tot_t += t * cnt
    for r in rows:
        print(f"{variant:8s} {r['op']:28s} {r['us']:9.1f} us {r['tflops']:8.1f} TF")
    summ = dict(variant=variant, weighted_tflops=round(tot_fl / tot_t / 1e12, 1), weighted_ms=round(tot_t * 1e3, 3), rows=rows)
    print(f"{variant:8s} FLOP-weighted over the step's GEMM mix: {summ['weighted_tflops']} TF  ({summ['weighted_ms']} ms)")
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "bf16_gemm_variants.jsonl"), "a") as f:
        f.write(json.dumps(summ) + "\n")
    os._exit(0)


if __name__ == "__main__":
    main()
