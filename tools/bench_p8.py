#!/usr/bin/env python
"""tools/bench_p8.py — round 6: the phase-pipelined 256 x 256 bf16 tile (cb_gemm_bf16_p8.hip) on a few large GEMM / conv shapes,
pinned with cb_debug_force_plan(256, 256, 1), against the planner's own choice. The schedule variant / ablation is chosen by the
environment (read once per process): CB_P8_SCHED=0|1, CB_P8_ABL=1..4 (ablations compute WRONG results: timing only).
    python tools/bench_p8.py [tag]     -> one JSON line appended to gpurun_out/p8_bench.jsonl"""
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
GEMMS = [(4096, 4096, 4096), (8192, 8192, 8192), (16384, 1280, 1280), (65536, 512, 4608), (262144, 256, 2304), (16384, 2560, 320), (524288, 128, 1152)]
CONVS = [(4, 128, 128, 512, 512), (4, 256, 256, 256, 256), (4, 64, 64, 512, 512), (4, 512, 512, 128, 128), (4, 256, 256, 128, 256)]


def timeit(fn, reps=10):
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
    tag = sys.argv[1] if len(sys.argv) > 1 else "p8"
    pinned = tag != "planner"
    ctx = lib.Context(dtype="bf16")
    L = lib.load()
    if pinned:
        if tag.startswith("p8w"):
            lib.check(L.cb_debug_force_plan(512, 128, 1))          # the 512 x 128 geometry
        else:
            lib.check(L.cb_debug_force_plan(256, 256, 1))
    g = torch.Generator().manual_seed(0)
    rows = []
    for (M, N, K) in GEMMS:
        A = (torch.rand(M, K, generator=g) * 2 - 1).to(BF).to(DEV); B = (torch.rand(N, K, generator=g) * 2 - 1).to(BF).to(DEV)
        if os.environ.get("CB_P8_ZERO") == "1":                  # DVFS probe: zero operands toggle no MFMA datapath bits
            A.zero_(); B.zero_()
        Cc = torch.empty(M, N, dtype=BF, device=DEV)
        d = lib.GemmDesc()
        d.A, d.B, d.C = A.data_ptr(), B.data_ptr(), Cc.data_ptr()
        d.M, d.N, d.K = M, N, K
        d.lda = d.ldb = K; d.ldc = N; d.batch = 1; d.alpha = 1.0; d.dtype = lib.DTYPE_BF16
        t = timeit(lambda: lib.check(L.cb_gemm(ctx.h, C.byref(d))))
        rows.append(dict(op=f"gemm {M}x{N}x{K}", us=round(t * 1e6, 1), tflops=round(2.0 * M * N * K / t / 1e12, 1)))
        del A, B, Cc
    for (N, H, W, Ci, Co) in ([] if os.environ.get("CB_P8_ZERO") == "1" else CONVS):
        x = torch.randn(N, H, W, Ci, generator=g).to(BF).to(DEV)
        w = (torch.randn(Co, Ci, 3, 3, generator=g) * (Ci * 9) ** -0.5).to(DEV)
        wp = torch.empty(Co * 9 * Ci, dtype=BF, device=DEV)
        lib.check(L.cb_pack_conv_weight(ctx.h, lib.ptr(w), lib.ptr(wp, BF), Co, Ci, 3, 0))
        b = torch.zeros(Co, device=DEV)
        y = torch.empty(N, H, W, Co, dtype=BF, device=DEV)
        t = timeit(lambda: lib.check(L.cb_conv2d_fwd(ctx.h, lib.ptr(x, BF), lib.ptr(wp, BF), lib.ptr(b), None, None, lib.ptr(y, BF), N, H, W, Ci, Co, 3, 0)))
        rows.append(dict(op=f"conv {N}x{H}x{W} {Ci}->{Co}", us=round(t * 1e6, 1), tflops=round(2.0 * N * H * W * Co * Ci * 9 / t / 1e12, 1)))
    rec = dict(tag=tag, sched=os.environ.get("CB_P8_SCHED", "0"), abl=os.environ.get("CB_P8_ABL", "0"), rows=rows)
    print(tag, rec["sched"], rec["abl"], " ".join(f"{r['op'].split()[0][0]}{r['tflops']:.0f}" for r in rows))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "p8_bench.jsonl"), "a") as f:
        f.write(json.dumps(rec) + "\n")
    os._exit(0)


if __name__ == "__main__":
    main()
