#!/usr/bin/env python
"""tools/bench_flash.py — micro-benchmark of the fused attention kernels on the SD v1-4 UNet's shapes, through cb_attention
(op level, C ABI). Per shape: forward time and forward+backward time (HIP events around a batch of calls; the op-level entry adds
two small copies per call), TFLOP/s with the algorithmic FLOP counts of bench.py (forward 4*N*M*d*H, backward 10*N*M*d*H).

    python tools/bench_flash.py [--dtype f32|bf16] [--batch B] [--reps 20]      -> one JSON line per shape + a total
"""
import argparse
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

# (name, Nq, Nk, channels, heads, layers of this shape in one UNet pass)
SHAPES = [("self 64x64 d40", 4096, 4096, 320, 8, 5), ("cross 64x64 d40", 4096, 77, 320, 8, 5),
          ("self 32x32 d80", 1024, 1024, 640, 8, 5), ("cross 32x32 d80", 1024, 77, 640, 8, 5),
          ("self 16x16 d160", 256, 256, 1280, 8, 5), ("cross 16x16 d160", 256, 77, 1280, 8, 5),
          ("self 8x8 d160", 64, 64, 1280, 8, 1), ("cross 8x8 d160", 64, 77, 1280, 8, 1)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--only", default="", help="substring of the shape name")
    a = ap.parse_args()
    from celebbasis_amd import lib
    dev = torch.device("cuda:0")
    ctx = lib.Context(dtype=a.dtype)
    tdt = lib.torch_dtype(a.dtype)
    L = lib.load()
    B = a.batch
    tot = dict(fwd_ms=0.0, bwd_ms=0.0, fwd_gf=0.0, bwd_gf=0.0)
    for name, Nq, Nk, Cc, H, layers in SHAPES:
        if a.only and a.only not in name:
            continue
        self_attn = Nq == Nk and not name.startswith("cross")
        g = torch.Generator().manual_seed(1)
        ld = 3 * Cc if self_attn else Cc
        q = (torch.randn(B, Nq, ld, generator=g) * 0.5).to(dev).to(tdt)
        kv = q if self_attn else (torch.randn(B, Nk, 2 * Cc, generator=g) * 0.5).to(dev).to(tdt)
        out = torch.empty(B, Nq, Cc, device=dev, dtype=tdt); dout = torch.randn(B, Nq, Cc, device=dev).to(tdt)
        dq = torch.zeros_like(q); dkv = dq if self_attn else torch.zeros_like(kv)
        d = lib.AttentionDesc()
        d.qbuf, d.ldq, d.q_off = q.data_ptr(), ld, 0
        d.kvbuf, d.ldkv = kv.data_ptr(), (ld if self_attn else 2 * Cc)
        d.k_off, d.v_off = (Cc, 2 * Cc) if self_attn else (0, Cc)
        d.B, d.Nq, d.Nk, d.C, d.heads = B, Nq, Nk, Cc, H
        d.scale, d.causal = (Cc // H) ** -0.5, 0
        d.out = out.data_ptr(); d.dtype = lib.dtype_code(a.dtype)
        times = {}
        for mode in ("fwd", "fwdbwd"):
            if mode == "fwdbwd":
                d.dout, d.dqbuf, d.dkvbuf = dout.data_ptr(), dq.data_ptr(), dkv.data_ptr()
            for _ in range(3):
                lib.check(L.cb_attention(ctx.h, C.byref(d)))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(a.reps):
                lib.check(L.cb_attention(ctx.h, C.byref(d)))
            e1.record(); torch.cuda.synchronize()
            times[mode] = e0.elapsed_time(e1) / a.reps
        fl = 4.0 * B * H * Nq * Nk * (Cc // H) * 1e-9
        fwd, bwd = times["fwd"], times["fwdbwd"] - times["fwd"]
        row = dict(shape=name, B=B, fwd_us=round(fwd * 1e3, 1), bwd_us=round(bwd * 1e3, 1), fwd_tflops=round(fl / fwd, 1),
                   bwd_tflops=round(2.5 * fl / max(bwd, 1e-9), 1), layers=layers)
        print(json.dumps(row), flush=True)
        tot["fwd_ms"] += fwd * layers; tot["bwd_ms"] += bwd * layers; tot["fwd_gf"] += fl * layers; tot["bwd_gf"] += 2.5 * fl * layers
    print(json.dumps(dict(total_per_unet_pass=dict(fwd_ms=round(tot["fwd_ms"], 3), bwd_ms=round(tot["bwd_ms"], 3),
                                                   fwd_tflops=round(tot["fwd_gf"] / tot["fwd_ms"], 1),
                                                   bwd_tflops=round(tot["bwd_gf"] / tot["bwd_ms"], 1)), dtype=a.dtype, B=B)))
    del ctx


if __name__ == "__main__":
    main()
