#!/usr/bin/env python
"""probe (GPU box): where the HIP bilinear resize and torch's CPU F.interpolate(align_corners=True) differ bit-wise, and whether a
numpy restatement of torch's operation order agrees with torch ON THIS HOST."""
import os, sys
import numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from celebbasis_amd import lib
from oracle import cb_oracle as O
f32 = np.float32
def fma(a, b, c): return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)
def mul(a, b): return (a * b).astype(f32)
ctx = lib.Context(); L = lib.load()
for S in (96, 512):
    Bf = 2
    faces = torch.rand(Bf, S, S, 6, generator=torch.Generator().manual_seed(5)) * 2 - 1
    wref, sref = O.face_warp_resize(torch.cat(faces.chunk(2, -1), 0), 112)
    ctx.ensure_warp_grid(S)
    warped = torch.empty(2 * Bf, S, S, 3, device="cuda"); small = torch.empty(2 * Bf, 112, 112, 3, device="cuda")
    fd = faces.cuda()
    lib.check(L.cb_face_warp_resize(ctx.h, lib.ptr(fd), lib.ptr(warped), lib.ptr(small), Bf, S, 2, 112)); ctx.sync()
    hip = small.permute(0, 3, 1, 2).cpu().numpy(); ref = wref.numpy(); tor = sref.numpy()
    H, nn = S, 112
    scale = f32(H - 1) / f32(nn - 1)
    fy = (scale * np.arange(nn, dtype=f32)).astype(f32); y0 = np.minimum(fy.astype(np.int64), H - 1); y1 = np.minimum(y0 + 1, H - 1)
    l1 = np.clip((fy - y0.astype(f32)).astype(f32), 0, 1); l0 = (f32(1) - l1).astype(f32)
    A = ref[:, :, y0][:, :, :, y0]; B = ref[:, :, y0][:, :, :, y1]; Cq = ref[:, :, y1][:, :, :, y0]; D = ref[:, :, y1][:, :, :, y1]
    bb = lambda v: np.broadcast_to(v, A.shape).astype(f32)
    lx0 = l0[None, None, None, :]; lx1 = l1[None, None, None, :]; ly0 = l0[None, None, :, None]; ly1 = l1[None, None, :, None]
    emu = fma(fma(A, bb(lx0), mul(B, lx1)), bb(ly0), mul(fma(Cq, bb(lx0), mul(D, lx1)), ly1))
    print(f"S={S}: warp equal {torch.equal(warped.permute(0,3,1,2).cpu(), wref)}; hip!=torch {(hip != tor).mean():.4f} max {np.abs(hip-tor).max():.3g}; "
          f"emu!=torch {(emu != tor).mean():.4f} max {np.abs(emu-tor).max():.3g}; hip!=emu {(hip != emu).mean():.4f} max {np.abs(hip-emu).max():.3g}")
    bad = np.argwhere(hip != tor)
    if len(bad):
        ys = np.unique(bad[:, 2]); xs = np.unique(bad[:, 3])
        print("   mismatching output rows", ys[:40], "cols", xs[:40])
    print("   torch threads", torch.get_num_threads(), torch.backends.cpu.get_cpu_capability(), torch.__config__.parallel_info().split("\n")[0])
