#!/usr/bin/env python
"""probe (any host, CPU only): F.affine_grid of the reference's constant matrix at 512 x 512 is evaluated by torch's CPU linspace + bmm, whose
rounding is host dependent. Prints digests of the pieces and the distance to a host-independent restatement
(grid = fma(t2, 1, fma(t1, y, t0 * x)) on torch.linspace(-1, 1, S), evaluated in float64-emulated fused multiply-adds)."""
import hashlib
import numpy as np, torch, torch.nn.functional as F
f32 = np.float32
S = 512
TM = [[1.07695457, -0.03625215, -1.56352194 / 512], [0.03625215, 1.07695457, -5.32134629 / 512]]
sha = lambda a: hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()[:12]
def fma(a, b, c): return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)
lin = torch.linspace(-1, 1, S).numpy()
g = F.affine_grid(torch.tensor([TM], dtype=torch.float32), size=[1, 3, S, S], align_corners=True)[0].numpy()
th = np.array(TM, dtype=f32)
X = np.broadcast_to(lin[None, :], (S, S)).astype(f32); Y = np.broadcast_to(lin[:, None], (S, S)).astype(f32)
one = np.ones_like(X)
chain = np.stack([fma(np.full_like(X, th[r, 2]), one, fma(np.full_like(X, th[r, 1]), Y, (th[r, 0] * X).astype(f32))) for r in range(2)], -1)
# a host-independent linspace: start + step * i in float64, rounded once
lin64 = (-1.0 + (2.0 / (S - 1)) * np.arange(S)).astype(f32)
print("cpu", torch.backends.cpu.get_cpu_capability(), "threads", torch.get_num_threads())
print("linspace sha", sha(lin), "vs float64-rounded linspace: differing", int((lin != lin64).sum()))
print("affine_grid sha", sha(g), "; fma-chain on torch.linspace sha", sha(chain), "differing", int((chain != g).sum()), "max", float(np.abs(chain - g).max()))
ix = ((g + 1) * f32(0.5 * (S - 1))); ixc = ((chain + 1) * f32(0.5 * (S - 1)))
print("pixel-coordinate difference chain vs affine_grid: max", float(np.abs(ix - ixc).max()), "mean", float(np.abs(ix - ixc).mean()))
