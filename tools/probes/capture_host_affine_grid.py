#!/usr/bin/env python
"""tools/probes/capture_host_affine_grid.py — round 6 (any host, no GPU needed): THIS host's F.affine_grid table of the reference's constant
face-warp matrix at 512 x 512 (meta_net.py:131-142, 253-257), stored as its difference from the host-independent restatement
oracle/replay.py::fixture_host_warp_grid (the table of the host that generated tests/golden/horizon_sd14_*.pt): the bmm inside
torch's CPU affine_grid rounds host dependently (tools/probes/probe_affine_grid_host.py). Output: an .npz with the packed mask of
differing elements, this host's values there, and digests — the fixture tests/golden/warp_grid_gpubox_512.npz is one such file,
captured on the GPU box, so that an oracle trajectory on the PRODUCT'S DEFAULT table of that host can be generated elsewhere."""
import hashlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.replay import fixture_host_warp_grid  # noqa: E402
from celebbasis_amd.lib import WARP_TRANS_MATRIX  # noqa: E402


def main(out):
    S = 512
    M = torch.tensor([WARP_TRANS_MATRIX], dtype=torch.float32)
    mine = torch.nn.functional.affine_grid(M, size=[1, 3, S, S], align_corners=True)[0].contiguous().numpy()
    ref = fixture_host_warp_grid(S).numpy()
    mask = mine.view(np.int32) != ref.view(np.int32)
    err = np.abs(mine.astype(np.float64) - ref.astype(np.float64)).max()
    assert err <= 2e-7, f"unexpected distance {err}"                      # one ulp of a coordinate in [-1.1, 1.1]
    np.savez_compressed(out, mask=np.packbits(mask.reshape(-1)), values=mine[mask].astype(np.float32), S=S,
                        sha1_host=hashlib.sha1(mine.tobytes()).hexdigest()[:12], sha1_fixture_host=hashlib.sha1(ref.tobytes()).hexdigest()[:12],
                        cpu=open("/proc/cpuinfo").read().split("model name")[1].split("\n")[0].strip(": \t") if os.path.exists("/proc/cpuinfo") else "?")
    print(f"{mask.sum()} of {mask.size} elements differ ({100.0 * mask.mean():.2f} %), max {err:.3e}; host table sha1 "
          f"{hashlib.sha1(mine.tobytes()).hexdigest()[:12]} -> {out}")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "warp_grid_host_512.npz"))
