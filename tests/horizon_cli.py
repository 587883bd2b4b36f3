"""Run one committed-horizon comparison from the command line (same code as the -m gpu horizon tests), so that build- or
environment-level switches can be A/B'd in separate processes:

    CB_BF16_KV_F32=1 python tests/horizon_cli.py bf16 20 kv32      -> gpurun_out/horizon_sd14_b4_bf16_kv32.json
    python tests/horizon_cli.py f32 100 base                       -> gpurun_out/horizon_sd14_b1_base.json

Test infrastructure (it imports the oracle's weight initialiser and replay stream); lives under tests/ for that reason."""
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE))


def main(which, steps, tag):
    from celebbasis_amd import build, lib
    from test_gpu_models import load, run_horizon
    build.build(verbose=False)
    bf = which == "bf16"
    ctx = lib.Context(dtype="bf16" if bf else "f32")
    fx = load(os.path.join(HERE, "golden"), "horizon_sd14_b4.pt" if bf else "horizon_sd14_b1.pt")
    rep = {}
    name = ("horizon_sd14_b4_bf16_" if bf else "horizon_sd14_b1_") + tag
    os.makedirs("gpurun_out", exist_ok=True)
    s = run_horizon(lib, ctx, fx, int(steps), name, rep, {}, save_traj=f"gpurun_out/{name}_traj.pt")
    s["env"] = {k: v for k, v in os.environ.items() if k.startswith("CB_")}
    print(json.dumps(s), flush=True)
    ctx.sync()


if __name__ == "__main__":
    main(*sys.argv[1:4])
