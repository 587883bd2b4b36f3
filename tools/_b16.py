"""B = 16 (kMaxBatch) smoke on the GPU: two identical steps from the same state give the same loss bit for bit, everything finite."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from celebbasis_amd import lib, synthetic
dt = sys.argv[1] if len(sys.argv) > 1 else "bf16"
state = synthetic.synthetic_state(synthetic.trainer_shapes(), seed=0)
ctx = lib.Context(dtype=dt)
tr = lib.Trainer(ctx, lib.WeightTable(state))
b = synthetic.synthetic_batch(16, step=0, device="cuda:0", identity=1)
losses = []
for rep in range(2):
    tr.set_state("mlp.weight", state["embedding_manager.meta_id_net.stylegan_mlp.net.0.weight"]); tr.set_state("mlp.bias", state["embedding_manager.meta_id_net.stylegan_mlp.net.0.bias"])
    tr.set_state("step", torch.zeros(1))
    for n in ("adam.m_weight", "adam.v_weight"): tr.set_state(n, torch.zeros(1024 * 512))
    for n in ("adam.m_bias", "adam.v_bias"): tr.set_state(n, torch.zeros(1024))
    tr.step(b["image"], b["faces"], b["ids"], b["tokens"], b["t"], b["noise"], b["post_noise"], 5e-3 * 16)
    ctx.sync()
    losses.append(float(tr.get_state("loss")[0]))
    assert torch.isfinite(tr.get_state("mlp.weight")).all() and torch.isfinite(tr.get_state("mlp.grad_weight")).all()
print("B16", dt, losses, "arena GiB", ctx.stats()["arena_peak_bytes"] / 2**30 if isinstance(ctx.stats(), dict) and "arena_peak_bytes" in ctx.stats() else ctx.stats())
assert losses[0] == losses[1] and 0.5 < losses[0] < 3.0
