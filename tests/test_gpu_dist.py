"""The collective path of the job on hardware (SURVEY §8 e): the ONE exchange is the checkpoint-merge all_gather. A one-GPU box
cannot take a scaling curve, but it can execute every line of that path on RCCL: a one-rank `nccl` process group, the gather on
CUDA tensors, `save_merged` writing the reference's embeddings_gs-*.pt (embedding_manager.py:396-410), compared with the file a
plain single-process run writes. Runs in a subprocess with a timeout so that an RCCL problem cannot hang the suite."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
out = {out!r}
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
from celebbasis_amd import sharding
from celebbasis_amd.main import save_merged
from test_gpu_models import mini_latent_diffusion, dv
from oracle.replay import make_batch
model, m, W = mini_latent_diffusion(seed=43)
model.learning_rate = 5e-3
model.train()
b = make_batch(2, 64, 16, 40000, 48136, 600, 10)
batch = dict(image=dv(b["image"]), tokens=b["tokens"], t=b["t"], noise=b["noise"], post_noise=b["post_noise"],
             image_ori=dict(faces=b["faces"], ids=b["ids"], num_ids=torch.ones(2, dtype=torch.int64)))
model.training_step(batch, 0)
# (a) the file a plain single-process run writes
save_merged(model, os.path.join(out, "single"), list(range(10)), list(range(10)), 0, 1, sharding)
# (b) the same state through a one-rank RCCL group: gather on CUDA tensors, then the merged file
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
blocks = torch.stack([torch.full((1024,), float(i + 1), device=dev) for i in (0, 3, 7)])
got = sharding.gather_identity_blocks([0, 3, 7], blocks, 10, device=dev)
assert got.shape == (10, 1024)
for i in range(10):
    assert torch.all(got[i] == (i + 1 if i in (0, 3, 7) else 0)), (i, got[i][:3])
save_merged(model, os.path.join(out, "rccl"), list(range(10)), list(range(10)), 0, 1, sharding)
dist.barrier()
dist.destroy_process_group()
open(os.path.join(out, "ok"), "w").write("ok")
"""


def test_one_rank_rccl_merge_writes_the_single_process_checkpoint(built_lib, tmp_path):
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, out=str(tmp_path), port=port))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0 and (tmp_path / "ok").exists(), (r.stdout[-1500:], r.stderr[-3000:])
    one = torch.load(tmp_path / "single" / "embeddings_gs-1.pt", weights_only=False)
    two = torch.load(tmp_path / "rccl" / "embeddings_gs-1.pt", weights_only=False)
    assert list(one.keys()) == list(two.keys()) == ["id_coefficients"] and len(two["id_coefficients"]) == 10
    for x, y in zip(one["id_coefficients"], two["id_coefficients"]):
        assert x.shape == y.shape == (2, 1, 16) and torch.equal(x, y)      # the merge moved every block through RCCL bit-exactly
    assert not torch.equal(one["id_coefficients"][1], one["id_coefficients"][2])
