"""Multi-GPU layout of the personalization job (SURVEY.md §8 e): identities are independent optimisation problems, so they
shard one-per-process with NO data-path collective; the frozen weights are replicated. The only exchange is the checkpoint
merge: an all_gather of each rank's (es, h, K) coefficient blocks (4 KiB per identity) so that rank 0 can write the
reference's `embeddings_gs-*.pt` (embedding_manager.py:396-410). backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU tests."""
import torch
import torch.distributed as dist


def identities_for_rank(n_identities, rank, world):
    """identity i -> rank i mod world (round-robin keeps per-rank counts within 1 of each other)"""
    return [i for i in range(n_identities) if i % world == rank]


def gather_identity_blocks(local_ids, local_blocks, n_identities, device=None):
    """all_gather the per-identity coefficient blocks. local_blocks: [len(local_ids), D] float32. Returns [n_identities, D]
    on every rank (rows of identities nobody owns stay zero). One collective per save; latency-bound (KiB payload)."""
    grouped = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size() if grouped else 1
    D = local_blocks.shape[1] if local_blocks.numel() else 0
    if not grouped:                       # no process group: a plain single-process run. (A one-rank GROUP still runs the collective.)
        out = torch.zeros(n_identities, D)
        for j, i in enumerate(local_ids):
            out[i] = local_blocks[j].cpu()
        return out
    backend = dist.get_backend()
    dev = device if device is not None else (torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu"))
    per = (n_identities + world - 1) // world
    dims = torch.tensor([D], dtype=torch.int64, device=dev)
    dist.all_reduce(dims, op=dist.ReduceOp.MAX)
    D = int(dims.item())
    pack = torch.zeros(per, D + 1, device=dev)          # column 0 carries identity id + 1 (0 = empty slot)
    for j, i in enumerate(local_ids):
        pack[j, 0] = i + 1
        pack[j, 1:] = local_blocks[j].to(dev)
    gathered = [torch.empty_like(pack) for _ in range(world)]
    dist.all_gather(gathered, pack)
    out = torch.zeros(n_identities, D)
    for g in gathered:
        g = g.cpu()
        for row in g:
            i = int(row[0].item()) - 1
            if i >= 0:
                out[i] = row[1:]
    return out
