"""CLI mirror of the reference's trainer harness main_id_embed.py (SURVEY §2 'Trainer harness'): same flags for the
personalization run (`--base cfg.yaml -t --actual_resume sd.ckpt -n name --gpus 0, [--embedding_manager_ckpt …]`), the
reference YAML consumed unchanged (its `target:` strings resolve to celebbasis_amd.ldm.* through install_ldm_alias()),
the same LR rule lr = accum * ngpu * bs * base_lr (main_id_embed.py:767-786) and the same checkpoint cadence/format
(embeddings_gs-{step}.pt every `every_n_train_steps`). PyTorch-Lightning is replaced by a plain loop; multi-GPU runs are one
process per GPU (torchrun), identities sharded across ranks, coefficients merged with one all_gather at every save."""
import argparse
import os
import random

import torch
import yaml


def load_config(paths, dotlist=()):
    """OmegaConf.load/merge + `key=value` dotlist (main_id_embed.py:588-592) on plain dicts"""
    def merge(a, b):
        for k, v in b.items():
            if isinstance(v, dict) and isinstance(a.get(k), dict):
                merge(a[k], v)
            else:
                a[k] = v
        return a
    cfg = {}
    for p in paths:
        with open(p) as f:
            merge(cfg, yaml.safe_load(f))
    for item in dotlist:
        key, val = item.split("=", 1)
        node = cfg
        parts = key.split(".")
        for k in parts[:-1]:
            node = node.setdefault(k, {})
        node[parts[-1]] = yaml.safe_load(val)
    return cfg


class DataModuleFromConfig:
    """main_id_embed.py:217-292: `data.params.{batch_size, train, validation}` -> dataloaders"""

    def __init__(self, batch_size, train=None, validation=None, test=None, num_workers=0, wrap=False, **ignored):
        from .ldm.util import instantiate_from_config
        self.batch_size, self.num_workers = batch_size, num_workers
        self.datasets = {}
        if train is not None:
            self.datasets["train"] = instantiate_from_config(train)
        if validation is not None:
            self.datasets["validation"] = instantiate_from_config(validation)

    @staticmethod
    def collate(samples):
        out = dict(image=torch.stack([s["image"] for s in samples]), caption=[s["caption"] for s in samples],
                   id_idx=torch.tensor([s["id_idx"] for s in samples]), img_idx=torch.tensor([s["img_idx"] for s in samples]))
        out["image_ori"] = dict(faces=torch.stack([s["image_ori"]["faces"] for s in samples]),
                                ids=torch.stack([s["image_ori"]["ids"] for s in samples]),
                                num_ids=torch.tensor([s["image_ori"]["num_ids"] for s in samples]))
        return out

    def train_dataloader(self):
        return torch.utils.data.DataLoader(self.datasets["train"], batch_size=self.batch_size, shuffle=True,
                                           num_workers=self.num_workers, collate_fn=self.collate)

    def device_feeder(self, device, tokenize=None):
        """the default feeder of `celebbasis_amd.main` on a GPU: same dataset object, same draws, pixel work on the device
        (ldm/data/face_id.py DeviceFeeder) instead of in `num_workers` CPU processes"""
        from .ldm.data.face_id import DeviceFeeder
        return DeviceFeeder(self.datasets["train"], self.batch_size, device, shuffle=True, tokenize=tokenize)


def get_parser():
    p = argparse.ArgumentParser(description="CelebBasis personalization on MI355X (flags of main_id_embed.py:47-178)")
    p.add_argument("-n", "--name", type=str, default="celebbasis")
    p.add_argument("-b", "--base", nargs="*", default=[], metavar="base_config.yaml")
    p.add_argument("-t", "--train", action="store_true")
    p.add_argument("-s", "--seed", type=int, default=23)
    p.add_argument("-l", "--logdir", type=str, default="logs")
    p.add_argument("--pipeline", action="store_true",
                   help="overlap the next batch's VAE encode / FR backbone (gradient-free first stage) with the current step on the prefetch "
                        "stream — bitwise the same results; worth ~2 % in fp32 and ~4 % in bf16 (profiles/README.md, round 3)")
    p.add_argument("--actual_resume", type=str, default=None, help="Stable Diffusion checkpoint (.ckpt / .safetensors)")
    p.add_argument("--embedding_manager_ckpt", type=str, default="")
    p.add_argument("--celeb_basis", type=str, default=None, help="celeb_basis.pt (es, 1+K, 768), the format written by extract_pt.py")
    p.add_argument("--fr_ckpt", type=str, default="./weights/glint360k_cosface_r100_fp16_0.1/backbone.pth")
    p.add_argument("--data_root", type=str, default=None)
    p.add_argument("--gpus", type=str, default="0,")
    p.add_argument("--max_steps", type=int, default=None)
    p.add_argument("--synthetic", action="store_true",
                   help="seeded random weights at the true shapes, seeded stand-in faces and hashed pseudo token ids (no checkpoints, "
                        "dataset or BPE vocabulary needed); without it missing inputs are errors")
    p.add_argument("--precision", choices=["f32", "bf16"], default="f32",
                   help="element type of the frozen UNet / VAE: f32 = the reference's arithmetic (BASELINE config 2), "
                        "bf16 = bf16 storage + bf16 MFMA with fp32 statistics (config 3); the CLIP text tower, the FR-net, the "
                        "coefficient head, its gradient and AdamW are fp32 in both")
    p.add_argument("--cpu_feeder", action="store_true",
                   help="feed from a torch DataLoader with `num_workers` CPU processes like the reference (main_id_embed.py:252-260) "
                        "instead of the device feeder (augmentations on the GPU, the default)")
    p.add_argument("--report", type=str, default=None,
                   help="write a JSON line with the wall-clock steps/s of this run (feeder + tokenizer + H2D + step, everything "
                        "the loop does) measured after `--report_warmup` steps")
    p.add_argument("--report_warmup", type=int, default=20)
    return p


def load_sd_checkpoint(path):
    """main_id_embed.py:25-44: .ckpt (torch.load -> ['state_dict']) or .safetensors"""
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(path)
    sd = torch.load(path, map_location="cpu", weights_only=False)
    return sd.get("state_dict", sd)


def main(argv=None):
    opt, unknown = get_parser().parse_known_args(argv)
    from . import install_ldm_alias, sharding
    install_ldm_alias()
    from .ldm.util import instantiate_from_config
    from .ldm.modules import frozen
    frozen.set_precision(opt.precision)
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    random.seed(opt.seed); torch.manual_seed(opt.seed)                      # seed_everything(opt.seed) main_id_embed.py:586
    config = load_config(opt.base, [u for u in unknown if "=" in u])
    if not torch.cuda.is_available():
        raise SystemExit("celebbasis_amd.main needs an MI355X (gfx950); there is no CPU training path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)
    model = instantiate_from_config(config["model"])
    if opt.synthetic:
        from . import synthetic
        model.cond_stage_model.allow_offline_tokens = True       # hashed pseudo ids when the BPE vocabulary is not on disk
        model.load_state_dict(synthetic.synthetic_state(synthetic.trainer_shapes(), seed=0))
    else:
        sd = load_sd_checkpoint(opt.actual_resume)
        fr = torch.load(opt.fr_ckpt, map_location="cpu", weights_only=False)
        sd.update({"embedding_manager.meta_id_net.id_model." + k: v.float() for k, v in fr.items()})
        if opt.celeb_basis:
            sd["celeb_embeddings"] = torch.load(opt.celeb_basis, map_location="cpu", weights_only=False)
        model.load_state_dict(sd, strict=False)
    if opt.embedding_manager_ckpt:
        model.embedding_manager.load(opt.embedding_manager_ckpt)
    data_cfg = config["data"]["params"]
    if opt.data_root:
        for split in ("train", "validation"):
            if split in data_cfg:
                data_cfg[split]["params"]["pickle_path"] = os.path.join(opt.data_root, "ffhq.pickle")
    if opt.synthetic:
        for split in ("train", "validation"):
            if split in data_cfg:
                data_cfg[split]["params"]["synthetic"] = True
    # identity sharding: each rank personalises its own subset of the DENSE identity slots 0..num_ids-1 the reference's dataset
    # assigns (face_id.py:556-583), so every rank writes the same id_coefficients[slot] a single-process run would (SURVEY §8 e);
    # the frozen weights are replicated. NOTE (documented, not emulated): the reference with num_ids > 1 trains ONE shared MLP on
    # all identities under DDP; here every rank trains its own MLP on its own identities — N independent num_ids=1-style runs.
    n_slots = int(data_cfg["train"]["params"].get("num_ids", 1))
    all_ids = list(range(n_slots))
    mine = [all_ids[i] for i in sharding.identities_for_rank(len(all_ids), rank, world)]
    if world > 1:
        if model.embedding_manager.test_mode == "image":
            raise SystemExit("test_mode 'image' saves the MLP itself; with identity sharding every rank holds a different MLP — "
                             "run it on one GPU (or use test_mode 'coefficient' / 'embedding')")
        data_cfg["train"]["params"]["only_slots"] = mine or all_ids[:1]
    data = DataModuleFromConfig(**data_cfg)
    bs, base_lr = data_cfg["batch_size"], config["model"]["base_learning_rate"]
    model.learning_rate = 1 * 1 * bs * base_lr       # accumulate_grad_batches * ngpu(per identity) * bs * base_lr
    lcfg = config.get("lightning", {})
    max_steps = opt.max_steps or lcfg.get("trainer", {}).get("max_steps", 800)
    every = lcfg.get("modelcheckpoint", {}).get("params", {}).get("every_n_train_steps", 200)
    ckptdir = os.path.join(opt.logdir, opt.name, "checkpoints")
    model.train()
    step = 0
    if opt.cpu_feeder:
        loader = data.train_dataloader
    else:
        feeder = data.device_feeder(dev, tokenize=model.cond_stage_model.tokenize)
        loader = lambda: feeder
    import time
    t_mark, step_mark = None, 0

    def lookahead(loader):
        """(batch, next batch or None): with --pipeline the next batch's first stage is enqueued before the current step"""
        it = iter(loader)
        cur = next(it, None)
        while cur is not None:
            nxt = next(it, None)
            yield cur, nxt
            cur = nxt

    while step < max_steps:
        for batch, nxt in lookahead(loader()):
            batch["image"] = batch["image"].to(dev)
            if opt.pipeline and nxt is not None:
                nxt["image"] = nxt["image"].to(dev)
                model.prefetch_first_stage(nxt)
            model.training_step(batch, step)
            step += 1
            if opt.report and step == min(opt.report_warmup, max_steps - 1):
                torch.cuda.synchronize(dev); t_mark, step_mark = time.perf_counter(), step
            if step % every == 0 or step == max_steps:
                save_merged(model, ckptdir, all_ids, mine, rank, world, sharding)
                if rank == 0:
                    print(f"[celebbasis_amd] step {step}/{max_steps} loss {model.last_loss():.5f} -> {ckptdir}", flush=True)
            if step >= max_steps:
                break
    if opt.report and t_mark is not None and rank == 0:
        import json
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t_mark
        rec = dict(what="python -m celebbasis_amd.main wall clock: feeder + tokenizer + H2D + step + checkpoint writes",
                   feeder="cpu_dataloader" if opt.cpu_feeder else "device", num_workers=data.num_workers if opt.cpu_feeder else 0,
                   precision=opt.precision, batch_size=bs, steps_timed=step - step_mark, seconds=dt,
                   steps_per_sec=(step - step_mark) / dt, samples_per_sec=(step - step_mark) * bs / dt, pipeline=bool(opt.pipeline),
                   world=world)
        os.makedirs(os.path.dirname(os.path.abspath(opt.report)), exist_ok=True)
        with open(opt.report, "w") as f:
            f.write(json.dumps(rec) + "\n")
        print("[celebbasis_amd] " + json.dumps(rec), flush=True)


def save_merged(model, ckptdir, all_ids, mine, rank, world, sharding):
    """gather every rank's identity blocks — coefficients (4 KiB each) and EMA embeddings (6 KiB each), so that both
    `test_mode: coefficient` and `test_mode: embedding` checkpoints are complete — and let rank 0 write the reference format"""
    model.sync_manager_from_trainer()
    em = model.embedding_manager
    import torch.distributed as dist
    if world > 1 or (dist.is_available() and dist.is_initialized()):
        nc, ne = em.num_es * em.meta_heads * em.inner_dim, em.num_es * em.meta_heads * em.token_dim
        blocks = torch.stack([torch.cat([em.id_coefficients[i].reshape(-1), em.id_embeddings[i].reshape(-1)]) for i in mine]) \
            if mine else torch.zeros(0, nc + ne)
        merged = sharding.gather_identity_blocks(mine, blocks, em.max_ids)
        for i in all_ids:
            em.id_coefficients[i] = merged[i, :nc].reshape(em.num_es, em.meta_heads, em.inner_dim)
            em.id_embeddings[i] = merged[i, nc:].reshape(em.num_es * em.meta_heads, em.token_dim)
    if rank == 0:
        os.makedirs(ckptdir, exist_ok=True)
        em.save(os.path.join(ckptdir, "embeddings.pt"))
        em.save(os.path.join(ckptdir, f"embeddings_gs-{model.global_step}.pt"))


if __name__ == "__main__":
    main()
