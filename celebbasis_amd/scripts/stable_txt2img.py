"""Counterpart of the reference's scripts/stable_txt2img.py for the CelebBasis test flow (02_start_test.sh): load the config
+ SD checkpoint + trained `embeddings_gs-*.pt`, build the conditioning of every prompt with the eval-mode embedding manager
(`test_mode: coefficient`), run DDIM with classifier-free guidance on the device and decode to images.

    python -m celebbasis_amd.scripts.stable_txt2img --config configs/stable-diffusion/aigc_id.yaml --ckpt sd-v1-4.ckpt \\
        --embedding_path logs/<run>/checkpoints/embeddings_gs-800.pt --celeb_basis weights/celeb_basis.pt \\
        --from-file infer_images/example_prompt.txt --eval_id1 0 --ddim_steps 50 --n_samples 8 --scale 10 --outdir outputs/

Flags keep the reference's names and defaults (stable_txt2img.py:60-230). `test_mode: image` takes the identities from face
images (--face_img1 / --face_img2: the reference hard-codes its evaluation-dataset paths, :272-299). PLMS and the evaluation
datasets are outside the CelebBasis hot path. The arithmetic all runs in libcelebbasis_hip.so; this file is orchestration.
"""
import argparse
import os
import time

import numpy as np
import torch


def get_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--prompt", type=str, default="a photo of sks person")
    p.add_argument("--from-file", type=str, default=None, help="one prompt per line")
    p.add_argument("--outdir", type=str, default="outputs/txt2img-samples")
    p.add_argument("--skip_grid", action="store_true")
    p.add_argument("--skip_save", action="store_true")
    p.add_argument("--ddim_steps", type=int, default=50)
    p.add_argument("--ddim_eta", type=float, default=0.0)
    p.add_argument("--n_iter", type=int, default=1)
    p.add_argument("--H", type=int, default=512)
    p.add_argument("--W", type=int, default=512)
    p.add_argument("--C", type=int, default=4)
    p.add_argument("--f", type=int, default=8)
    p.add_argument("--n_samples", type=int, default=8)
    p.add_argument("--n_rows", type=int, default=0)
    p.add_argument("--scale", type=float, default=10.0)
    p.add_argument("--config", type=str, required=True)
    p.add_argument("--ckpt", type=str, required=True)
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--embedding_path", type=str, required=True)
    p.add_argument("--celeb_basis", type=str, default=None, help="celeb basis tensor (es, 1+K, 768) when the checkpoint has none")
    p.add_argument("--neg_prompt", type=str, default="")
    p.add_argument("--fixed_code", action="store_true")
    p.add_argument("--eval_id1", type=int, default=0)
    p.add_argument("--eval_id2", type=int, default=-1)
    p.add_argument("--img_suffix", type=str, default="")
    p.add_argument("--face_img1", type=str, default=None, help="`test_mode: image`: face image of the first person (stable_txt2img.py:290-308)")
    p.add_argument("--face_img2", type=str, default=None, help="face image of the second person (defaults to --face_img1)")
    p.add_argument("--precision", choices=["f32", "bf16"], default="f32",
                   help="UNet / VAE element type (the reference samples under fp16 autocast, stable_txt2img.py:320-322; bf16 is the MI355X counterpart)")
    p.add_argument("--allow_offline_tokens", action="store_true", help="hashed stand-in token ids when the CLIP BPE vocabulary is not on disk (synthetic runs only)")
    return p


def make_grid(batch, nrow):
    """(n,3,H,W) in [0,1] -> (3, rows*H, nrow*W) (torchvision.utils.make_grid without padding)"""
    n, c, h, w = batch.shape
    rows = (n + nrow - 1) // nrow
    canvas = torch.zeros(c, rows * h, nrow * w)
    for i in range(n):
        r, q = divmod(i, nrow)
        canvas[:, r * h:(r + 1) * h, q * w:(q + 1) * w] = batch[i]
    return canvas


def main(argv=None):
    from PIL import Image
    import celebbasis_amd
    from celebbasis_amd.main import load_config, load_sd_checkpoint
    from celebbasis_amd.ldm.util import instantiate_from_config
    from celebbasis_amd.ldm.models.diffusion.ddim import DDIMSampler
    opt = get_parser().parse_args(argv)
    torch.manual_seed(opt.seed); np.random.seed(opt.seed)
    celebbasis_amd.install_ldm_alias()
    from celebbasis_amd.ldm.modules import frozen
    frozen.set_precision(opt.precision)
    cfg = load_config([opt.config])
    model = instantiate_from_config(cfg["model"])
    model.load_state_dict(load_sd_checkpoint(opt.ckpt), strict=False)
    if opt.celeb_basis:
        model.cond_stage_model.load_celeb_basis(torch.load(opt.celeb_basis, map_location="cpu", weights_only=False))
    model.embedding_manager.load(opt.embedding_path)
    model.cond_stage_model.allow_offline_tokens = bool(opt.allow_offline_tokens)
    model.eval()
    device = model.device
    sampler = DDIMSampler(model)
    os.makedirs(opt.outdir, exist_ok=True)
    sample_path = os.path.join(opt.outdir, "samples"); os.makedirs(sample_path, exist_ok=True)
    base_count, grid_count = len(os.listdir(sample_path)), len(os.listdir(opt.outdir)) - 1
    bs = opt.n_samples
    n_rows = opt.n_rows if opt.n_rows > 0 else bs
    prompts_all = [opt.prompt] if not opt.from_file else [l for l in open(opt.from_file).read().splitlines() if l.strip()]
    id2 = opt.eval_id2 if opt.eval_id2 >= 0 else opt.eval_id1
    faces = None
    if model.embedding_manager.test_mode == "image":        # stable_txt2img.py:290-308: the identities come from face images
        if not opt.face_img1:
            raise SystemExit("test_mode 'image' needs --face_img1 (and optionally --face_img2)")

        def pil_to_4d(path):                                  # transforms.Resize((512,512)) + ToTensor + Normalize(0.5, 0.5)
            img = Image.open(path).convert("RGB").resize((opt.W, opt.H), Image.BILINEAR)
            t = torch.from_numpy(np.asarray(img).copy()).float() / 127.5 - 1.0
            return t.unsqueeze(0).repeat(bs, 1, 1, 1).to(device)
        faces = torch.cat([pil_to_4d(opt.face_img1), pil_to_4d(opt.face_img2 or opt.face_img1)], dim=-1)
    # stable_txt2img.py:310-318: ALWAYS two identities — 'sks' takes ids[:, 0], 'ks' takes ids[:, 1] (= eval_id1 unless --eval_id2)
    image_ori = {"faces": faces, "ids": torch.tensor([opt.eval_id1, id2], device=device).unsqueeze(0).repeat(bs, 1),
                 "num_ids": torch.ones(bs, dtype=torch.long, device=device) * 2}
    shape = [opt.C, opt.H // opt.f, opt.W // opt.f]
    start_code = torch.randn([bs] + shape, device=device) if opt.fixed_code else None
    tic = time.time()
    with torch.no_grad(), model.ema_scope():
        for _ in range(opt.n_iter):
            for prompt in prompts_all:
                uc = model.get_learned_conditioning(bs * [opt.neg_prompt]) if opt.scale != 1.0 else None
                c = model.get_learned_conditioning(bs * [prompt], image_ori=image_ori)
                samples, _ = sampler.sample(S=opt.ddim_steps, conditioning=c, batch_size=bs, shape=shape, verbose=False,
                                            unconditional_guidance_scale=opt.scale, unconditional_conditioning=uc,
                                            eta=opt.ddim_eta, x_T=start_code)
                x = model.first_stage_model.decode(samples, scale_factor=model.scale_factor, to01=True).cpu()   # clamp((x+1)/2) on the device
                if not opt.skip_save:
                    for xi in x:
                        Image.fromarray((255.0 * xi.permute(1, 2, 0).numpy()).astype(np.uint8)).save(os.path.join(sample_path, f"{base_count:05}.jpg"))
                        base_count += 1
                if not opt.skip_grid:
                    grid = (255.0 * make_grid(x, n_rows).permute(1, 2, 0).numpy()).astype(np.uint8)
                    Image.fromarray(grid).save(os.path.join(opt.outdir, f'{grid_count:04}-{prompt.replace(" ", "-")}_{opt.img_suffix[:40]}.jpg'))
                    grid_count += 1
    print(f"Your samples are ready and waiting for you here: \n{opt.outdir} \n({time.time() - tic:.1f} s)")


if __name__ == "__main__":
    main()
