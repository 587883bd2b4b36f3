"""CPU tests (no GPU): the C-ABI library loads and exports every declared symbol, the host logic (weight-name tables,
graph construction, arena sizing, FLOP accounting) runs on a DRY context, the reference YAML instantiates the host mirror
classes unchanged, the checkpoint format matches the reference's, and the multi-GPU merge works over gloo (world_size 2)."""
import os
import re
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_PARAMS = os.path.join(ROOT, "tests", "golden", "aigc_id_params.json")   # parameter values of the reference config (data)


def test_abi_exports_every_declared_symbol(built_lib):
    from celebbasis_amd import lib
    hdr = open(os.path.join(ROOT, "include", "celebbasis_hip.h")).read()
    declared = set(re.findall(r"\b(cb_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    missing = [s for s in sorted(declared) if not hasattr(built_lib, s)]
    assert not missing, f"declared in the header but not exported: {missing}"
    assert declared == set(lib.ABI_SYMBOLS), declared ^ set(lib.ABI_SYMBOLS)
    assert built_lib.cb_version() == 303          # CB_ABI_VERSION of include/celebbasis_hip.h (struct_size-guarded descriptors)


def test_integration_stub_matches_the_library(built_lib):
    """INTEGRATION.md §2 shows the reference-side ctypes stub a maintainer would add. Its code block is executed here against
    the built library (no GPU: definitions only + the calls that need no device): the version it asserts is the library's, every
    entry point it names is exported, its `cb_batch` has the layout of the typed binding, its opaque 512-byte `cb_trainer_cfg`
    buffer is large enough and `cb_trainer_cfg_default` writes the struct_size the library will check."""
    import ctypes as C
    import re
    from celebbasis_amd import lib
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = re.search(r"```python\n(# ldm/models/diffusion/native_step.py.*?)```", text, re.S).group(1)
    block = block.replace('C.CDLL("libcelebbasis_hip.so")', f"C.CDLL({built_lib._name!r})")
    ns = {}
    exec(compile(block, "INTEGRATION.md:native_step", "exec"), ns)          # runs the stub's own `assert lib.cb_version() == 303`
    so = ns["lib"]
    for name in set(re.findall(r"lib\.(cb_[a-z0-9_]+)", block)):
        assert hasattr(so, name), name
    stub_batch = ns["cb_batch"]
    assert [(n, t) for n, t in stub_batch._fields_] == [(n, t) for n, t in lib.Batch._fields_]
    assert C.sizeof(stub_batch) == C.sizeof(lib.Batch)
    assert C.sizeof(lib.TrainerCfg) <= 512
    cfg = (C.c_byte * 512)()
    ns["check"](so.cb_trainer_cfg_default(cfg))
    assert C.cast(cfg, C.POINTER(C.c_int))[0] == C.sizeof(lib.TrainerCfg)
    with pytest.raises(RuntimeError):                              # the stub's error path: status -> cb_last_error() text
        ns["check"](so.cb_trainer_create(None, None, None, None))


def test_header_structs_and_ctypes_binding_agree(tmp_path):
    """the C header is what a C / cgo / JNI caller compiles against, celebbasis_amd/lib.py is what the tests call through: a plain-C
    translation unit (gcc, no HIP) prints sizeof and every field offset of the header's structs, which must equal the ctypes
    classes' — field for field, by name. Also proves the header is self-contained C (no torch / HIP types in the signatures)."""
    import ctypes as C
    import shutil
    from celebbasis_amd import lib
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    pairs = [("cb_gemm_desc", lib.GemmDesc), ("cb_attention_desc", lib.AttentionDesc), ("cb_unet_cfg", lib.UNetCfg), ("cb_vae_cfg", lib.VaeCfg),
             ("cb_clip_cfg", lib.ClipCfg), ("cb_trainer_cfg", lib.TrainerCfg), ("cb_batch", lib.Batch)]
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "celebbasis_hip.h"', "int main(void) {"]
    for cname, cls in pairs:
        lines.append(f'  printf("{cname} %zu", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'  printf(" {fname}=%zu", offsetof({cname}, {fname}));')
        lines.append('  printf("\\n");')
    lines += ['  printf("version %d\\n", CB_ABI_VERSION);', "  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines) + "\n")
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.strip().splitlines()
    assert out[-1] == f"version {lib.ABI_VERSION}"
    for (cname, cls), line in zip(pairs, out):
        toks = line.split()
        assert toks[0] == cname and int(toks[1]) == C.sizeof(cls), (cname, toks[1], C.sizeof(cls))
        got = dict(t.split("=") for t in toks[2:])
        for fname, _ in cls._fields_:
            assert int(got[fname]) == getattr(cls, fname).offset, (cname, fname, got[fname], getattr(cls, fname).offset)


def test_errors_do_not_cross_the_boundary(built_lib):
    from celebbasis_amd import lib
    ctx = lib.Context(device=-1)
    wt = lib.WeightTable({"time_embed.0.weight": (5, 5)})
    with pytest.raises(lib.CBError) as e:
        lib.UNet(ctx, wt, lib.unet_cfg())
    assert "time_embed.0.weight" in str(e.value) and e.value.code == -3      # CB_STATUS_SHAPE, names the tensor
    with pytest.raises(lib.CBError) as e:
        lib.UNet(ctx, lib.WeightTable({}), lib.unet_cfg())
    assert e.value.code == -4                                                 # CB_STATUS_MISSING


def test_no_gpu_means_loud_failure_not_fallback(built_lib):
    from celebbasis_amd import lib
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="MI355X"):
        lib.Context()


def test_dry_training_step_accounting(built_lib):
    """the native graph at SD v1-4 shapes: FLOPs per stage equal the reference's FlopCounter probe"""
    from celebbasis_amd import lib, synthetic
    ctx = lib.Context(device=-1)
    wt = lib.WeightTable(synthetic.trainer_shapes())
    gf = lambda: ctx.stats()["gemm_flops"] / 1e9
    u = lib.UNet(ctx, wt, lib.unet_cfg(), prefix="model.diffusion_model.")
    x, t, c = torch.zeros(1, 4, 64, 64), torch.zeros(1, dtype=torch.int64), torch.zeros(1, 77, 768)
    ctx.reset_stats(); u.forward(x, t, c, need_grad=False); fwd = gf()
    ctx.reset_stats(); u.forward(x, t, c, need_grad=True); both = gf()
    assert abs(fwd - 803.3) < 0.5, fwd                       # UNet forward 803.3 GFLOP
    # backward to the context: 865.9 GFLOP by autograd; here +17.0 (the stride-2 dgrad gathers all 9 taps) and -37.8 (the three
    # Upsample layers' input gradients run as one 4x4 stride-2 conv over dy: 16 instead of 36 taps per input element) = 845.1
    assert abs((both - fwd) - 845.1) < 0.5, both - fwd
    v = lib.VAE(ctx, wt, lib.vae_cfg(), prefix="first_stage_model.")
    ctx.reset_stats(); v.encode(torch.zeros(1, 512, 512, 3)); assert abs(gf() - 1116.7) < 0.5
    f = lib.FrNet(ctx, wt, prefix="embedding_manager.meta_id_net.id_model.")
    ctx.reset_stats(); f.forward(torch.zeros(2, 112, 112, 3)); assert abs(gf() - 48.4) < 0.2
    tr = lib.Trainer(ctx, wt)
    ctx.reset_stats(); tr.step(1, None, None, None, None, None, None, 5e-3)
    st = ctx.stats()
    assert 2820 < st["gemm_flops"] / 1e9 < 2860             # 2.84 TFLOP / step (SURVEY §8 d: 2.86 by autograd's count, -37.8 G in the up-conv gradients)
    assert st["arena_peak_bytes"] < 40 * 2 ** 30 and st["pool_bytes"] < 24 * 2 ** 30
    ctx.reset_stats(); tr.step(4, None, None, None, None, None, None, 5e-3)
    assert ctx.stats()["gemm_flops"] / 1e9 > 4 * 2800        # scales with the batch


def test_dry_bf16_training_step_and_sampler(built_lib):
    """BASELINE config 3 on a dry context (no GPU): the bf16 graph passes every shape / alignment check of the bf16 kernels
    (k-contiguous operands, 8-element chunks, padded RGB / latent channels) at bs=4, accounts the same algorithmic FLOPs as fp32
    (+ the padded first / last convolutions) and needs about half the activation arena; dtype mismatches are rejected"""
    from celebbasis_amd import lib, synthetic
    from oracle import cb_oracle as O
    shapes = dict(synthetic.trainer_shapes()); shapes.update(synthetic.decoder_shapes())
    res = {}
    for dt in ("f32", "bf16"):
        ctx = lib.Context(device=-1, dtype=dt)
        tr = lib.Trainer(ctx, lib.WeightTable(shapes))
        assert tr.cfg.dtype == ctx.dtype
        ctx.reset_stats()
        tr.step(4, None, None, None, None, None, None, 2e-2)
        res[dt] = ctx.stats()
        u = tr.unet()
        s = O.ddim_schedule(O.make_schedule()["alphas_cumprod"], 2, 0.0)
        u.ddim_sample(torch.zeros(2, 4, 64, 64), torch.zeros(2, 77, 768), torch.zeros(2, 77, 768), s["timesteps"], s["alphas"],
                      s["alphas_prev"], s["sigmas"], s["sqrt_one_minus_alphas"], scale=10.0)
        lib.VAEDecoder(ctx, lib.WeightTable(shapes), lib.vae_cfg(), prefix="first_stage_model.").decode(torch.zeros(1, 4, 64, 64), 0.18215)
    f, b = res["f32"], res["bf16"]
    assert abs(b["gemm_flops"] / f["gemm_flops"] - 1.0) < 5e-3
    assert b["arena_peak_bytes"] < 0.6 * f["arena_peak_bytes"]
    ctx = lib.Context(device=-1, dtype="bf16")
    cfg = lib.trainer_cfg_default()
    tr = lib.Trainer.__new__(lib.Trainer)
    with pytest.raises(lib.CBError) as e:       # a cfg that says f32 on a bf16 context
        import ctypes as C
        h = C.c_void_p()
        lib.check(lib.load().cb_trainer_create(ctx.h, lib.WeightTable(shapes).h, C.byref(cfg), C.byref(h)))
    assert "dtype" in str(e.value)
    with pytest.raises(ValueError):
        lib.Context(device=-1, dtype="fp8")


def test_dry_sampler_accounting(built_lib):
    """sampling path at SD v1-4 shapes on a dry context: DDIM-50 with guidance = 50 batch-2 UNet passes (SURVEY §8 A10:
    50 x 2 x 803 GF), and the native VAE decoder graph does exactly the FLOPs torch's FlopCounter sees in the oracle"""
    from torch.utils.flop_counter import FlopCounterMode
    from celebbasis_amd import lib, synthetic
    from oracle import cb_oracle as O
    ctx = lib.Context(device=-1)
    shapes = dict(synthetic.trainer_shapes()); shapes.update(synthetic.decoder_shapes())
    wt = lib.WeightTable(shapes)
    gf = lambda: ctx.stats()["gemm_flops"] / 1e9
    u = lib.UNet(ctx, wt, lib.unet_cfg(), prefix="model.diffusion_model.")
    s = O.ddim_schedule(O.make_schedule()["alphas_cumprod"], 50, 0.0)
    z = torch.zeros(1, 4, 64, 64); c = torch.zeros(1, 77, 768)
    tabs = (s["timesteps"], s["alphas"], s["alphas_prev"], s["sigmas"], s["sqrt_one_minus_alphas"])
    ctx.reset_stats(); u.ddim_sample(z, c, c, *tabs, scale=10.0)
    assert abs(gf() - 50 * 2 * 803.27) < 5.0, gf()
    ctx.reset_stats(); u.ddim_sample(z, c, None, *tabs, scale=1.0)
    assert abs(gf() - 50 * 803.27) < 5.0, gf()
    d = lib.VAEDecoder(ctx, wt, lib.vae_cfg(), prefix="first_stage_model.")
    ctx.reset_stats(); img = d.decode(z, 0.18215)
    assert tuple(img.shape) == (1, 3, 512, 512)
    W = {k: torch.empty(v, device="meta") for k, v in O.vae_decoder_param_shapes(O.VAE_SD14).items()}
    with FlopCounterMode(display=False) as fc:
        O.vae_decode(W, O.VAE_SD14, torch.empty(1, 4, 64, 64, device="meta"))
    assert abs(gf() - fc.get_total_flops() / 1e9) < 0.5, (gf(), fc.get_total_flops() / 1e9)


def test_product_tables_match_oracle_and_reference_digests(golden_dir):
    import json
    from celebbasis_amd import synthetic
    from oracle import cb_oracle as O
    from oracle.gen_golden import keys_digest
    assert synthetic.trainer_shapes() == {k: tuple(v) for k, v in O.trainer_param_shapes().items()}
    from celebbasis_amd.ldm.modules.diffusionmodules.openaimodel import UNetModel
    with open(os.path.join(golden_dir, "state_dict_digests.json")) as f:
        d = json.load(f)
    assert keys_digest(UNetModel.param_shapes(synthetic.UNET_SD14)) == d["unet_sd14"]
    # the caption templates of the feeder are the reference's list (face_id.py:17-99), by digest
    import hashlib
    from celebbasis_amd.ldm.data.face_id import imagenet_templates_small
    assert len(imagenet_templates_small) == d["imagenet_templates_small"]["count"]
    assert hashlib.sha256("\n".join(imagenet_templates_small).encode()).hexdigest() == d["imagenet_templates_small"]["sha256"]


def _write_face_pickle(tmp_path, n, images_per_id=1):
    """n identities as PNG files named <stem>.png (+ a second image <stem>.1.png when images_per_id > 1) and the path-list pickle
    the reference's start_align.py / gen_pickle.py would have written"""
    import pickle
    import numpy as np
    from PIL import Image
    paths = []
    rs = np.random.RandomState(0)
    for i in range(n):
        for j in range(images_per_id):
            p = tmp_path / (f"{i:05d}.png" if j == 0 else f"{i:05d}.{j}.png")
            Image.fromarray(rs.randint(0, 255, (40, 40, 3), dtype=np.uint8)).save(p)
            paths.append(str(p))
    pk = tmp_path / "ffhq.pickle"
    with open(pk, "wb") as f:
        pickle.dump(paths, f)
    return str(pk)


def test_dataset_identity_slots_follow_the_reference(tmp_path):
    """face_id.py:556-583: images grouped by basename stem, specific_ids = walk positions, dense use_id slots in the batch;
    a missing pickle is an error (no silent synthetic stand-in) unless synthetic=True is passed"""
    from celebbasis_amd.ldm.data.face_id import FaceIdDatasetOneShot
    pk = _write_face_pickle(tmp_path, 5, images_per_id=2)
    ds = FaceIdDatasetOneShot(pk, num_ids=2, specific_ids=[1, 3], diff_cnt=0, image_size=32, repeats=3)
    assert [ds.img_dict[k]["id"] for k in (0, 1)] == ["00001", "00003"] and len(ds) == 6
    assert all(len(ds.img_dict[k]["images"]) == 1 for k in (0, 1))          # images_per_id = 1: the first image of each group
    seen = {int(ds[i]["image_ori"]["ids"][0]) for i in range(len(ds))}
    assert seen == {0, 1} and ds[0]["id_idx"] in (0, 1)
    assert tuple(ds[0]["image"].shape) == (32, 32, 3) and tuple(ds[0]["image_ori"]["faces"].shape) == (32, 32, 6)
    rng = FaceIdDatasetOneShot(pk, num_ids=3, specific_ids="1-4", diff_cnt=0, image_size=32)
    assert [rng.img_dict[k]["id"] for k in (0, 1, 2)] == ["00001", "00002", "00003"]
    shard = FaceIdDatasetOneShot(pk, num_ids=2, specific_ids=[1, 3], diff_cnt=0, image_size=32, only_slots=[1])
    assert {int(shard[i]["image_ori"]["ids"][0]) for i in range(4)} == {1}
    with pytest.raises(ValueError):
        FaceIdDatasetOneShot(pk, num_ids=4, specific_ids=[1, 3], diff_cnt=0)     # "Reach last. Not enough images"
    with pytest.raises((ValueError, FileNotFoundError)):
        FaceIdDatasetOneShot(str(tmp_path / "missing.pickle"), num_ids=1, diff_cnt=0)
    with pytest.raises(ValueError):
        FaceIdDatasetOneShot(None, num_ids=1, diff_cnt=0)
    syn = FaceIdDatasetOneShot(None, num_ids=2, specific_ids=[1, 2], diff_cnt=0, image_size=32, synthetic=True)
    assert sorted(syn.faces) == [0, 1]


def test_reference_yaml_is_drop_in(tmp_path):
    """the `model:` and `data:` sections of configs/stable-diffusion/aigc_id.yaml (kept as a data fixture) instantiate the
    host mirror through the reference's own target strings"""
    import celebbasis_amd
    celebbasis_amd.install_ldm_alias()
    from celebbasis_amd.main import load_config
    from celebbasis_amd.ldm.util import instantiate_from_config
    import json
    import yaml
    ref_yaml = tmp_path / "aigc_id.yaml"                     # the config as a YAML file again, as the CLI would receive it
    ref_yaml.write_text(yaml.safe_dump(json.load(open(REF_PARAMS))))
    pk = _write_face_pickle(tmp_path, 4)                     # 4 identities on disk; the YAML asks for walk positions [1, 2]
    cfg = load_config([str(ref_yaml)], ["data.params.batch_size=1", f"data.params.train.params.pickle_path={pk}",
                                         f"data.params.validation.params.pickle_path={pk}", "data.params.num_workers=0"])
    assert cfg["model"]["target"] == "ldm.models.diffusion.ddpm.LatentDiffusion"
    m = instantiate_from_config(cfg["model"])
    assert type(m).__name__ == "LatentDiffusion" and type(m.embedding_manager).__name__ == "EmbeddingManagerId"
    assert m.embedding_manager.string_to_token_dict["sks"] == 48136
    assert sum(p.numel() for p in m.embedding_manager.trainable_parameters()) == 525312
    assert m.embedding_manager.embedding_parameters() == [] and m.embedding_manager.embedding_neg_loss() == 0.0
    assert abs(float(m.sqrt_alphas_cumprod[500]) - 0.5256736) < 1e-6
    data = instantiate_from_config(cfg["data"])
    b = next(iter(data.train_dataloader()))
    assert tuple(b["image"].shape) == (1, 512, 512, 3) and tuple(b["image_ori"]["faces"].shape) == (1, 512, 512, 6)
    assert b["image_ori"]["ids"].dtype == torch.int64 and "sks" in b["caption"][0]
    # identity bookkeeping of face_id.py:556-583: specific_ids [1, 2] are WALK positions, the batch carries DENSE slots 0 / 1
    ds = data.datasets["train"]
    assert [ds.img_dict[k]["id"] for k in sorted(ds.img_dict)] == ["00001", "00002"]
    assert int(b["image_ori"]["ids"][0, 0]) in (0, 1) and int(b["image_ori"]["ids"][0, 0]) == int(b["image_ori"]["ids"][0, 1])
    # checkpoint format of embedding_manager.py:396-426: {"id_coefficients": [max_ids x (es, h, inner)]}
    em = m.embedding_manager
    p = tmp_path / "embeddings_gs-200.pt"
    em.save(str(p))
    ck = torch.load(str(p), weights_only=False)
    assert list(ck.keys()) == ["id_coefficients"] and len(ck["id_coefficients"]) == 10
    assert tuple(ck["id_coefficients"][0].shape) == (2, 1, 512) and ck["id_coefficients"][0].dtype == torch.float32
    em.id_coefficients = None
    em.load(str(p))
    assert torch.equal(em.id_coefficients[3], ck["id_coefficients"][3])
    # missing-weights error path: the native trainer cannot be built without loading weights
    with pytest.raises(RuntimeError):
        m._state_for_trainer()


def test_b2_method_contracts(tmp_path):
    """SURVEY.md §8 B2, item by item: the attributes, methods, argument names and return types the reference's callers rely on
    (main_id_embed.py, ddim.py, stable_txt2img.py, extract_pt.py) exist on the host mirror with the reference's meaning"""
    import inspect
    import json
    import celebbasis_amd
    celebbasis_amd.install_ldm_alias()
    from celebbasis_amd.ldm.util import instantiate_from_config
    cfg = json.load(open(REF_PARAMS))
    m = instantiate_from_config(cfg["model"])
    em, cs = m.embedding_manager, m.cond_stage_model
    # LatentDiffusion: `.learning_rate` set by the caller (main_id_embed.py:779)
    assert hasattr(m, "learning_rate")
    m.learning_rate = 2 * 5e-3
    # training_step(batch, idx) -> loss ; on_save_checkpoint(ckpt) ; get_learned_conditioning(c, face_img=None, image_ori=None)
    assert list(inspect.signature(m.training_step).parameters)[:2] == ["batch", "batch_idx"]
    assert "return tr.loss_tensor" in inspect.getsource(type(m).training_step)
    assert list(inspect.signature(m.on_save_checkpoint).parameters)[0] == "checkpoint"
    glc = inspect.signature(m.get_learned_conditioning).parameters
    assert list(glc) == ["c", "face_img", "image_ori"] and glc["face_img"].default is None and glc["image_ori"].default is None
    assert list(inspect.signature(m.apply_model).parameters)[:3] == ["x_noisy", "t", "cond"]
    assert list(inspect.signature(m.decode_first_stage).parameters)[0] == "z"
    assert list(inspect.signature(m.q_sample).parameters) == ["x_start", "t", "noise"]
    with m.ema_scope():
        pass
    # buffers DDIMSampler reads (ddim.py:16, 28-34, 120)
    for name in ("betas", "alphas_cumprod", "alphas_cumprod_prev"):
        t = getattr(m, name)
        assert isinstance(t, torch.Tensor) and t.shape == (1000,) and t.dtype == torch.float32
    assert m.num_timesteps == 1000 and abs(float(m.alphas_cumprod[0]) - 0.99915) < 1e-6
    assert isinstance(type(m).device, property)
    # configure_optimizers() -> torch.optim.AdamW over embedding_parameters() + trainable_parameters(), lr = learning_rate, torch defaults
    opt = m.configure_optimizers()
    assert isinstance(opt, torch.optim.AdamW)
    g = opt.param_groups[0]
    assert g["lr"] == m.learning_rate and g["betas"] == (0.9, 0.999) and g["eps"] == 1e-8 and g["weight_decay"] == 1e-2
    assert sum(p.numel() for p in g["params"]) == 525312
    # cond_stage_model.encode(text, embedding_manager=, face_img=, image_ori=) (ddpm.py:602-603)
    enc = inspect.signature(cs.forward).parameters
    assert list(enc)[:4] == ["text", "embedding_manager", "face_img", "image_ori"] and cs.encode.__func__ is not None
    # embedding manager call signature + small API
    fw = inspect.signature(em.forward).parameters
    assert list(fw) == ["tokenized_text", "embedded_text", "face_image", "img_ori", "celeb_embeddings"]
    assert em.embedding_parameters() == []
    tp = em.trainable_parameters()
    assert all(isinstance(p, torch.nn.Parameter) for p in tp) and [tuple(p.shape) for p in tp] == [(1024, 512), (1024,)]
    assert float(em.embedding_neg_loss()) == 0.0
    assert isinstance(em.id_coefficients, list) and len(em.id_coefficients) == 10 and tuple(em.id_coefficients[0].shape) == (2, 1, 512)
    assert isinstance(em.id_embeddings, list) and len(em.id_embeddings) == 10 and tuple(em.id_embeddings[0].shape) == (2, 768)
    assert em.id_coefficients[0] is em.id_coefficients[9]           # one randn block shared by all slots (embedding_manager.py:229-231)
    # id_embeddings start at the initializer word once the token table is known (embedding_manager.py:239-253)
    table = torch.randn(49408, 768)
    em.init_id_embeddings(table)
    assert torch.equal(em.id_embeddings[4], table[1710].unsqueeze(0).repeat(2, 1))          # 'face' = 1710
    em.save(str(tmp_path / "e.pt")); em.load(str(tmp_path / "e.pt"))
    # the same Parameter objects survive a trainer sync (trainable_parameters() stays valid for the caller's optimiser)
    w0 = em.meta_id_net.weight
    em.meta_id_net.load_trainable_state_dict({"stylegan_mlp.net.0.weight": torch.zeros(1024, 512), "stylegan_mlp.net.0.bias": torch.ones(1024)})
    assert em.meta_id_net.weight is w0 and float(w0.abs().sum()) == 0.0
    # state-dict key prefixes of SD checkpoints (main_id_embed.py:36, strict=False)
    sd = {"model.diffusion_model.time_embed.0.bias": torch.zeros(1280), "model_ema.decay": torch.zeros(()),
          "first_stage_model.quant_conv.bias": torch.zeros(8),
          "cond_stage_model.transformer.text_model.final_layer_norm.bias": torch.zeros(768)}
    missing, _ = m.load_state_dict(sd, strict=False)
    assert "model.diffusion_model.time_embed.0.bias" not in missing and "first_stage_model.quant_conv.bias" not in missing
    assert "cond_stage_model.transformer.text_model.final_layer_norm.bias" not in missing and len(missing) > 100
    # use_rm_mlp (meta_net.py:173-175): the coefficients themselves become the trainable parameter
    cfg["model"]["params"]["personalization_config"]["params"]["use_rm_mlp"] = True
    em2 = instantiate_from_config(cfg["model"]["params"]["personalization_config"], embedder=cs)
    assert tuple(em2.trainable_parameters()[-1].shape) == (10, 2, 1, 512)


def test_safetensors_checkpoint_loader(tmp_path):
    """main_id_embed.py:25-44: `.safetensors` and `.ckpt` (with or without the 'state_dict' wrapper) give the same state dict"""
    from safetensors.torch import save_file
    from celebbasis_amd.main import load_sd_checkpoint
    sd = {"model.diffusion_model.out.2.bias": torch.arange(4.0), "first_stage_model.quant_conv.weight": torch.randn(8, 8, 1, 1)}
    save_file(sd, str(tmp_path / "m.safetensors"))
    torch.save({"state_dict": sd}, tmp_path / "m.ckpt")
    torch.save(sd, tmp_path / "bare.ckpt")
    for name in ("m.safetensors", "m.ckpt", "bare.ckpt"):
        got = load_sd_checkpoint(str(tmp_path / name))
        assert set(got) == set(sd) and all(torch.equal(got[k], sd[k]) for k in sd), name


def test_identity_sharding_and_gloo_merge(tmp_path):
    from celebbasis_amd import sharding
    assert sharding.identities_for_rank(8, 3, 8) == [3]
    assert sorted(sum((sharding.identities_for_rank(10, r, 4) for r in range(4)), [])) == list(range(10))
    script = tmp_path / "w.py"
    script.write_text(f"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, {ROOT!r})
from celebbasis_amd import sharding
dist.init_process_group("gloo")
r, w = dist.get_rank(), dist.get_world_size()
mine = sharding.identities_for_rank(5, r, w)
blocks = torch.stack([torch.full((1024,), float(i + 1)) for i in mine])
out = sharding.gather_identity_blocks(mine, blocks, 5)
assert out.shape == (5, 1024)
for i in range(5):
    assert torch.all(out[i] == i + 1), (r, i, out[i][:3])
dist.barrier()
if r == 0:
    open({str(tmp_path / 'ok')!r}, "w").write("ok")
""")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                    "--master-port", "29533", str(script)], check=True, env=env, timeout=300)
    assert (tmp_path / "ok").read_text() == "ok"
    # fewer identities than ranks: the rank that owns none contributes an EMPTY block list (what main.py hands over: zeros(0, D))
    # and still takes part in both collectives; everybody ends up with the same merged table
    script2 = tmp_path / "w3.py"
    script2.write_text(f"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, {ROOT!r})
from celebbasis_amd import sharding
dist.init_process_group("gloo")
r, w = dist.get_rank(), dist.get_world_size()
mine = sharding.identities_for_rank(2, r, w)
assert (mine == []) == (r == 2)
blocks = torch.stack([torch.full((12,), float(10 * (i + 1))) for i in mine]) if mine else torch.zeros(0, 12)
out = sharding.gather_identity_blocks(mine, blocks, 4)
assert out.shape == (4, 12) and torch.all(out[0] == 10) and torch.all(out[1] == 20) and torch.all(out[2:] == 0), (r, out[:, 0])
dist.barrier()
if r == 2:
    open({str(tmp_path / 'ok3')!r}, "w").write("ok")
""")
    subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr", "127.0.0.1",
                    "--master-port", "29534", str(script2)], check=True, env=env, timeout=300)
    assert (tmp_path / "ok3").read_text() == "ok"


def test_bench_launches_its_own_ranks_and_checks_the_merge():
    """`python bench.py --gpus 2` with no launcher environment re-executes itself under torch.distributed.run (one rank per GPU);
    here on dry contexts with gloo: world size, identity sharding, the checkpoint-merge gather and the JSON contract of the N > 1
    line. A launcher / --gpus mismatch is an error instead of a silently wrong `n_gpus`."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=str(ROOT))
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["merged_identities"] == 2 and d["scaling"] == "weak" and d["dry"] is True
    assert d["config"]["global_batch"] == 2 and abs(d["identities_per_sec"] - d["value"] / 800) < 1e-5
    assert d["cpu_baseline"] is None and d["steps"] == 2 and d["warmup"] == 1
    env["WORLD_SIZE"] = "1"; env["RANK"] = "0"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=str(ROOT))
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)


def test_run_config5_launches_its_own_ranks_and_checks_the_merge():
    """`python tools/run_config5.py --gpus 2 --dry`: the config-5 driver re-executes itself under torch.distributed.run like bench.py;
    on dry contexts with gloo it walks its identity shards (5 identities on 2 ranks: 3 + 2), merges the per-identity blocks with the
    one all_gather of the job and checks every identity's block on rank 0. A launcher / --gpus mismatch is an error."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    script = os.path.join(ROOT, "tools", "run_config5.py")
    r = subprocess.run([sys.executable, script, "--gpus", "2", "--identities", "5", "--train-steps", "2", "--dry"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=str(ROOT))
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["merged_identities"] == 5 and d["identities_per_rank"] == [3, 2] and d["dry"] is True
    env["WORLD_SIZE"] = "1"; env["RANK"] = "0"
    r = subprocess.run([sys.executable, script, "--gpus", "2", "--identities", "2", "--dry"], capture_output=True, text=True, timeout=300,
                       env=env, cwd=str(ROOT))
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)


def test_eight_rank_launch_of_bench_and_config5_dry():
    """the two 8-GPU forms of BASELINE.json (config 4: 8 identities, one per GPU; config 5: 32 identities, 4 per GPU back to back) as the
    driver would launch them, on dry contexts with gloo: `python bench.py --gpus 8 --dry` and `python tools/run_config5.py --gpus 8
    --identities 32 --dry` start their own eight ranks, shard `i -> rank i mod 8`, run the one all_gather of the job and check the merge on
    rank 0; the bench line carries every rank's own clock (`per_rank`). No 8-GPU node is available to the builder: this is the launcher,
    sharding, merge and JSON contract at the real world size."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dry", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=str(ROOT))
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 8 and d["merged_identities"] == 8 and d["scaling"] == "weak" and d["config"]["global_batch"] == 8
    assert [p["rank"] for p in d["per_rank"]] == list(range(8)) and all(p["elapsed_s"] > 0 for p in d["per_rank"])
    assert abs(d["value"] - 8 * d["steps"] / max(p["elapsed_s"] for p in d["per_rank"])) <= 0.02 * d["value"]      # value = whole job / MAX over ranks
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "run_config5.py"), "--gpus", "8", "--identities", "32", "--train-steps", "1", "--dry"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=str(ROOT))
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 8 and d["merged_identities"] == 32 and d["identities_per_rank"] == [4] * 8 and d["dry"] is True


def test_build_guard_flags_scratch_in_counted_wait_kernels():
    """celebbasis_amd/build.py parses -Rpass-analysis=kernel-resource-usage remarks and fails the build when a kernel whose software
    pipeline counts its VMEM instructions (cb_gemm_glds.hip, cb_gemm_bf16_p8.hip, cb_flash_bf16_v2.hip) has scratch, spills or a dynamic stack"""
    from celebbasis_amd import build
    ok = """x.hip:5:1: remark: Function Name: _ZN2cb16gemm_glds_kernelIfLi64ELi64ELi0ELi128ELi2EEEvNS_5KArgsE [-Rpass-analysis=kernel-resource-usage]
x.hip:5:1: remark:     VGPRs: 47 [-Rpass-analysis=kernel-resource-usage]
x.hip:5:1: remark:     ScratchSize [bytes/lane]: 0 [-Rpass-analysis=kernel-resource-usage]
x.hip:5:1: remark:     Dynamic Stack: False [-Rpass-analysis=kernel-resource-usage]
x.hip:5:1: remark:     SGPRs Spill: 0 [-Rpass-analysis=kernel-resource-usage]
x.hip:5:1: remark:     VGPRs Spill: 0 [-Rpass-analysis=kernel-resource-usage]
x.hip:9:1: remark: Function Name: _ZN2cb20splitk_reduce_kernelIfEEvNS_5KArgsE [-Rpass-analysis=kernel-resource-usage]
x.hip:9:1: remark:     ScratchSize [bytes/lane]: 64 [-Rpass-analysis=kernel-resource-usage]
"""
    bad, seen = build.check_no_scratch(ok, "gemm_glds")
    assert bad == [] and seen == 1                       # the spilling kernel is not one of the guarded ones
    spill = ok.replace("ScratchSize [bytes/lane]: 0", "ScratchSize [bytes/lane]: 104").replace("VGPRs Spill: 0", "VGPRs Spill: 25")
    bad, seen = build.check_no_scratch(spill, "gemm_glds")
    assert seen == 1 and {b[1] for b in bad} == {"ScratchSize [bytes/lane]", "VGPRs Spill"} and bad[0][0].startswith("_ZN2cb16gemm_glds")
    assert build.check_no_scratch("", "gemm_glds") == ([], 0)          # no remarks at all is an error for the caller (seen == 0)
    d = build.source_digest()
    assert set(d) == {"src_sha", "plan_table_sha"} and len(d["src_sha"]) == 16 and d == build.source_digest()
    assert set(build.NO_SCRATCH) == {"cb_gemm_glds.hip", "cb_gemm_bf16_p8.hip", "cb_flash_bf16_v2.hip"}


def test_four_ranks_against_a_deleted_build_dir_compile_once(tmp_path):
    """N ranks importing the package against a missing library all call build(): the flock in celebbasis_amd/build.py lets one of them
    compile while the others wait and then find everything up to date. Four processes start together against an empty build directory
    with a stub compiler that logs every invocation: each source is compiled exactly once, the library is linked exactly once."""
    stub = tmp_path / "hipcc_stub.py"
    stub.write_text("""#!/usr/bin/env python3
import os, sys, time
a = sys.argv[1:]
out = a[a.index('-o') + 1]
with open(os.environ['STUB_LOG'], 'a') as f:
    f.write(('link ' if '-shared' in a else 'cc ') + os.path.basename(out) + '\\n')
time.sleep(0.15)
if '-Rpass-analysis=kernel-resource-usage' in a:
    sys.stderr.write('x.hip:1:1: remark: Function Name: gemm_glds_stub [-Rpass]\\nx.hip:1:1: remark:     ScratchSize [bytes/lane]: 0 [-Rpass]\\n'
                     'x.hip:2:1: remark: Function Name: gemm_bf16_p8_stub [-Rpass]\\nx.hip:2:1: remark:     ScratchSize [bytes/lane]: 0 [-Rpass]\\n')
open(out, 'w').write('stub')
""")
    stub.chmod(0o755)
    rank = tmp_path / "rank.py"
    rank.write_text(f"""import os, sys
sys.path.insert(0, {ROOT!r})
from celebbasis_amd import build
build.BUILD = {str(tmp_path / '_build')!r}; build.LIB = {str(tmp_path / 'lib.so')!r}
print(build.build(verbose=False))
""")
    env = dict(os.environ, HIPCC=str(stub), STUB_LOG=str(tmp_path / "log.txt"))
    procs = [subprocess.Popen([sys.executable, str(rank)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for _ in range(4)]
    outs = [p.communicate(timeout=120) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-400:] for o in outs]
    assert all(o[0].strip().endswith("lib.so") for o in outs)
    from celebbasis_amd import build
    log = (tmp_path / "log.txt").read_text().split("\n")
    assert sum(l.startswith("link ") for l in log) == 1, log
    assert len([l for l in log if l.startswith("cc ") and "probe" not in l]) == len(build.SOURCES), log


def test_bench_uses_committed_profiles_only_when_their_stamp_is_this_tree(tmp_path, monkeypatch):
    """bench.py quotes rocprofv3 / PMC numbers from profiles/ only if they were measured on the running tree (stamp = sha of the kernel
    sources + plan table); anything else is `profile_stale` and contributes nothing to the line"""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    spec.loader.exec_module(bench)
    st = bench.tree_stamp()
    (tmp_path / "profiles").mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    body = "name,calls,total_ms,ms_per_step,avg_us,min_us,max_us,pct\n" \
           '"gemm_glds_kernel<float, 128, 128, 2, 128, 2, true>",378,79.8,8.867,211.1,72.2,736.1,20.8\n"splitk_reduce_kernel<float>",3924,22.5,2.509,5.75,2.8,14.6,5.9\n'
    good = f"# stamp src_sha={st['src_sha']} plan_table_sha={st['plan_table_sha']} nsteps=9\n# rocprofv3\n" + body
    (tmp_path / "profiles" / "r06z_bench_kernel_stats.csv").write_text(good)
    src, stale, per, base = bench.rocprof_trace(False)
    assert src == "r06z_bench_kernel_stats.csv" and stale is False and abs(base["splitk_reduce_kernel"] - 2.509) < 1e-9
    assert per["gemm_glds_kernel<float, 128, 128, 2, 128, 2, true>"][0] == 42.0           # launches per step from calls / nsteps
    # one profiler label = the scalar-offset (FAST) and the per-lane form of the kernel; the 80-column tiles: <BM, A kind, stages, FAST>
    assert bench.trace_symbols("gemm_glds_kernel<float,128,128,CONVF>") == ["gemm_glds_kernel<float, 128, 128, 2, 128, 2, true>",
                                                                            "gemm_glds_kernel<float, 128, 128, 2, 128, 2, false>"]
    assert bench.trace_symbols("gemm_glds80_kernel<float,32,80,ROW>")[0] == "gemm_glds80_kernel<32, 0, 2, true>"
    assert bench.trace_symbols("gemm_bf16_p8_kernel<bf16,256,256,CONVF>") == ["gemm_bf16_p8_kernel<2, 2, 0>"]      # <A kind, waves along M, ablation>
    assert bench.trace_symbols("gemm_bf16_p8_kernel<bf16,512,128,ROW>") == ["gemm_bf16_p8_kernel<0, 4, 0>"]
    (tmp_path / "profiles" / "r06z_bench_kernel_stats.csv").write_text(good.replace(st["src_sha"], "0" * 16))
    src, stale, per, base = bench.rocprof_trace(False)
    assert stale is True and per == {} and base == {}
    pmc = {"stamp": st, "per_symbol": {"gemm_glds_kernel<float, 128, 128, 2, 128, 2, true>": {"launches": 210, "fetch_bytes_per_launch": 90e6, "write_bytes_per_launch": 24e6}}}
    (tmp_path / "profiles" / "r06z_pmc_hbm_traffic.json").write_text(json.dumps(pmc))
    t, det = bench.pmc_traffic("gemm_glds_kernel<float,128,128,CONVF>")
    assert t == 204000000 and det["stale"] is False and det["fetch_size_bytes_per_launch_raw"] == 90000000      # 2 x FETCH_SIZE + WRITE_SIZE
    pmc["stamp"] = {"src_sha": "x", "plan_table_sha": "y"}
    (tmp_path / "profiles" / "r06z_pmc_hbm_traffic.json").write_text(json.dumps(pmc))
    t, det = bench.pmc_traffic("gemm_glds_kernel<float,128,128,CONVF>")
    assert t is None and det["stale"] is True


def test_bench_line_contract_of_committed_profile():
    """the bench.py JSON line committed under profiles/ carries every field of the driver's contract (keys and types), the
    roofline and cpu_baseline objects, and internally consistent numbers"""
    import glob
    import json
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "BENCH_r0*_run*.json")),
                   key=lambda p: (int(p.split("BENCH_r")[-1].split("_")[0]), int(p.split("run")[-1].split(".")[0])))
    full = [json.load(open(p)) for p in files]
    full = [d for d in full if d.get("cpu_baseline")]
    assert full, "no committed bench line with a cpu_baseline"
    d = full[-1]
    for k, t in dict(metric=str, value=float, unit=str, n_gpus=int, steps=int, warmup=int, ms_per_step=float, higher_is_better=bool,
                     scaling=str, dtype=str, data=str, config=dict, roofline=dict, cpu_baseline=dict).items():
        assert isinstance(d[k], t), (k, type(d[k]))
    assert d["vs_baseline"] is None and d["scaling"] == "weak" and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - d["n_gpus"] * 1e3 / d["ms_per_step"]) < 0.02 * d["value"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert "traffic" in r and r["peak"] == 157.3
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and isinstance(c["sample"], str)
    assert d["value"] / c["value"] > 50            # the GPU path is two orders of magnitude above the host baseline
    if "config3" in d:                             # round 2 lines: the config-3 (bf16, bs=4) secondary and the scaling fields
        c3 = d["config3"]
        assert c3["dtype"] == "bf16" and c3["batch"] == 4 and c3["roofline"]["peak"] == 2500.0
        assert abs(c3["samples_per_s"] - 4e3 / c3["ms_per_step"]) < 0.02 * c3["samples_per_s"]
        assert d["merged_identities"] == d["n_gpus"] and d["identities_per_sec"] > 0


def test_feeder_draws_and_lookahead():
    """host side of the feeder: parameter draws stay in the reference's ranges (face_id.py:526-532, 719-727) and the
    one-batch lookahead used by --pipeline yields every batch exactly once, in order"""
    import random
    from celebbasis_amd.ldm.data.face_id import FaceIdDatasetOneShot
    ds = FaceIdDatasetOneShot(pickle_path=None, num_ids=1, size=64, seed=1, diff_cnt=0, synthetic=True)
    ds.rng = random.Random(0)
    for _ in range(200):
        (flip, b, c, s, h), order = ds._jitter_params()
        assert flip in (0.0, 1.0) and all(0.8 <= v <= 1.2 for v in (b, c, s)) and -0.01 <= h <= 0.01
        assert sorted(order) == [0, 1, 2, 3]                     # ColorJitter's fn_idx: a permutation of the four ops
        rh, rw, ph, pw = ds._bg_geometry()
        assert 1 <= rh <= 64 and 1 <= rw <= 64 and 0 <= ph <= 64 - rh and 0 <= pw <= 64 - rw
        assert rh >= int(64 * 0.1) and abs(rw - rh) <= int(0.1 * rh) + 1
    # splits (face_id.py:516-539, 627): only 'dev' is special — it keeps the flip but has no ColorJitter and no `_add_bg`; 'val' (the
    # shipped validation block, aigc_id.yaml:110-119) is augmented like 'train'; every split repeats its list
    mk = lambda split: FaceIdDatasetOneShot(pickle_path=None, num_ids=2, size=32, seed=1, diff_cnt=0, synthetic=True, split=split, repeats=3)
    tr_, val, dev = mk("train"), mk("val"), mk("dev")
    assert len(tr_) == len(val) == len(dev) == 6
    a, b = tr_[1], val[1]
    assert torch.equal(a["image"], b["image"]) and torch.equal(a["image_ori"]["faces"], b["image_ori"]["faces"]) and a["caption"] == b["caption"]
    flips = 0
    for i in range(20):
        d = dev[i]
        face = dev.faces[i % 2]
        img1, img3 = d["image_ori"]["faces"][..., :3], d["image_ori"]["faces"][..., 3:]
        for im in (img1, img3):
            same, flipped = float((im - face).abs().max()), float((im - face.flip(1)).abs().max())
            assert min(same, flipped) < 1e-6                    # flip or not, nothing else
            flips += flipped < same
        assert torch.equal(d["image"], img1)                    # no rescale-paste on 'dev'
    assert 5 <= flips <= 35                                     # RandomHorizontalFlip(p=0.5) stays on
    src = open(os.path.join(ROOT, "celebbasis_amd", "main.py")).read()
    ns = {}
    start = src.index("    def lookahead(loader):"); end = src.index("    while step < max_steps:")
    exec("def make():\n" + src[start:end] + "    return lookahead\n", ns)
    pairs = list(ns["make"]()([10, 11, 12]))
    assert pairs == [(10, 11), (11, 12), (12, None)] and list(ns["make"]()([])) == []


def test_oracle_jitter_vs_pil():
    """A0 pin. The reference's `trans` (face_id.py:526-532) receives PIL images, so torchvision 0.11.3's ColorJitter runs PIL's
    arithmetic: ImageEnhance.Brightness / Contrast / Color (a blend of the image against black / its mean gray / its gray
    version) and, for hue, a shift of the H plane of the 8-bit HSV image — each quantised to uint8. torchvision itself is not
    installed; PIL is, so the oracle's float restatement (and with it the device kernel, which the GPU suite checks against the
    oracle) is pinned against PIL's results to that quantisation: <= 1.5 LSB per brightness / contrast / saturation op
    (PIL truncates), hue within the 8-bit HSV round trip (mean <= 1.5 LSB, max <= 16 LSB — a wrong sign is tens of LSB),
    and a whole permuted chain of ops (the matching order is the closest one). Also: the feeder's four single-op passes compose to the
    oracle's permuted chain (DeviceAugmenter.passes, host logic only)."""
    np = pytest.importorskip("numpy")
    Image = pytest.importorskip("PIL.Image")
    ImageEnhance = pytest.importorskip("PIL.ImageEnhance")
    from celebbasis_amd.ldm.data.face_id import DeviceAugmenter
    from oracle import cb_oracle as O
    g = torch.Generator().manual_seed(0)
    S = 96
    low = torch.rand(1, 3, 8, 8, generator=g)
    u8 = (torch.nn.functional.interpolate(low, size=S, mode="bicubic", align_corners=True)[0].permute(1, 2, 0).clamp(0, 1) * 255).round().to(torch.uint8)
    pil = Image.fromarray(u8.numpy(), "RGB")
    x = u8.float() / 255 * 2 - 1                               # ToTensor + Normalize(0.5, 0.5) of the same image, HWC
    lsb = lambda a, b: (a - b).abs() * 127.5
    to_t = lambda im: torch.from_numpy(np.asarray(im).copy()).float() / 255 * 2 - 1

    def pil_hue(im, hf):                                       # torchvision functional_pil.adjust_hue
        h, s_, v = im.convert("HSV").split()
        nh = np.array(h, dtype=np.uint8)
        with np.errstate(over="ignore"):
            nh = nh + np.uint8(int(hf * 255) % 256)
        return Image.merge("HSV", (Image.fromarray(nh, "L"), s_, v)).convert("RGB")

    ops = {0: lambda im, f: ImageEnhance.Brightness(im).enhance(f), 1: lambda im, f: ImageEnhance.Contrast(im).enhance(f),
           2: lambda im, f: ImageEnhance.Color(im).enhance(f), 3: pil_hue}
    for f in (0.8, 0.93, 1.0, 1.13, 1.2):
        for op in (0, 1, 2):
            args = [1.0, 1.0, 1.0, 0.0]; args[op] = f
            d = lsb(to_t(ops[op](pil, f)), O.color_jitter_flip(x, False, *args))
            assert float(d.max()) <= 1.5, (op, f, float(d.max()))
    for hf in (-0.01, 0.004, 0.01, 0.1, -0.25, 0.4):
        d = lsb(to_t(pil_hue(pil, hf)), O.color_jitter_flip(x, False, 1.0, 1.0, 1.0, hf))
        wrong = lsb(to_t(pil_hue(pil, hf)), O.color_jitter_flip(x, False, 1.0, 1.0, 1.0, -hf))
        assert float(d.mean()) <= 1.5 and float(d.max()) <= 16.0, (hf, float(d.mean()), float(d.max()))
        if abs(hf) >= 0.1:
            assert float(wrong.mean()) > 10 * float(d.mean())    # the check has teeth: the opposite shift is far outside the bar
    # a permuted chain, flip included (PIL: FLIP_LEFT_RIGHT, then the ops in fn_idx order, uint8 after each)
    params, order = [1.0, 1.17, 0.84, 1.12, 0.0], [2, 0, 3, 1]
    im = pil.transpose(Image.FLIP_LEFT_RIGHT)
    for op in order:
        if op != 3:
            im = ops[op](im, params[1 + op])
    d = lsb(to_t(im), O.color_jitter_flip(x, True, *params[1:], order=order))
    # (PIL's blend truncates, so each op adds up to -1 LSB of bias: three ops = mean 1.5, max 3)
    assert float(d.max()) <= 4.5 and float(d.mean()) <= 2.0, (float(d.max()), float(d.mean()))
    for wrong_order in ([0, 1, 2, 3], [1, 0, 2, 3]):             # contrast's pivot depends on what ran before it: the order is visible
        other = lsb(to_t(im), O.color_jitter_flip(x, True, *params[1:], order=wrong_order))
        assert float(other.mean()) > 1.5 * float(d.mean()), (wrong_order, float(other.mean()), float(d.mean()))
    # the device feeder's decomposition: four passes of the fixed-order op with three neutral factors each == the permuted chain
    params[4] = 0.008
    rows = DeviceAugmenter.passes(params, order)
    y = x
    for k in range(4):
        r = rows[5 * k:5 * k + 5]
        y = O.color_jitter_flip(y, bool(r[0]), *r[1:])
    assert float((y - O.color_jitter_flip(x, True, *params[1:], order=order)).abs().max()) <= 2e-6
    assert [r for r in rows[0::5]] == [1.0, 0.0, 0.0, 0.0]


def test_argument_errors_on_a_dry_context(built_lib):
    """edge cases are rejected with a status + message before anything is launched (checked on a dry context, no GPU):
    batch above the supported maximum, prompt longer than the row-surgery kernel handles, zero scale factor, eta != 0
    without noise, bad GEMM shape"""
    import ctypes as C
    from celebbasis_amd import lib, synthetic
    from oracle import cb_oracle as O
    ctx = lib.Context(device=-1)
    L = lib.load()
    shapes = dict(synthetic.trainer_shapes()); shapes.update(synthetic.decoder_shapes())
    wt = lib.WeightTable(shapes)
    tr = lib.Trainer(ctx, wt)
    tr.step(16, None, None, None, None, None, None, 5e-3)          # the largest supported batch plans on a dry context
    for bad in (0, 17):
        with pytest.raises(lib.CBError) as e:
            tr.step(bad, None, None, None, None, None, None, 5e-3)
        assert "batch" in str(e.value)
    rc = L.cb_row_surgery(ctx.h, None, None, None, None, None, 1, 300, 768, 2, C.c_int64(48136))
    assert rc != 0 and b"T=300" in L.cb_last_error()
    dec = lib.VAEDecoder(ctx, wt, lib.vae_cfg(), prefix="first_stage_model.")
    with pytest.raises(lib.CBError) as e:
        dec.decode(torch.zeros(1, 4, 8, 8), scale_factor=0.0)
    assert "scale_factor" in str(e.value)
    u = lib.UNet(ctx, wt, lib.unet_cfg(), prefix="model.diffusion_model.")
    s = O.ddim_schedule(O.make_schedule()["alphas_cumprod"], 4, 0.5)
    with pytest.raises(lib.CBError) as e:
        u.ddim_sample(torch.zeros(1, 4, 8, 8), torch.zeros(1, 77, 768), None, s["timesteps"], s["alphas"], s["alphas_prev"], s["sigmas"],
                      s["sqrt_one_minus_alphas"], noise=None)
    assert "noise" in str(e.value)
    d = lib.GemmDesc(); d.M, d.N, d.K, d.batch = 0, 8, 8, 1
    assert L.cb_gemm(ctx.h, C.byref(d)) != 0 and b"shape" in L.cb_last_error()


def test_plan_table_is_well_formed(tmp_path):
    """celebbasis_amd/csrc/cb_plan_table.inc (generated by tools/make_plan_table.py from offline autotune records): every entry parses
    as "<key> <BM> <BN> <split>", keys are unique, tiles are ones the kernels implement (the 8-wave tiles — 256x128 and the
    phase-pipelined 256x256 / 512x128 — only for the bf16 64-element k-tile, the latter two only for K % 64 == 0 ROW / plain-conv
    problems; the 80-column tiles only for fp32 ROW / plain-conv problems without a batch), and the generator reproduces the committed
    table from the committed records (round 4's fp32 records, round 6's bf16 re-measurement, the records carried over from round 2)"""
    import subprocess
    import sys
    inc = os.path.join(ROOT, "celebbasis_amd", "csrc", "cb_plan_table.inc")
    lines = [l.strip() for l in open(inc) if l.strip().startswith('"') and len(l.strip()) > 2]
    assert len(lines) > 100
    keys = set()
    for l in lines:
        body = l.strip('"').replace("\\n", "")
        key, bm, bn, sp = body.split()[:4]
        f = key.split("|")
        assert len(f) == 8 and key not in keys, key
        keys.add(key)
        bm, bn, sp = int(bm), int(bn), int(sp)
        assert (bm, bn) in ((128, 128), (128, 64), (64, 128), (64, 64), (256, 128), (64, 80), (32, 80), (16, 80), (256, 256), (512, 128)) and sp >= 1
        if bn == 80:                                      # fp32 (dtype 0), k-tile 32, ROW (0) or conv (2) without upsample / transposed flags, batch 1
            assert f[0] == "32" and f[1] == "0" and f[2] in ("0", "2") and f[6] == "1" and (int(f[7]) & 3) == 0
        if bm >= 256:
            assert f[0] == "64" and f[1] == "1"          # the 8-wave tiles exist for bf16 (dtype code 1), 64-element k-tiles only
        if (bm, bn) in ((256, 256), (512, 128)):         # cb_gemm_bf16_p8.hip: whole 64-element k-tiles, ROW or plain conv gathers
            assert int(f[5]) % 64 == 0 and f[2] in ("0", "2") and (int(f[7]) & 3) == 0
        assert -(-int(f[5]) // int(f[0])) >= sp           # never more splits than k-tiles
    before = open(inc).read()
    try:
        # fp32 plans: round 4's records (unchanged summation orders: the 800-step trajectories are the ones validated in round 5);
        # bf16 plans: round 6's re-measurement with the phase-pipelined tiles among the candidates
        recs = [os.path.join(ROOT, "profiles", n) for n in ("r04_autotune.txt:f32", "r04_autotune_carried_from_r02.txt", "r06_autotune.txt:bf16")]
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_plan_table.py")] + recs, capture_output=True, text=True, check=True).stdout
        assert "shapes measured" in out and open(inc).read() == before, "cb_plan_table.inc is not what the generator makes of the committed records"
    finally:
        open(inc, "w").write(before)


def test_upsample_conv_gradient_identity():
    """the algebra behind CB_CONV_UP_DGRAD4 / pack_conv_up_dgrad_kernel, checked against torch autograd on the CPU (float64): the input
    gradient of `conv3x3(nearest_upsample_x2(x))` (openaimodel.py Upsample) equals ONE 4x4 stride-2 pad-1 convolution over dy whose kernel
    rows / columns are the 3x3 kernel's combined as {w2, w1+w2, w0+w1, w0}"""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(0)
    N, C, Co, H, W = 2, 5, 7, 6, 9
    x = torch.randn(N, C, H, W, dtype=torch.float64, generator=g).requires_grad_(True)
    w = torch.randn(Co, C, 3, 3, dtype=torch.float64, generator=g)
    y = F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), w, padding=1)
    dy = torch.randn(y.shape, dtype=torch.float64, generator=g)
    (dx_ref,) = torch.autograd.grad((y * dy).sum(), x)

    def comb(t, dim):
        w0, w1, w2 = t.select(dim, 0), t.select(dim, 1), t.select(dim, 2)
        return torch.stack([w2, w1 + w2, w0 + w1, w0], dim=dim)
    w4 = comb(comb(w, 2), 3)                                          # [Co, C, 4, 4]
    dx = F.conv2d(dy, w4.transpose(0, 1), stride=2, padding=1)
    assert dx.shape == dx_ref.shape and (dx - dx_ref).abs().max().item() < 1e-12


def test_warp_grid_and_graph_replay_entries_on_a_dry_context(built_lib):
    """round-5 ABI entries without a GPU: cb_ctx_set_warp_grid validates its table (size, finite normalised coordinates), accepts the
    reference's own affine_grid through lib.Context.ensure_warp_grid / set_warp_grid and drops it on NULL; cb_trainer_set_graph_replay /
    cb_trainer_graph_stats are callable on a dry trainer, whose steps never capture anything (a dry context computes nothing)."""
    import ctypes as C
    import torch
    from celebbasis_amd import lib, synthetic
    L = built_lib
    ctx = lib.Context(device=-1)
    g = torch.zeros(8, 8, 2)
    assert L.cb_ctx_set_warp_grid(ctx.h, C.c_void_p(g.data_ptr()), 1) != 0 and b"S = 1" in L.cb_last_error()
    bad = g.clone(); bad[0, 0, 0] = float("nan")
    assert L.cb_ctx_set_warp_grid(ctx.h, C.c_void_p(bad.data_ptr()), 8) != 0 and b"normalised coordinate" in L.cb_last_error()
    far = g.clone(); far[0, 0, 0] = 100.0
    assert L.cb_ctx_set_warp_grid(ctx.h, C.c_void_p(far.data_ptr()), 8) != 0
    ctx.ensure_warp_grid(64)
    want = torch.nn.functional.affine_grid(torch.tensor([lib.WARP_TRANS_MATRIX], dtype=torch.float32), size=[1, 3, 64, 64], align_corners=True)[0]
    assert ctx._warp_S == 64 and torch.equal(ctx.warp_grid_host, want)
    ctx.ensure_warp_grid(64)                                     # cached: same object
    assert ctx.warp_grid_host.data_ptr() == ctx.warp_grid_host.data_ptr()
    ctx.set_warp_grid(want + 0.0)
    assert L.cb_ctx_set_warp_grid(ctx.h, None, 0) == 0           # NULL drops the table
    with pytest.raises(AssertionError):
        ctx.set_warp_grid(torch.zeros(8, 4, 2))                  # not square
    tr = lib.Trainer(ctx, lib.WeightTable(synthetic.trainer_shapes()))
    assert ctx._warp_S == 512                                    # the trainer installs the table of its image size
    tr.set_graph_replay(True)
    for _ in range(3):
        tr.step(1, None, None, None, None, None, None, 5e-3)
    assert tr.graph_stats() == dict(captures=0, replays=0)
    tr.set_graph_replay(False)
