"""Feeder counterpart of ldm/data/face_id.py `FaceIdDatasetStyleGAN3` / `FaceIdDatasetOneShot` (reference :471-757).
It produces the same sample dict (face_id.py:598-644): image (512,512,3) f32 in [-1,1], image_ori {faces (512,512,6), ids (2,),
num_ids}, caption, id_idx, img_idx — with the same 81-template caption list and `diff_cnt=0 => id_cnt=2`.
Augmentations (flip, colour jitter, `_add_bg` rescale-paste): the random draws are made on the host like the reference's
transforms; the pixel work runs either per sample on the CPU (`__getitem__`, DataLoader workers, as in the reference) or for a
whole batch on the device (`DeviceAugmenter`: cb_face_augment + cb_paste_on_canvas), which is what keeps the feeder ahead of a
21 steps/s training loop. ColorJitter's four ops run in a per-call random order, as torchvision's do. The formulas are PIL's
(the backend the reference's PIL inputs select) in float, i.e. without PIL's uint8 rounding after every op; bit-exact parity with
that rounding is not a goal: parity of the training step is defined on a replay stream. Without a dataset on disk the feeder synthesises seeded faces so the rest of the pipeline can run offline."""
import pickle
import random

import torch
import torch.nn.functional as F

_OBJ = ["a photo of", "a rendering of", "a cropped photo of the", "the photo of", "a photo of a clean", "a photo of a dirty",
        "a dark photo of the", "a photo of my", "a photo of the cool", "a close-up photo of", "a bright photo of the",
        "a cropped photo of", "a photo of the", "a good photo of the", "a photo of one", "a close-up photo of the",
        "a rendition of the", "a photo of the clean", "a rendition of", "a photo of a nice", "a good photo of",
        "a photo of the nice", "a photo of the small", "a photo of the weird", "a photo of the large", "a photo of a cool",
        "a photo of a small"]


def _templates():
    """the 81 entries of imagenet_templates_small (face_id.py:17-99): 27 'photo' prompts then the same 27 positions with
    'illustration' / 'depiction' substituted where the original says 'a photo of…' plus the shared render/crop/dark/close-up lines"""
    base = ['a photo of a {}', 'a rendering of a {}', 'a cropped photo of the {}', 'the photo of a {}', 'a photo of a clean {}',
            'a photo of a dirty {}', 'a dark photo of the {}', 'a photo of my {}', 'a photo of the cool {}', 'a close-up photo of a {}',
            'a bright photo of the {}', 'a cropped photo of a {}', 'a photo of the {}', 'a good photo of the {}', 'a photo of one {}',
            'a close-up photo of the {}', 'a rendition of the {}', 'a photo of the clean {}', 'a rendition of a {}',
            'a photo of a nice {}', 'a good photo of a {}', 'a photo of the nice {}', 'a photo of the small {}',
            'a photo of the weird {}', 'a photo of the large {}', 'a photo of a cool {}', 'a photo of a small {}']
    out = list(base)
    for word in ("an illustration", "a depiction"):
        for t in base:
            keep = ("rendering" in t or "cropped" in t or t.startswith("the photo") or "dark photo" in t or "close-up" in t
                    or "bright photo" in t or "good photo" in t or "rendition" in t)
            out.append(t if keep else t.replace("a photo", word, 1))
    return out


imagenet_templates_small = _templates()


class FaceIdDatasetStyleGAN3(torch.utils.data.Dataset):
    """Same constructor keywords and identity bookkeeping as the reference (face_id.py:471-597): the pickle is a list of image
    paths; paths are grouped by basename stem (one group = one identity, in order of first appearance); `specific_ids` are
    0-based WALK positions over those groups; the selected identities get DENSE slots 0..num_ids-1 (`use_id`), and it is the
    dense slot that goes into `image_ori.ids` / `id_idx` — hence into `id_coefficients[slot]` of the saved checkpoint, where
    02_start_test.sh reads `--eval_id1 0`. `images_per_id` is 1 as in the reference (:485).
    Build-only keywords: `only_slots` restricts the samples served to a subset of the dense slots (identity sharding across
    ranks keeps the reference's slot numbering); `synthetic=True` replaces the pickle by seeded smooth random faces (offline
    runs, benchmarks) — without it a missing / unreadable pickle is an error, never a silent stand-in."""

    def __init__(self, pickle_path=None, num_ids=10, specific_ids=None, image_size=512, repeats=100, flip_p=0.5, split="train",
                 diff_cnt=32, size=None, placeholder="sks", seed=23, synthetic=False, only_slots=None, **ignored):
        if diff_cnt != 0:
            raise NotImplementedError("diff_cnt > 0 feeds the contrastive losses, which are dead under loss_type 'none' "
                                      "(aigc_id.yaml:31,104); the shipped configs set diff_cnt: 0")
        if isinstance(specific_ids, str):                       # 'lo-hi' (face_id.py:493-499)
            import re
            if re.match(r"\d+-\d+$", specific_ids) is None:
                raise ValueError("Specific_ids not supported.")
            lo, hi = [int(x) for x in specific_ids.split("-")]
            specific_ids = list(range(lo, hi))
        self.size = int(size if size is not None else image_size)
        self.flip_p, self.split, self.repeats = flip_p, split, repeats      # 'dev' is the reference's only special split (face_id.py:533-539, 627)
        self.num_ids, self.specific_ids, self.images_per_id = int(num_ids), (list(specific_ids) if specific_ids is not None else None), 1
        self.placeholder, self.seed, self.synthetic = placeholder, seed, bool(synthetic)
        self.rng = random.Random(seed)
        self._rng_owner = None
        self.img_dict = {}                                      # dense slot -> {'id': stem, 'images': [path]}
        if self.synthetic:
            for slot in range(self.num_ids):
                walk = self.specific_ids[slot] if self.specific_ids is not None and slot < len(self.specific_ids) else slot
                self.img_dict[slot] = {"id": f"synthetic{walk:05d}", "images": [None], "walk": walk}
        else:
            self._load_from_pickle(pickle_path)
        self.ids = sorted(self.img_dict) if only_slots is None else [int(i) for i in only_slots]
        for i in self.ids:
            if i not in self.img_dict:
                raise ValueError(f"only_slots: dense identity slot {i} does not exist (num_ids={self.num_ids})")
        self.faces = {slot: self._load_face(slot) for slot in self.ids}

    def _load_from_pickle(self, path):
        """face_id.py:556-583 `_load_from_pickle` ('1. load train'; reg_ids is 0)"""
        if not path:
            raise ValueError("FaceIdDataset: pickle_path is required (pass synthetic=True for seeded stand-in faces)")
        import os
        with open(path, "rb") as handle:
            pickle_list = pickle.load(handle)
        groups = {}
        for img in pickle_list:
            groups.setdefault(os.path.basename(img).split(".")[0], []).append(img)
        walk_id = use_id = 0
        for stem, images in groups.items():
            if len(self.img_dict) >= self.num_ids:
                break
            if len(images) >= self.images_per_id:
                if self.specific_ids is not None and walk_id not in self.specific_ids:
                    walk_id += 1
                    continue
                self.img_dict[use_id] = {"id": stem, "images": images[:self.images_per_id], "walk": walk_id}
                walk_id += 1
                use_id += 1
        if use_id < self.num_ids:
            raise ValueError("Reach last. Not enough images for num_ids=%d, only %d." % (self.num_ids, use_id))

    def _load_face(self, slot):
        entry = self.img_dict[slot]
        if self.synthetic:                                      # a seeded smooth synthetic "face"
            g = torch.Generator().manual_seed(1000 + entry["walk"])
            low = torch.rand(1, 3, 16, 16, generator=g) * 2 - 1
            return F.interpolate(low, size=self.size, mode="bilinear", align_corners=True)[0].permute(1, 2, 0).contiguous()
        from PIL import Image
        import numpy as np
        img = Image.open(entry["images"][0]).convert("RGB").resize((self.size, self.size), Image.BILINEAR)   # transforms.Resize
        return torch.from_numpy(np.asarray(img).copy()).float() / 127.5 - 1.0

    def _worker_rng(self):
        """the reference draws from the global `random` / `np.random`, which DataLoader reseeds per worker and per epoch; a
        generator copied into every worker would replay one stream 8 times, so each worker (re)seeds its own from the
        loader-provided seed"""
        info = torch.utils.data.get_worker_info()
        key = None if info is None else info.seed
        if key != self._rng_owner and info is not None:
            self.rng = random.Random((self.seed << 32) ^ info.seed)
            self._rng_owner = key
        return self.rng

    def __len__(self):
        return len(self.ids) * self.repeats                     # face_id.py:516-522: every split repeats its list

    def _aug(self, img):
        """flip + colour jitter on the CPU worker (face_id.py:526-532), same parameter draws and formulas as the device path"""
        (flip, b, c, sat, hue), order = self._jitter_params()
        x = (img.flip(1) if flip else img + 0) * 0.5 + 0.5
        gray = lambda t: 0.2989 * t[..., 0] + 0.587 * t[..., 1] + 0.114 * t[..., 2]
        for op in order:                                        # ColorJitter.forward: fn_idx = randperm(4)
            if op == 0:
                x = (x * b).clamp(0, 1)
            elif op == 1:
                x = (c * x + (1 - c) * gray(x).mean()).clamp(0, 1)
            elif op == 2:
                x = (sat * x + (1 - sat) * gray(x).unsqueeze(-1)).clamp(0, 1)
            elif hue != 0.0:
                mx, mn = x.max(-1).values, x.min(-1).values
                d = mx - mn
                s_ = torch.where(mx > 0, d / mx.clamp_min(1e-30), torch.zeros_like(mx))
                dd = torch.where(d > 0, d, torch.ones_like(d))
                r, g, bl = x[..., 0], x[..., 1], x[..., 2]
                h = torch.where(mx == r, (g - bl) / dd, torch.where(mx == g, 2 + (bl - r) / dd, 4 + (r - g) / dd))
                h = (torch.where(d > 0, (h / 6) % 1.0, torch.zeros_like(h)) + hue) % 1.0
                fi = torch.floor(h * 6); f = h * 6 - fi; i = fi.long() % 6
                p_, q, t = (mx * (1 - s_)).clamp(0, 1), (mx * (1 - f * s_)).clamp(0, 1), (mx * (1 - (1 - f) * s_)).clamp(0, 1)
                pick = lambda *a: torch.stack(a, -1).gather(-1, i.unsqueeze(-1)).squeeze(-1)
                x = torch.stack([pick(mx, q, p_, p_, t, mx), pick(t, mx, mx, q, p_, p_), pick(p_, p_, t, mx, mx, q)], -1)
        return x * 2 - 1

    def _bg_geometry(self):
        """the draws of face_id.py:719-727: rh = min(int(h U[0.1,1]), h), rw = min(int(rh U[0.9,1.1]), w), offsets randint"""
        S = self.size
        rh = max(1, min(int(S * self.rng.uniform(0.1, 1.0)), S))
        rw = max(1, min(int(rh * self.rng.uniform(0.9, 1.1)), S))
        pos_h = self.rng.randrange(S - rh) if S > rh else 0
        pos_w = self.rng.randrange(S - rw) if S > rw else 0
        return rh, rw, pos_h, pos_w

    def _add_bg(self, img):
        """face_id.py:712-730: bilinear (align_corners=True) rescale, paste at a random offset on a -1 canvas"""
        rh, rw, y0, x0 = self._bg_geometry()
        small = F.interpolate(img.permute(2, 0, 1)[None], size=(rh, rw), mode="bilinear", align_corners=True)[0].permute(1, 2, 0)
        canvas = torch.full((self.size, self.size, 3), -1.0)
        canvas[y0:y0 + rh, x0:x0 + rw] = small
        return canvas

    def _jitter_params(self):
        """([flip, brightness, contrast, saturation, hue], order) of RandomHorizontalFlip + ColorJitter((0.8,1.2),(0.8,1.2),(0.8,1.2),0.01):
        the flip coin, then ColorJitter.get_params' draws in its order — the permutation of the four ops (fn_idx) first, then
        the four factors"""
        r = self.rng
        flip = 1.0 if r.random() < self.flip_p else 0.0
        if self.split == "dev":                                 # face_id.py:533-539: the 'dev' transform flips but has no ColorJitter
            return [flip, 1.0, 1.0, 1.0, 0.0], [0, 1, 2, 3]
        order = r.sample(range(4), 4)
        return [flip, r.uniform(0.8, 1.2), r.uniform(0.8, 1.2), r.uniform(0.8, 1.2), r.uniform(-0.01, 0.01)], order

    def __getitem__(self, i):
        self._worker_rng()
        ident = self.ids[i % len(self.ids)]                   # dense slot = id_idx of face_id.py:650-653
        face = self.faces[ident]
        img, img3 = self._aug(face), self._aug(face)
        train_img = self._add_bg(img) if self.split != "dev" else img       # face_id.py:627
        caption = self.rng.choice(imagenet_templates_small).format(f"face of {self.placeholder} person")
        return dict(image=train_img, image_ori=dict(faces=torch.cat([img, img3], -1), ids=torch.tensor([ident, ident]), num_ids=1),
                    caption=caption, id_idx=ident, img_idx=i)


class FaceIdDatasetOneShot(FaceIdDatasetStyleGAN3):
    """face_id.py:751-757: same dataset, `pickle_path` required"""

    def __init__(self, pickle_path=None, **kwargs):
        super().__init__(pickle_path, **kwargs)


class DeviceAugmenter:
    """Batch-level counterpart of `__getitem__` with the pixel work on the GPU: the base faces stay resident in HBM, the host
    draws the per-sample parameters — from the dataset's generator, in `__getitem__`'s order (augmentation 1, augmentation 2,
    `_add_bg` geometry, caption), so the same generator state yields the same batch on either path — and native launches
    produce `image` (B,S,S,3), `image_ori.faces` (B,S,S,6) and the ids/captions of face_id.py:598-644. ColorJitter's per-sample op
    order: cb_face_augment applies brightness -> contrast -> saturation -> hue, and every op is the identity at its neutral
    factor (1, 1, 1, 0), so pass k of four runs the k-th op of EACH sample's own permutation with the other three neutral."""

    SLOTS = 4                                                   # pinned staging buffers in flight (one H2D copy per batch)

    def __init__(self, dataset, device="cuda"):
        from ... import lib as _lib
        from ..modules.frozen import context_for
        self.ds, self.dev, self._lib = dataset, torch.device(device), _lib
        self.ctx = context_for(self.dev)
        self.idents = list(dataset.ids)
        self.base = torch.stack([dataset.faces[i] for i in self.idents]).to(self.dev).float().contiguous()     # (n_ids,S,S,3)
        self.idents_dev = torch.tensor(self.idents, dtype=torch.int64, device=self.dev)
        self._stage, self._turn = [None] * self.SLOTS, 0

    def draws(self, indices):
        """host half of a batch: per sample the draws of `__getitem__`, in its order"""
        ds = self.ds
        p1, p3, geom, caps = [], [], [], []
        for _ in indices:
            p1.append(ds._jitter_params()); p3.append(ds._jitter_params())
            if ds.split != "dev":
                geom.append(ds._bg_geometry())
            caps.append(ds.rng.choice(imagenet_templates_small).format(f"face of {ds.placeholder} person"))
        return p1, p3, geom, caps

    @staticmethod
    def passes(params, order):
        """[flip, b, c, s, h] + op order -> the 4 x 5 parameters of the four single-op passes (the flip rides on the first)"""
        rows = []
        for k, op in enumerate(order):
            row = [params[0] if k == 0 else 0.0, 1.0, 1.0, 1.0, 0.0]
            row[1 + op] = params[1 + op]
            rows += row
        return rows

    def _to_device(self, values):
        """ONE asynchronous H2D copy per batch, from a pinned staging buffer: a pageable source would make the copy wait for the
        stream to drain, i.e. put the host's feeder work in series with every training step. A slot is reused SLOTS batches later,
        after its copy event"""
        n = len(values)
        slot = self._turn % self.SLOTS
        self._turn += 1
        entry = self._stage[slot]
        if entry is None or entry[0].numel() < n:
            entry = [torch.empty(max(n, 256), dtype=torch.float32).pin_memory(), None]
            self._stage[slot] = entry
        if entry[1] is not None:
            entry[1].synchronize()
        entry[0][:n] = torch.tensor(values, dtype=torch.float32)
        devp = entry[0][:n].to(self.dev, non_blocking=True)
        entry[1] = torch.cuda.Event()
        entry[1].record(torch.cuda.current_stream(self.dev))
        return devp

    def batch(self, indices):
        L = self._lib
        B = len(indices)
        which = [i % len(self.idents) for i in indices]
        p1, p3, geom, caps = self.draws(indices)
        # staging layout: pass k of four, the batch's first augmentations then its second ones [4][2B][5] | paste integers [B][4] | face index [2B]
        rows1, rows3 = [self.passes(*a) for a in p1], [self.passes(*b) for b in p3]
        flat = [v for k in range(4) for rows in (rows1, rows3) for r in rows for v in r[5 * k:5 * k + 5]]
        flat += [float(v) for g in geom for v in g] + [float(w) for w in which] * 2
        devp = self._to_device(flat)
        o_geom = 40 * B
        o_which = o_geom + 4 * len(geom)
        idx = devp[o_which:o_which + 2 * B].to(torch.int64)
        both = self.base[idx]                                   # (2B,S,S,3): rows 0..B-1 become `img`, rows B..2B-1 `img3`
        for k in range(4):
            both = L.face_augment(self.ctx, both, devp[10 * B * k:10 * B * (k + 1)].view(2 * B, 5))
        img, img3 = both[:B], both[B:]
        image = L.paste_on_canvas(self.ctx, img, devp[o_geom:o_which].view(B, 4).to(torch.int32)) if geom else img
        ident = self.idents_dev[idx[:B]]
        return dict(image=image, caption=caps, id_idx=torch.tensor([self.idents[w] for w in which], dtype=torch.int64),
                    img_idx=torch.tensor(list(indices)),
                    image_ori=dict(faces=torch.cat([img, img3], dim=-1), ids=torch.stack([ident, ident], dim=1),
                                   num_ids=torch.ones(B, dtype=torch.int64)))


class DeviceFeeder:
    """The training loop's default feeder on a GPU: what `DataLoader(dataset, batch_size, shuffle=True)` + 8 CPU workers are in
    the reference (main_id_embed.py:252-260), with the pixel work of every batch done by `DeviceAugmenter` on the GPU the step
    runs on. One pass = one epoch of `len(dataset)` samples in `torch.randperm` order (RandomSampler), last batch kept
    (drop_last=False). `tokenize` (captions -> int64 [B,77]) is applied once per distinct caption and cached: the feeder serves
    81 templates x one placeholder, so steady-state batches carry ready `tokens` and the step never waits for a BPE pass."""

    def __init__(self, dataset, batch_size, device="cuda", shuffle=True, tokenize=None):
        self.ds, self.batch_size, self.shuffle, self.tokenize = dataset, int(batch_size), shuffle, tokenize
        self.aug = DeviceAugmenter(dataset, device)
        self._tok = {}

    def __len__(self):
        return (len(self.ds) + self.batch_size - 1) // self.batch_size

    def _tokens(self, caps):
        rows = []
        for c in caps:
            if c not in self._tok:                              # cached ON the device: a steady-state batch needs no H2D copy for its tokens
                self._tok[c] = self.tokenize([c])[0].to(torch.int64).to(self.aug.dev)
            rows.append(self._tok[c])
        return torch.stack(rows)

    def __iter__(self):
        n = len(self.ds)
        order = torch.randperm(n).tolist() if self.shuffle else list(range(n))
        for i in range(0, n, self.batch_size):
            b = self.aug.batch(order[i:i + self.batch_size])
            if self.tokenize is not None:
                b["tokens"] = self._tokens(b["caption"])
            yield b
