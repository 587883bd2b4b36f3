"""Feeder counterpart of ldm/data/face_id.py `FaceIdDatasetStyleGAN3` / `FaceIdDatasetOneShot` (reference :471-757).
It produces the same sample dict (face_id.py:598-644): image (512,512,3) f32 in [-1,1], image_ori {faces (512,512,6), ids (2,),
num_ids}, caption, id_idx, img_idx — with the same 81-template caption list and `diff_cnt=0 => id_cnt=2`.
Augmentations (flip, colour jitter, `_add_bg` rescale-paste) are restated on torch tensors; bit-exact torchvision parity is not
a goal: parity of the training step is defined on a replay stream. Without a dataset on disk the feeder
synthesises seeded faces so the rest of the pipeline can run offline."""
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
    def __init__(self, pickle_path=None, split="train", num_ids=1, specific_ids=None, images_per_id=1, repeats=1000, reg_ids=1000,
                 reg_repeats=0, diff_cnt=0, size=512, flip_p=0.5, placeholder="sks", seed=23, **ignored):
        self.size, self.flip_p, self.split, self.repeats = size, flip_p if split == "train" else 0.0, split, repeats
        self.ids = list(specific_ids) if specific_ids else list(range(1, num_ids + 1))
        self.placeholder = placeholder
        self.rng = random.Random(seed)
        self.faces = self._load(pickle_path)

    def _load(self, path):
        faces = {}
        try:
            with open(path, "rb") as f:
                items = pickle.load(f)
            from PIL import Image
            import numpy as np
            for i, ident in enumerate(self.ids):
                img = Image.open(items[(ident - 1) % len(items)]).convert("RGB").resize((self.size, self.size), Image.BICUBIC)
                faces[ident] = torch.from_numpy(np.asarray(img)).float() / 127.5 - 1.0
        except Exception:
            for ident in self.ids:       # offline: a seeded smooth synthetic "face"
                g = torch.Generator().manual_seed(1000 + ident)
                low = torch.rand(1, 3, 16, 16, generator=g) * 2 - 1
                faces[ident] = F.interpolate(low, size=self.size, mode="bilinear", align_corners=True)[0].permute(1, 2, 0).contiguous()
        return faces

    def __len__(self):
        return len(self.ids) * (self.repeats if self.split == "train" else 1)

    def _aug(self, img):
        """flip + mild brightness/contrast/saturation jitter in [0.8, 1.2] (face_id.py:526-532)"""
        if self.rng.random() < self.flip_p:
            img = img.flip(1)
        x = (img + 1) * 0.5
        b, c, s = (self.rng.uniform(0.8, 1.2) for _ in range(3))
        x = (x * b).clamp(0, 1)
        m = x.mean()
        x = ((x - m) * c + m).clamp(0, 1)
        gray = x.mean(dim=-1, keepdim=True)
        x = ((x - gray) * s + gray).clamp(0, 1)
        return x * 2 - 1

    def _add_bg(self, img):
        """face_id.py:712-730: rescale by U[0.1,1]·aspect U[0.9,1.1], paste at a random offset on a -1 canvas"""
        S = self.size
        sc = self.rng.uniform(0.1, 1.0)
        h = max(8, min(S, int(S * sc)))
        w = max(8, min(S, int(S * sc * self.rng.uniform(0.9, 1.1))))
        small = F.interpolate(img.permute(2, 0, 1)[None], size=(h, w), mode="bilinear", align_corners=False)[0].permute(1, 2, 0)
        canvas = torch.full((S, S, 3), -1.0)
        y0, x0 = self.rng.randint(0, S - h), self.rng.randint(0, S - w)
        canvas[y0:y0 + h, x0:x0 + w] = small
        return canvas

    def __getitem__(self, i):
        ident = self.ids[i % len(self.ids)]
        face = self.faces[ident]
        img, img3 = self._aug(face), self._aug(face)
        train_img = self._add_bg(img) if self.split == "train" else img
        caption = self.rng.choice(imagenet_templates_small).format(f"face of {self.placeholder} person")
        return dict(image=train_img, image_ori=dict(faces=torch.cat([img, img3], -1), ids=torch.tensor([ident, ident]), num_ids=1),
                    caption=caption, id_idx=ident, img_idx=i)


class FaceIdDatasetOneShot(FaceIdDatasetStyleGAN3):
    pass
