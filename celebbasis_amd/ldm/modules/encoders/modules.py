"""Host-side mirror of ldm/modules/encoders/modules.py `FrozenCLIPEmbedder` (reference :157-631).

The reference wraps HF `CLIPTextModel` and monkey-patches its embedding layer so the embedding manager runs between the
token-embedding lookup and the position embedding (modules.py:176-298). Here the text tower runs in libcelebbasis_hip.so
(cb_clip_*) with the same hook point; state-dict keys are the HF ones under `transformer.text_model.`.
The celeb basis (modules.py:472-624) is normally shipped as data (`celeb_embeddings`, shape (es, 1+K, 768), e.g. the
`celeb_basis.pt` that scripts/extract_pt.py writes): coefficients are only meaningful against the basis they were trained
with. `_get_celeb_embeddings` rebuilds it from the name list when the BPE vocab and the token table are available
(`celebbasis_amd.basis`, pinned against the reference's own method in tests/golden/basis_mini.pt).
"""
import torch

from .... import lib as _lib
from ..frozen import FrozenModule


class FrozenCLIPEmbedder(FrozenModule):
    PREFIX = "transformer.text_model."

    def __init__(self, version="openai/clip-vit-large-patch14", device="cuda", max_length=77, celeb_txt=None, use_celeb=False,
                 use_svd=False, n_components=512, rm_repeats=True, use_sample_reduce=False, n_samples=513, use_flatten=True,
                 num_embeds_per_token=2, vocab=49408, width=768, layers=12, heads=12, mlp=3072):
        self.cfg = dict(vocab=vocab, width=width, layers=layers, heads=heads, ctx_len=max_length, mlp=mlp)
        self.version, self.device, self.max_length = version, device, max_length
        self.use_celeb, self.n_components, self.num_embeds_per_token = use_celeb, n_components, num_embeds_per_token
        self.celeb_txt, self.use_svd, self.rm_repeats, self.use_flatten = celeb_txt, use_svd, rm_repeats, use_flatten
        if use_sample_reduce:
            raise NotImplementedError("use_sample_reduce (off in aigc_id.yaml:111) is outside the CelebBasis path")
        # opt-in only (main.py --synthetic, tests): without the BPE vocabulary on disk strings are an ERROR, because hashed
        # stand-in ids would condition real SD weights on garbage and still "succeed"
        self.allow_offline_tokens = False
        self.celeb_embeddings = None          # (es, 1+n_components, width); set by load_celeb_basis / a checkpoint
        self._tokenizer = None
        super().__init__({self.PREFIX + k: v for k, v in self.param_shapes(self.cfg).items()})

    @staticmethod
    def param_shapes(cfg):
        w, S = cfg["width"], {}
        S["embeddings.token_embedding.weight"] = (cfg["vocab"], w)
        S["embeddings.position_embedding.weight"] = (cfg["ctx_len"], w)
        for i in range(cfg["layers"]):
            p = f"encoder.layers.{i}."
            for proj in ("q_proj", "k_proj", "v_proj", "out_proj"):
                S[p + f"self_attn.{proj}.weight"] = (w, w); S[p + f"self_attn.{proj}.bias"] = (w,)
            for ln in ("layer_norm1", "layer_norm2"):
                S[p + ln + ".weight"] = (w,); S[p + ln + ".bias"] = (w,)
            S[p + "mlp.fc1.weight"] = (cfg["mlp"], w); S[p + "mlp.fc1.bias"] = (cfg["mlp"],)
            S[p + "mlp.fc2.weight"] = (w, cfg["mlp"]); S[p + "mlp.fc2.bias"] = (w,)
        S["final_layer_norm.weight"] = (w,); S["final_layer_norm.bias"] = (w,)
        return S

    def load_state_dict(self, sd, strict=True, prefix=""):
        sd = {k: v for k, v in sd.items() if "position_ids" not in k}
        return super().load_state_dict(sd, strict=strict, prefix=prefix)

    def load_celeb_basis(self, basis):
        es, k1, w = basis.shape
        assert es == self.num_embeds_per_token and w == self.cfg["width"], "basis shape (es, 1+K, width) expected"
        self.celeb_embeddings = basis.detach().float().contiguous()

    def freeze(self):
        return self

    def _get_celeb_embeddings(self, n_components=None, tokenize=None):
        """modules.py:472-624: build the (es, 1+K, width) basis from `celeb_txt` and the loaded token-embedding table.
        `tokenize(name) -> [ids]` overrides the BPE tokenizer (which needs the vocab files on disk)."""
        from .... import basis as _basis
        names = _basis.read_names(self.celeb_txt, self.rm_repeats)
        if tokenize is None:
            prev, self.allow_offline_tokens = self.allow_offline_tokens, False   # stand-in ids would give a meaningless basis
            try:
                rows = [self.tokenize(n)[0].tolist() for n in names]
            finally:
                self.allow_offline_tokens = prev
        else:
            rows = [[49406] + list(tokenize(n)) + [49407] for n in names]
        table = self.state_dict()[self.PREFIX + "embeddings.token_embedding.weight"]
        self.celeb_embeddings = _basis.build_celeb_basis(table, rows, self.n_components if n_components is None else n_components,
                                                         self.num_embeds_per_token, self.use_svd, self.use_flatten)
        return self.celeb_embeddings

    def save_celeb_embeddings(self, pth_path):
        """modules.py:626-630"""
        assert isinstance(self.celeb_embeddings, torch.Tensor), "No celeb_embeddings!"
        torch.save(self.celeb_embeddings.cpu(), pth_path)

    def _build(self, ctx):
        return _lib.Clip(ctx, _lib.WeightTable(self.state_dict()), _lib.clip_cfg(**self.cfg), prefix=self.PREFIX)

    # -- tokenizer: HF CLIPTokenizer when its vocab files are on disk; pre-tokenised ids always work
    def tokenize(self, text):
        if isinstance(text, torch.Tensor):
            return text.to(torch.int64)
        if self._tokenizer is None:
            try:
                from transformers import CLIPTokenizer
                tok = CLIPTokenizer.from_pretrained(self.version, local_files_only=True)
                probe = tok("a photo of", padding="max_length", max_length=self.max_length, return_tensors="pt")["input_ids"][0, :4].tolist()
                if probe != [49406, 320, 1125, 539]:      # known answers, modules.py:198-262
                    raise RuntimeError("tokenizer vocab is not the CLIP BPE vocab")
                self._tokenizer = tok
            except Exception as e:  # no vocab on disk and no network
                if not self.allow_offline_tokens:
                    raise RuntimeError("CLIP BPE vocab files are not available offline: pass pre-tokenised int64 ids [B,77] "
                                       "(e.g. batch['tokens']) instead of strings, or opt in to hashed stand-in ids for synthetic "
                                       "runs (main.py --synthetic / cond_stage_model.allow_offline_tokens = True)") from e
                return self._offline_tokens([text] if isinstance(text, str) else list(text))
        enc = self._tokenizer(text, truncation=True, max_length=self.max_length, return_length=True,
                              return_overflowing_tokens=False, padding="max_length", return_tensors="pt")
        return enc["input_ids"]

    def _offline_tokens(self, texts):
        """Stand-in used ONLY when the BPE vocab is not on disk (this image has no network): words the reference itself quotes
        (modules.py:198-262) keep their true ids, every other word gets a stable pseudo id. Meaningless with real SD weights —
        with those, the vocab files are in the local HF cache and the real tokenizer above is used."""
        import zlib
        known = {"a": 320, "photo": 1125, "of": 539, "face": 1710, "sks": 48136, "ks": 662, "ata": 4236, "tre": 6033, "person": 2533}
        if not getattr(self, "_warned", False):
            print("[celebbasis_amd] CLIP BPE vocab not found offline: using stand-in token ids for unknown words", flush=True)
            self._warned = True
        out = torch.full((len(texts), self.max_length), 49407, dtype=torch.int64)
        for b, t in enumerate(texts):
            ids = [49406] + [known.get(w, 1000 + zlib.crc32(w.encode()) % 39000) for w in t.lower().replace("-", " ").split()]
            ids = ids[: self.max_length - 1]
            out[b, :len(ids)] = torch.tensor(ids)
        return out

    def forward(self, text, embedding_manager=None, face_img=None, image_ori=None, **kwargs):
        """same hook order as modules.py:176-298: token embedding -> embedding manager -> (+position, layers, final LN)"""
        tokens = self.tokenize(text).to(self.device if not isinstance(text, torch.Tensor) else text.device).contiguous()
        h = self.native(tokens.device)
        embedded = h.embed_tokens(tokens)
        if embedding_manager is not None:
            embedded = embedding_manager(tokens, embedded, face_img, image_ori, self.celeb_embeddings)
        embedded = embedded.contiguous()
        return h.forward(embedded, need_grad=bool(kwargs.get("need_grad", False)))

    def encode(self, text, **kwargs):
        return self(text, **kwargs)

    __call__ = forward
