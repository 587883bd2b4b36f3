"""Host-side mirror of ldm/modules/embedding_manager.py `EmbeddingManagerId` (reference :187-532): same constructor keywords
(aigc_id.yaml:23-37), same public surface (`forward`, `save`, `load`, `trainable_parameters`, `embedding_parameters`,
`embedding_neg_loss`, `id_coefficients`, `id_embeddings`) and the same on-disk format. In training the whole forward
(FR-net -> MLP -> L2-norm -> basis -> EMA -> row surgery) runs fused inside cb_trainer_step; this class owns the state."""
import ctypes as C

import torch

from ... import lib as _lib
from .frozen import context_for
from .id_embedding.iresnet import IResNet

# single-token CLIP BPE ids quoted by the reference itself (modules.py:198-262, 02_start_test.sh:106-107)
KNOWN_TOKENS = {"sks": 48136, "ks": 662, "ata": 4236, "tre": 6033, "face": 1710, "a": 320, "photo": 1125, "of": 539}


class MetaIdNet:
    """state holder for meta_net.py:103-377: frozen id_model + the trainable EqualLinear (`stylegan_mlp.net.0`), and with
    `use_rm_mlp` (:173-175) the directly optimised coefficients `coef` (num_ids, es, heads, inner). The trainable tensors are
    torch.nn.Parameter objects (what `trainable_parameters()` hands to the caller's optimiser in the reference); the fused native
    step owns the working copies and `LatentDiffusion.sync_manager_from_trainer` refreshes these."""

    def __init__(self, inner_dim=512, num_embeds_per_token=2, heads=1, fr_dim=512, mlp_depth=1, generator=None, use_rm_mlp=False,
                 num_ids=10):
        assert heads == 1 and mlp_depth == 1, "the shipped configuration (meta_heads 1, meta_mlp_depth 1) is the hot path"
        self.id_model = IResNet((3, 13, 30, 3))
        out = inner_dim * num_embeds_per_token * heads
        P = torch.nn.Parameter
        self.weight = P(torch.randn(out, fr_dim, generator=generator))      # EqualLinear init (meta_net.py:30-32)
        self.bias = P(torch.zeros(out))
        self.use_rm_mlp = bool(use_rm_mlp)
        self.coef = P(torch.randn(num_ids, num_embeds_per_token, heads, inner_dim, generator=generator)) if use_rm_mlp else None

    def parameters(self):
        """the tensors that receive gradients (the frozen id_model's stay None in the reference: meta_net.py:253-264)"""
        return [self.weight, self.bias] + ([self.coef] if self.coef is not None else [])

    def trainable_state_dict(self):
        d = {"stylegan_mlp.net.0.weight": self.weight.data, "stylegan_mlp.net.0.bias": self.bias.data}
        if self.coef is not None:
            d["coef"] = self.coef.data
        return d

    def load_trainable_state_dict(self, sd):
        self.weight.data = sd["stylegan_mlp.net.0.weight"].float().clone()
        self.bias.data = sd["stylegan_mlp.net.0.bias"].float().clone()
        if self.coef is not None and "coef" in sd:
            self.coef.data = sd["coef"].float().clone()


class EmbeddingManagerId:
    def __init__(self, embedder, placeholder_strings=None, initializer_words=None, max_ids=10, num_embeds_per_token=1,
                 momentum=0.9, meta_mlp_depth=2, loss_type=None, meta_inner_dim=512, meta_heads=1, use_rm_mlp=False,
                 test_mode="coefficient", save_fp16=True, **kwargs):
        if loss_type not in (None, "none"):
            raise NotImplementedError("only loss_type 'none' (aigc_id.yaml:31) is on the hot path")
        assert test_mode in ("coefficient", "embedding", "image")
        self.embedder = embedder
        self.placeholder_strings = list(placeholder_strings or ["sks"])
        self.max_ids, self.num_es, self.meta_heads = max_ids, num_embeds_per_token, meta_heads
        self.momentum, self.test_mode, self.save_fp16, self.loss_type = momentum, test_mode, save_fp16, loss_type
        self.use_rm_mlp = bool(use_rm_mlp)
        self.token_dim = embedder.cfg["width"] if hasattr(embedder, "cfg") else 768
        self.inner_dim = meta_inner_dim
        self.training = True
        self.string_to_token_dict = {}
        for s in self.placeholder_strings:
            self.string_to_token_dict[s] = self._token_for(s)
        g = torch.Generator().manual_seed(torch.initial_seed() % (2 ** 31))
        self.meta_id_net = MetaIdNet(meta_inner_dim, num_embeds_per_token, meta_heads, mlp_depth=meta_mlp_depth, generator=g,
                                     use_rm_mlp=use_rm_mlp)
        # dictionaries (embedding_manager.py:225-253): coefficients start as ONE randn block shared by all slots; embeddings start
        # at the initializer word's token embedding (repeated es*h times) or at torch.rand when no initializer word is given
        self.id_coefficients = [torch.randn(num_embeds_per_token, meta_heads, meta_inner_dim, generator=g)] * max_ids
        self.initializer_words = list(initializer_words or [])
        rows = num_embeds_per_token * meta_heads
        if self.initializer_words:
            # the token table arrives with the checkpoint (the reference's CLIP is pretrained at construction): filled by
            # init_id_embeddings() as soon as the embedder's weights are loaded; zeros only until then
            self.id_embeddings = [torch.zeros(rows, self.token_dim) for _ in range(max_ids)]
            self._id_emb_pending = True
        else:
            self.id_embeddings = [torch.rand(rows, self.token_dim, generator=g) for _ in range(max_ids)]
            self._id_emb_pending = False
        self.celeb_embeddings = None

    def _token_for(self, s):
        try:
            ids = self.embedder.tokenize(s)
        except RuntimeError:
            return KNOWN_TOKENS.get(s)
        assert int((ids != 49407).sum()) == 2, f"String '{s}' maps to more than a single token"   # embedding_manager.py:18-24
        return int(ids[0, 1])      # None for strings whose id is unknown offline (only [0] is used when num_ids == 1)

    # ---- reference API
    def embedding_parameters(self):
        return []

    def trainable_parameters(self):
        """embedding_manager.py:432-437: list(self.meta_id_net.parameters()) — torch.nn.Parameter objects"""
        return list(self.meta_id_net.parameters())

    def init_id_embeddings(self, token_table):
        """embedding_manager.py:239-253: every id_embeddings[idx] = embedding(initializer_words[0]) repeated (es*h) times.
        `token_table`: the embedder's token_embedding.weight (vocab, width). No-op once done or after a checkpoint load."""
        if not self._id_emb_pending:
            return
        tok = self._token_for(self.initializer_words[0])
        if tok is None:
            raise RuntimeError(f"the BPE id of initializer word '{self.initializer_words[0]}' is unknown offline")
        row = token_table[tok].detach().float().cpu()
        rows = self.num_es * self.meta_heads
        self.id_embeddings = [row.unsqueeze(0).repeat(rows, 1).clone() for _ in range(self.max_ids)]
        self._id_emb_pending = False

    def embedding_neg_loss(self):
        return 0.0

    def train(self, mode=True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    def save(self, ckpt_path):
        """embedding_manager.py:396-410 — {"id_coefficients": [max_ids x (es, h, inner)]} (fp16 when save_fp16)"""
        cast = (lambda x: x.detach().cpu().half()) if self.save_fp16 else (lambda x: x.detach().cpu().float())
        d = {}
        if self.test_mode == "coefficient":
            d["id_coefficients"] = [cast(x) for x in self.id_coefficients]
        if self.test_mode == "embedding":
            d["id_embeddings"] = [cast(x) for x in self.id_embeddings]
        if self.test_mode == "image":
            d["meta_id_net"] = {k: v.detach().cpu() for k, v in self.meta_id_net.trainable_state_dict().items()}
        torch.save(d, ckpt_path)

    def load(self, ckpt_path):
        """embedding_manager.py:412-426"""
        ckpt = torch.load(ckpt_path, map_location="cpu", weights_only=False)
        if ckpt.get("id_coefficients") is not None:
            self.id_coefficients = [x.float() for x in ckpt["id_coefficients"]]
        if ckpt.get("id_embeddings") is not None:
            self.id_embeddings = [x.float() for x in ckpt["id_embeddings"]]
            self._id_emb_pending = False
        if ckpt.get("meta_id_net") is not None:
            self.meta_id_net.load_trainable_state_dict(ckpt["meta_id_net"])

    # ---- inference-time hook (embedding_manager.py:312-319 test branch + 464-475): coefficients -> embedding -> row surgery
    def _predicted_embedding(self, id_idx, basis, dev):
        """embedding_manager.py:464-475 `_momentum_update` in eval mode: (es*h, W) for one identity"""
        es, K, W = self.num_es, self.inner_dim, self.token_dim
        if self.test_mode != "coefficient":
            return self.id_embeddings[id_idx].to(dev).float().reshape(es, W)
        L, ctx = _lib.load(), context_for(dev)
        coef = self.id_coefficients[id_idx].to(dev).float().reshape(es, K).contiguous()
        z = torch.empty(es, W, device=dev)
        for e in range(es):      # z[e] = coef[e] @ basis[e,1:] + basis[e,0]   (einsum 'e h k, e k c -> e h c')
            d = _lib.GemmDesc()
            d.A = coef[e].data_ptr(); d.B = basis[e, 1:].data_ptr(); d.C = z[e].data_ptr()
            d.M, d.N, d.K = 1, W, K
            d.lda, d.ldb, d.ldc = K, W, W
            d.a_layout, d.b_layout, d.batch = 0, 1, 1
            d.bias = basis[e, 0].data_ptr(); d.alpha, d.beta = 1.0, 0.0
            _lib.check(L.cb_gemm(ctx.h, C.byref(d)))
        return z

    def _embedding_from_faces(self, faces, basis, dev):
        """`test_mode: image` (embedding_manager.py:296-300 with faces given): FR-net + MLP head on every face chunk of
        `faces` (B,S,S,3k) -> z (k, B, es*h, W), the k-th entry belonging to the k-th person of the prompt"""
        L, ctx = _lib.load(), context_for(dev)
        B, S, _, c3 = faces.shape
        k = c3 // 3
        es, K, W = self.num_es, self.inner_dim, self.token_dim
        fr = self.meta_id_net.id_model.native(dev)
        f = faces.to(dev).float().contiguous()
        warped = torch.empty(k * B, S, S, 3, device=dev); small = torch.empty(k * B, fr.in_size, fr.in_size, 3, device=dev)
        ctx.ensure_warp_grid(S)
        _lib.check(L.cb_face_warp_resize(ctx.h, _lib.ptr(f), _lib.ptr(warped), _lib.ptr(small), B, S, k, fr.in_size))
        feat = fr.forward(small)
        Wm, bm = self.meta_id_net.weight.data.to(dev).float().contiguous(), self.meta_id_net.bias.data.to(dev).float().contiguous()
        coef = torch.empty(k * B, es * K, device=dev); z = torch.empty(k * B, es * W, device=dev)
        _lib.check(L.cb_celeb_head_forward(ctx.h, _lib.ptr(feat), _lib.ptr(Wm), _lib.ptr(bm), _lib.ptr(basis), k * B, feat.shape[1], es, K, W,
                                           _lib.ptr(coef), _lib.ptr(z)))
        return z.reshape(k, B, es * self.meta_heads, W)

    def forward(self, tokenized_text, embedded_text, face_image=None, img_ori=None, celeb_embeddings=None):
        if img_ori is None:
            return embedded_text
        if self.training:
            raise RuntimeError("the training-mode manager forward is fused into the native training step "
                               "(LatentDiffusion.training_step -> cb_trainer_step); call it through the trainer")
        dev = embedded_text.device
        ctx = context_for(dev)
        L = _lib.load()
        B, T, W = embedded_text.shape
        reps = self.num_es * self.meta_heads
        basis = (celeb_embeddings if celeb_embeddings is not None else self.celeb_embeddings).to(dev).float().contiguous()
        ids, num_ids = img_ori["ids"], img_ori.get("num_ids")
        # one / two / three persons per prompt (embedding_manager.py:323-391): person k is written where placeholder_strings[k] occurs
        counts = [int(num_ids[b]) if num_ids is not None else 1 for b in range(B)]
        nph = max(counts)
        if nph < 1 or nph > 3 or nph > len(self.placeholder_strings):
            raise ValueError(f"num_ids={nph} needs that many placeholder_strings (have {self.placeholder_strings}); 1..3 are supported")
        z = torch.zeros(B, nph * reps, W, device=dev)
        ph = torch.full((B, nph), -1, dtype=torch.int64)
        from_faces = None
        if self.test_mode == "image" and img_ori.get("faces") is not None:
            from_faces = self._embedding_from_faces(img_ori["faces"], basis, dev)
            if from_faces.shape[0] < nph:
                raise ValueError(f"{nph} persons in the prompt but only {from_faces.shape[0]} face images were given")
        for b in range(B):
            for k in range(counts[b]):
                tok = self.string_to_token_dict[self.placeholder_strings[k]]
                if tok is None:
                    raise RuntimeError(f"the BPE id of placeholder '{self.placeholder_strings[k]}' is unknown offline")
                ph[b, k] = tok
                z[b, k * reps:(k + 1) * reps] = from_faces[k, b] if from_faces is not None else self._predicted_embedding(int(ids[b][k]), basis, dev)
        ph = ph.to(dev)
        src = embedded_text.contiguous()
        out = torch.empty_like(src)
        mp = torch.empty(B * T, dtype=torch.int32, device=dev)
        toks = tokenized_text.contiguous()
        _lib.check(L.cb_row_surgery_multi(ctx.h, _lib.ptr(src), _lib.ptr(z), _lib.ptr(toks, torch.int64),
                                          _lib.ptr(out), C.c_void_p(mp.data_ptr()), B, T, W, reps, _lib.ptr(ph, torch.int64), nph))
        # the reference MUTATES embedded_text in place and returns it (embedding_manager.py:347-394); rows shift, so the surgery
        # itself cannot run in place
        embedded_text.copy_(out)
        return embedded_text

    __call__ = forward
