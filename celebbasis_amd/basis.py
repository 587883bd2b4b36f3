"""Celeb-basis construction (reference `FrozenCLIPEmbedder._get_celeb_embeddings`,
ldm/modules/encoders/modules.py:472-624) — an init-time, one-off host computation (a 646x768 SVD), not part of the step.

The basis for pseudo-word position j (j = 0 .. num_embeds_per_token-1) is built from the CLIP *token embeddings* of the
(j+1)-th BPE token of every celebrity name that has one:

    X_j  : (k_j, W) rows = token_embedding[tok] for names sorted by string, duplicates of the NAME removed (rm_repeats)
    mean : X_j.mean(0);  U S V^T = svd(X_j - mean);  basis_j = [mean; V^T[:n_components]]         -> (1 + K, W)

Two behaviours of the reference that matter for bit-compatibility with bases it has written (`celeb_basis.pt`) are kept:
  * its per-column "seen token" sets hold torch tensors, whose hash is the object id, so repeated TOKENS are never
    dropped in practice (modules.py:518-534) — `dedupe_tokens=False` reproduces that; `True` gives the filtering the code
    reads as if it intended;
  * the singular vectors come straight from LAPACK gesdd (`torch.svd`), signs included. `sign="lapack"` keeps them;
    `sign="positive_max"` flips every basis row so that its largest-magnitude entry is positive (a convention that is stable
    across BLAS builds — use it only for bases you also train against).
"""
import torch


def read_names(celeb_txt, rm_repeats=True):
    """modules.py:474-486: one name per line; optional de-duplication; sorted"""
    with open(celeb_txt, "r") as f:
        names = f.read().splitlines()
    return sorted(set(names)) if rm_repeats else sorted(names)


def name_token_columns(token_rows, special_from=49406, flatten=False, dedupe_tokens=False):
    """token_rows: per name (already in sorted-name order) the int ids of its padded row (BOS, tokens, EOS/pad ...) or just
    the name's own tokens. Returns a list of columns; column c lists (row index, position) pairs of the c-th non-empty token
    position (modules.py:513-545). With flatten=True there is a single column walking every name's tokens row-major."""
    rows = [[int(t) for t in r] for r in token_rows]
    width = max((len(r) for r in rows), default=0)
    cols = []
    if flatten:
        seen, col = set(), []
        for i, r in enumerate(rows):
            for j, t in enumerate(r):
                if t >= special_from or (dedupe_tokens and t in seen):
                    continue
                col.append((i, j)); seen.add(t)
        return [col]
    for j in range(width):
        seen, col = set(), []
        for i, r in enumerate(rows):
            if j >= len(r) or r[j] >= special_from or (dedupe_tokens and r[j] in seen):
                continue
            col.append((i, j)); seen.add(r[j])
        if col:
            cols.append(col)
    return cols


def build_celeb_basis(token_embedding, token_rows, n_components=512, num_embeds_per_token=2, use_svd=True, use_flatten=False,
                      dedupe_tokens=False, sign="lapack", special_from=49406):
    """-> (num_embeds_per_token, 1 + n_components, W) float32 (modules.py:547-619); raises if fewer token columns exist than
    pseudo-word positions are requested."""
    table = token_embedding.detach().float().cpu()
    cols = name_token_columns(token_rows, special_from, use_flatten, dedupe_tokens)
    out = []
    for col in cols[:num_embeds_per_token]:
        x = torch.stack([table[int(token_rows[i][j])] for (i, j) in col])           # (k_j, W)
        if use_svd:
            mean = x.mean(dim=0, keepdim=True)
            _, _, v = torch.svd(x - mean, some=False)                               # v: (W, W), columns = right-singular vectors
            vt = v.t()[:n_components]
            if sign == "positive_max":
                lead = vt.gather(1, vt.abs().argmax(dim=1, keepdim=True))
                vt = vt * torch.where(lead < 0, -torch.ones_like(lead), torch.ones_like(lead))
            elif sign != "lapack":
                raise ValueError("sign must be 'lapack' or 'positive_max'")
            x = torch.cat([mean, vt], dim=0)
        out.append(x.unsqueeze(0))
    if use_flatten:
        out = out * num_embeds_per_token
    if len(out) < num_embeds_per_token:
        raise ValueError(f"only {len(out)} token columns are populated but num_embeds_per_token={num_embeds_per_token}")
    return torch.cat(out, dim=0).contiguous()
