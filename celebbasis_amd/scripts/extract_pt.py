"""Counterpart of the reference's scripts/extract_pt.py (:86-139): turn a trained `embeddings_gs-*.pt` into the per-identity
interchange files

    <save_celeb_basis_path>                      celeb basis (es, 1+K, 768)                       (modules.py:626-630)
    <folder>/id_coefficient_{:03d}.pt            (es, h, K) coefficients                          (extract_pt.py:127-139)
    <folder>/id_embedding_{:03d}.pt              (es*h, 768) = einsum('e h k, e k c -> e h c', coef, basis[:,1:]) + basis[:,0]
    (…_fp16_… variants with --save_fp16)

The contraction runs through the native library (cb_gemm) like every other piece of arithmetic in this package, so the tool
needs the MI355X just as the reference's needs its GPU for the model load.

    python -m celebbasis_amd.scripts.extract_pt --embedding_path logs/.../embeddings_gs-800.pt --celeb_basis weights/celeb_basis.pt
"""
import argparse
import os

import torch


def ti_embeddings(id_coefficients, celeb_basis, device="cuda"):
    """[(es, h, K)] * max_ids, (es, 1+K, W) -> [(es*h, W)] * max_ids"""
    import ctypes as C
    from .. import lib as _lib
    from ..ldm.modules.frozen import context_for
    ctx, L = context_for(device), _lib.load()
    basis = celeb_basis.to(device).float().contiguous()
    es, k1, W = basis.shape
    out = []
    for coef in id_coefficients:
        c = coef.to(device).float().contiguous()
        e, h, K = c.shape
        assert e == es and K == k1 - 1, "coefficients do not match the basis"
        z = torch.empty(es, h, W, device=device)
        for i in range(es):
            d = _lib.GemmDesc()
            d.A, d.B, d.C = c[i].data_ptr(), basis[i, 1:].data_ptr(), z[i].data_ptr()
            d.M, d.N, d.K = h, W, K
            d.lda, d.ldb, d.ldc = K, W, W
            d.a_layout, d.b_layout, d.batch = 0, 1, 1
            d.alpha, d.beta = 1.0, 0.0
            d.bias = basis[i, 0].data_ptr()
            _lib.check(L.cb_gemm(ctx.h, C.byref(d)))
        out.append(z.reshape(es * h, W))
    torch.cuda.synchronize()
    return out


def extract(id_coefficients, celeb_basis, save_celeb_basis_path, folder, save_fp16=False, device="cuda"):
    """extract_pt.py:96-139"""
    torch.save(celeb_basis.detach().cpu(), save_celeb_basis_path)
    z = ti_embeddings(id_coefficients, celeb_basis, device)
    os.makedirs(folder, exist_ok=True)
    tag, cast = ("_fp16", lambda t: t.cpu().half()) if save_fp16 else ("", lambda t: t.cpu())
    for idx, (zi, ci) in enumerate(zip(z, id_coefficients)):
        torch.save(cast(zi), os.path.join(folder, "id_embedding{}_{:03d}.pt".format(tag, idx)))
        torch.save(cast(ci.detach()), os.path.join(folder, "id_coefficient{}_{:03d}.pt".format(tag, idx)))
    return z


def main(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--embedding_path", required=True, help="embeddings_gs-*.pt written by training (embedding_manager.py:396-410)")
    p.add_argument("--celeb_basis", required=True, help="celeb basis tensor (es, 1+K, 768), e.g. a celeb_basis.pt of the reference")
    p.add_argument("--save_celeb_basis_path", default="./weights/celeb_basis.pt")
    p.add_argument("--save_ti_embedding_save_folder", default="./weights/ti_id_embeddings")
    p.add_argument("--save_fp16", action="store_true")
    a = p.parse_args(argv)
    ckpt = torch.load(a.embedding_path, map_location="cpu", weights_only=False)
    coefs = [c.float() for c in ckpt["id_coefficients"]]
    basis = torch.load(a.celeb_basis, map_location="cpu", weights_only=False).float()
    os.makedirs(os.path.dirname(os.path.abspath(a.save_celeb_basis_path)), exist_ok=True)
    z = extract(coefs, basis, a.save_celeb_basis_path, a.save_ti_embedding_save_folder, a.save_fp16)
    print(f"TI ID embeddings saved to {a.save_ti_embedding_save_folder}, shape={tuple(z[0].shape)}")


if __name__ == "__main__":
    main()
