"""CPU ORACLE — TEST INFRASTRUCTURE ONLY. The seeded replay stream the parity tests and oracle/gen_horizon.py share: one batch per
optimiser step with the tensors FaceIdDatasetStyleGAN3.__getitem__ yields (face_id.py:598-644: `image` HWC in [-1, 1], `faces` =
two augmentations concatenated on the channel axis, `ids` = [id, id]) plus the per-step randomness of p_losses (t, noise) and of
the posterior sample, all drawn from a torch CPU generator keyed by `seed` so that the GPU run and the oracle see the same bytes."""
import torch


_BLUR = {}


def low_pass(x_hwc, sigma=4.0, radius=12):
    """separable Gaussian blur of [B, S, S, C] noise (reflect borders, float64 accumulation, rescaled to max |x| = 1): "photo-like" faces —
    neighbouring pixels differ by ~3e-2 instead of O(1), so the fp32 rounding of a warp coordinate (6e-5 pixel at 512 px) moves a sampled
    value by ~1e-6 instead of ~3e-5 (tools/probes/probe_warp_rounding.py). Written as two float64 matrix products with the S x S banded
    Toeplitz matrix of the kernel (16x faster than conv2d in float64; same bytes as the reflect-padded convolution)."""
    S = x_hwc.shape[1]
    if (S, sigma, radius) not in _BLUR:
        k = torch.arange(-radius, radius + 1, dtype=torch.float64)
        ker = torch.exp(-k ** 2 / (2.0 * sigma ** 2)); ker = ker / ker.sum()
        idx = torch.arange(S)[:, None] + torch.arange(-radius, radius + 1)[None, :]                  # source pixel of tap j for output pixel i
        idx = torch.where(idx < 0, -idx, idx); idx = torch.where(idx >= S, 2 * (S - 1) - idx, idx)   # reflect, edge not repeated
        T = torch.zeros(S, S, dtype=torch.float64)
        T.index_put_((torch.arange(S)[:, None].expand_as(idx), idx), ker[None, :].expand(S, -1), accumulate=True)
        _BLUR[(S, sigma, radius)] = T
    T = _BLUR[(S, sigma, radius)]
    x = x_hwc.double().permute(0, 3, 1, 2)                    # [B, C, H, W]
    x = torch.matmul(T, torch.matmul(x, T.t()))               # along W, then along H
    x = x.permute(0, 2, 3, 1)
    return (x / x.abs().amax()).float().contiguous()


FIXTURE_HOST_GRID_SHA1 = {512: "8ce7204e5307"}


def fixture_host_warp_grid(S=512):
    """the face warp's sampling grid (meta_net.py:253-257: F.affine_grid of the constant trans_matrix, align_corners=True) AS THE HOST THAT
    GENERATED tests/golden/horizon_sd14_*.pt EVALUATED IT. torch's CPU affine_grid = linspace + a K = 3 bmm; the linspace is the same on
    every host, the bmm's rounding is not (tools/probes/probe_affine_grid_host.py: the fixture host's result equals
    fma(t2, 1, fma(t1, y, t0 * x)) bit for bit, the GPU box's EPYC differs from it in 14 % of the elements by one ulp = up to 6e-5 pixel).
    A run that is compared with those trajectories step by step samples the faces on THIS table; the formula is evaluated with
    float64-emulated fused multiply-adds, so it gives the same bytes everywhere (digest checked)."""
    import hashlib
    import numpy as np
    from oracle.cb_oracle import TRANS_MATRIX
    f32 = np.float32
    fma = lambda a, b, c: (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)     # exact product, one rounding
    lin = torch.linspace(-1, 1, S).numpy()
    if S == 512 and hashlib.sha1(lin.tobytes()).hexdigest()[:12] != "0aa2f976c8dd":
        # a host whose torch rounds linspace differently (another vector width): the fixture host's 512 values are kept as data
        import json
        import os
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "warp_linspace_512.json")) as f:
            lin = np.array([float.fromhex(v) for v in json.load(f)["values"]], dtype=f32)
    th = np.array(TRANS_MATRIX, dtype=f32)
    X = np.broadcast_to(lin[None, :], (S, S)).astype(f32); Y = np.broadcast_to(lin[:, None], (S, S)).astype(f32)
    g = np.stack([fma(np.full_like(X, th[r, 2]), np.ones_like(X), fma(np.full_like(X, th[r, 1]), Y, (th[r, 0] * X).astype(f32))) for r in range(2)], -1)
    if S in FIXTURE_HOST_GRID_SHA1:
        sha = hashlib.sha1(np.ascontiguousarray(g).tobytes()).hexdigest()[:12]
        assert sha == FIXTURE_HOST_GRID_SHA1[S], f"fixture-host warp grid digest {sha} != {FIXTURE_HOST_GRID_SHA1[S]}"
    return torch.from_numpy(np.ascontiguousarray(g))


def make_batch(B, S, lat, vocab, placeholder, seed, n_ids, photo=False):
    g = torch.Generator().manual_seed(seed)
    tokens = torch.randint(0, vocab - 30, (B, 77), generator=g)
    for b in range(B):
        tokens[b, 3 + 2 * b] = placeholder
    if photo:                                # same draws, the two face crops low-pass filtered (the `image` feeds the VAE: no sub-pixel sampling there)
        image = torch.rand(B, S, S, 3, generator=g) * 2 - 1
        faces = low_pass(torch.rand(B, S, S, 6, generator=g) * 2 - 1)
        return dict(image=image, faces=faces, ids=torch.stack([torch.tensor([1 + (b % (n_ids - 1))] * 2) for b in range(B)]), tokens=tokens,
                    t=torch.randint(0, 1000, (B,), generator=g), noise=torch.randn(B, 4, lat, lat, generator=g),
                    post_noise=torch.randn(B, 4, lat, lat, generator=g))
    return dict(image=torch.rand(B, S, S, 3, generator=g) * 2 - 1, faces=torch.rand(B, S, S, 6, generator=g) * 2 - 1,
                ids=torch.stack([torch.tensor([1 + (b % (n_ids - 1))] * 2) for b in range(B)]), tokens=tokens,
                t=torch.randint(0, 1000, (B,), generator=g), noise=torch.randn(B, 4, lat, lat, generator=g),
                post_noise=torch.randn(B, 4, lat, lat, generator=g))


def ref_stream_batch(s, lat, S=64, B=2):
    """element s of the replay stream that oracle/gen_golden.py::step_harness feeds the REFERENCE's composed step (the default MLP
    run): images from seed 2000 + s, posterior noise from manual_seed(900 + s), (t, noise) from manual_seed(1000 + s) in the order
    LatentDiffusion.forward / p_losses draw them (ddpm.py:926-936, 1069-1075). Uses the global torch RNG like the reference does."""
    gi = torch.Generator().manual_seed(2000 + s)
    image = torch.rand(B, S, S, 3, generator=gi) * 2 - 1
    faces = torch.rand(B, S, S, 6, generator=gi) * 2 - 1
    tokens = torch.full((B, 77), 49407, dtype=torch.long); tokens[:, 0] = 49406
    tokens[0, 1:9] = torch.tensor([320, 1125, 539, 320, 1710, 539, 48136, 2533]); tokens[1, 1:5] = torch.tensor([1125, 48136, 1125, 320])
    torch.manual_seed(900 + s)
    post_noise = torch.randn(B, 4, lat, lat)
    torch.manual_seed(1000 + s)
    # p_losses draws randn_like(x_start), and the reference's latent is channels_last (the encoder is fed a permuted HWC image): the fill
    # of a non-contiguous tensor walks storage order with the scalar sampler, so the draw has to be made on the same memory format
    t = torch.randint(0, 1000, (B,)).long()
    noise = torch.randn_like(torch.empty(B, 4, lat, lat).contiguous(memory_format=torch.channels_last)).contiguous()
    return dict(image=image, faces=faces, ids=torch.tensor([[1, 1], [2, 2]]), tokens=tokens, t=t, noise=noise, post_noise=post_noise)
