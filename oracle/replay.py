"""CPU ORACLE — TEST INFRASTRUCTURE ONLY. The seeded replay stream the parity tests and oracle/gen_horizon.py share: one batch per
optimiser step with the tensors FaceIdDatasetStyleGAN3.__getitem__ yields (face_id.py:598-644: `image` HWC in [-1, 1], `faces` =
two augmentations concatenated on the channel axis, `ids` = [id, id]) plus the per-step randomness of p_losses (t, noise) and of
the posterior sample, all drawn from a torch CPU generator keyed by `seed` so that the GPU run and the oracle see the same bytes."""
import torch


def make_batch(B, S, lat, vocab, placeholder, seed, n_ids):
    g = torch.Generator().manual_seed(seed)
    tokens = torch.randint(0, vocab - 30, (B, 77), generator=g)
    for b in range(B):
        tokens[b, 3 + 2 * b] = placeholder
    return dict(image=torch.rand(B, S, S, 3, generator=g) * 2 - 1, faces=torch.rand(B, S, S, 6, generator=g) * 2 - 1,
                ids=torch.stack([torch.tensor([1 + (b % (n_ids - 1))] * 2) for b in range(B)]), tokens=tokens,
                t=torch.randint(0, 1000, (B,), generator=g), noise=torch.randn(B, 4, lat, lat, generator=g),
                post_noise=torch.randn(B, 4, lat, lat, generator=g))
