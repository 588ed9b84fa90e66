"""Build container only: run the REAL reference augmenter (oracle/ref_loader.load_reference_augmenter) in eval mode on
seeded inputs and commit inputs + outputs as a data-only fixture (tests/golden/aug_small.npz).

    python -m oracle.gen_golden_aug
"""
import os

import numpy as np
import torch

from oracle import augmenter as OA
from oracle import ref_loader as RL

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "aug_small.npz")


def main():
    cls = RL.load_reference_augmenter()
    NZ, Z, D, ND, A, B, scale = 6, 3, 52, 20, 3, 21, 0.1       # D / 5 = 10: rows that are not a multiple of 4 floats
    sd = OA.random_state_dict(NZ, Z, D, ND, seed=31)
    m = cls(noise_dim=NZ, latent_dim=Z, input_dim=D, n_dim=ND)
    m.load_state_dict(sd)
    m.eval()
    g = torch.Generator().manual_seed(5)
    x = (torch.rand(B, D, generator=g) < 0.3).float() * torch.randn(B, D, generator=g).abs() * 3
    out = {"dims": np.array([NZ, Z, D, ND, A, B]), "scale": np.array(scale), "x": x.numpy()}
    for k, v in sd.items():
        out["sd/" + k] = v.numpy()
    # the reference draws z (randn) and then eps (randn_like) from torch's global generator
    torch.manual_seed(77)
    z0 = torch.randn(A, B, NZ)
    eps = torch.randn(A, B, Z)
    torch.manual_seed(77)
    with torch.no_grad():
        s, xa = m(x.expand(A, -1, -1), True, scale)
    out.update({"b/z0": z0.numpy(), "b/eps": eps.numpy(), "b/s": s.numpy(), "b/x_aug": xa.numpy()})
    torch.manual_seed(78)
    z1 = torch.randn(B, NZ)
    e1 = torch.randn(B, Z)
    torch.manual_seed(78)
    with torch.no_grad():
        s1, xa1 = m(x, False, 1.0)
    out.update({"u/z0": z1.numpy(), "u/eps": e1.numpy(), "u/s": s1.numpy(), "u/x_aug": xa1.numpy()})
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
