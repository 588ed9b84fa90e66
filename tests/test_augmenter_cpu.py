"""CPU: the augmenter oracle (oracle/augmenter.py) against the reference-generated fixture (tests/golden/aug_small.npz,
written by oracle/gen_golden_aug.py from the real reference class) and, in the build container, against the live
reference class on fresh random cases."""
import os

import numpy as np
import pytest
import torch

from oracle import augmenter as OA
from oracle import ref_loader as RL

GOLD = os.path.join(os.path.dirname(__file__), "golden", "aug_small.npz")


def _sd(g):
    return {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd/")}


def test_oracle_matches_reference_fixture():
    g = np.load(GOLD)
    sd = _sd(g)
    x = torch.from_numpy(g["x"])
    A = int(g["dims"][4])
    s, xa = OA.forward_eval(sd, x.expand(A, -1, -1), torch.from_numpy(g["b/z0"]), torch.from_numpy(g["b/eps"]), float(g["scale"]))
    assert torch.allclose(s, torch.from_numpy(g["b/s"]), rtol=1e-5, atol=1e-6)
    assert torch.allclose(xa, torch.from_numpy(g["b/x_aug"]), rtol=1e-5, atol=1e-6)
    s, xa = OA.forward_eval(sd, x, torch.from_numpy(g["u/z0"]), torch.from_numpy(g["u/eps"]), 1.0)
    assert torch.allclose(s, torch.from_numpy(g["u/s"]), rtol=1e-5, atol=1e-6)
    assert torch.allclose(xa, torch.from_numpy(g["u/x_aug"]), rtol=1e-5, atol=1e-6)
    assert float(xa.abs().max()) > 0.1 and float((xa > 0).float().mean()) > 0.05     # not a dead network


@pytest.mark.skipif(not RL.reference_available(), reason="/root/reference not present")
@pytest.mark.parametrize("cfg", [(5, 2, 40, 20, 2, 9), (16, 10, 100, 60, 4, 33)])
@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-12), (torch.float32, 1e-5)])
def test_oracle_equals_live_reference(cfg, dtype, tol):
    NZ, Z, D, ND, A, B = cfg
    cls = RL.load_reference_augmenter()
    old = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        sd = OA.random_state_dict(NZ, Z, D, ND, seed=3, dtype=dtype)
        m = cls(noise_dim=NZ, latent_dim=Z, input_dim=D, n_dim=ND)
        m.load_state_dict(sd)
        m.eval()
        x = torch.randn(B, D, dtype=dtype).abs()
        torch.manual_seed(1)
        z0, eps = torch.randn(A, B, NZ, dtype=dtype), torch.randn(A, B, Z, dtype=dtype)
        torch.manual_seed(1)
        with torch.no_grad():
            s_ref, x_ref = m(x.expand(A, -1, -1), True, 0.1)
        s, xa = OA.forward_eval(sd, x.expand(A, -1, -1), z0, eps, 0.1)
        assert float((s - s_ref).abs().max()) <= tol * max(1.0, float(s_ref.abs().max()))
        assert float((xa - x_ref).abs().max()) <= tol * max(1.0, float(x_ref.abs().max()))
    finally:
        torch.set_default_dtype(old)


def test_mirror_constructor_matches_reference_keys_and_init():
    """Same sub-module names / shapes as the reference class, and (in the build container) bit-identical initial
    parameters for the same torch seed."""
    import distributed_vae_amd  # noqa: F401
    from distributed_vae_amd.augmentation import Augmenter_smartseq
    torch.manual_seed(11)
    m = Augmenter_smartseq(noise_dim=6, latent_dim=3, input_dim=52, n_dim=20)
    sd = OA.random_state_dict(6, 3, 52, 20)
    assert list(m.state_dict().keys()) == list(sd.keys())
    for k, v in m.state_dict().items():
        assert v.shape == sd[k].shape, k
    if RL.reference_available():
        torch.manual_seed(11)
        r = RL.load_reference_augmenter()(noise_dim=6, latent_dim=3, input_dim=52, n_dim=20)
        for (k, v), (k2, v2) in zip(m.state_dict().items(), r.state_dict().items()):
            assert k == k2 and torch.equal(v, v2), k


def test_forward_without_gpu_fails_loudly():
    import distributed_vae_amd  # noqa: F401
    from distributed_vae_amd import _native as N
    from distributed_vae_amd.augmentation import Augmenter_smartseq
    m = Augmenter_smartseq(noise_dim=6, latent_dim=3, input_dim=52, n_dim=20).eval()
    with pytest.raises(N.NativeError):
        m(torch.zeros(2, 4, 52), True, 0.1)
    m.train()
    with pytest.raises(NotImplementedError):
        m(torch.zeros(2, 4, 52), True, 0.1)
