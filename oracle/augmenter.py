"""TEST INFRASTRUCTURE ONLY -- CPU restatement (plain torch) of the reference augmenter's eval-mode forward.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product path
(distributed-vae_amd/augmentation.py -> csrc/augment.hip) never does.

Follows mmidas/augmentation/udagan.py:281-329 (`Augmenter_smartseq.forward`) for a module in eval mode, the state the
trainer uses it in (mmidas/cpl_mixvae.py:184, :422-423), and `reparam_trick` (mmidas/augmentation/aug_utils.py:51-65):
  z   = elu(bnz(noise(scale * randn)))                 bnz: affine BatchNorm1d, eps 1e-5, running statistics
  h   = relu(bn_i(fc_i(h))), i = 1..4                  bn_i: BatchNorm1d(affine=False, eps=1e-10), running statistics
  h   = relu(bn_5(fc_5(cat(h, z))))
  mu  = bn_mu(fc_mu(h));  sigma = sigmoid(fc_sigma(h));  s = randn_like(sigma) * sigma + mu
  h   = relu(bn_i(fc_i(.))), i = 6..10;   out = relu(fc_11(h))
The permutes of the `batched` branch only move the feature axis to where BatchNorm1d expects it; with running
statistics they do not change any value, so one code path serves both branches.  The two normal draws are explicit
arguments (z0: [.., noise_dim], eps: [.., latent_dim]) in the order the reference draws them.

Pinned against the live reference class in the build container (tests/test_augmenter_cpu.py, class compiled in memory
by oracle/ref_loader.py) and by the reference-generated fixture tests/golden/aug_small.npz (oracle/gen_golden_aug.py).
"""
from typing import Dict

import torch
import torch.nn.functional as F

EPS_BN = 1e-10      # udagan.py:230-276
EPS_BNZ = 1e-5      # nn.BatchNorm1d default (udagan.py:227)


def _bn(v, sd, name, eps, affine=False):
    y = (v - sd[name + ".running_mean"]) / torch.sqrt(sd[name + ".running_var"] + eps)
    if affine:
        y = y * sd[name + ".weight"] + sd[name + ".bias"]
    return y


def _lin(v, sd, name):
    b = sd.get(name + ".bias")
    return F.linear(v, sd[name + ".weight"], b)


def forward_eval(sd: Dict[str, torch.Tensor], x: torch.Tensor, z0: torch.Tensor, eps: torch.Tensor, scale: float = 1.0,
                 gemm_round=None):
    """x: [..., D]; z0: [..., noise_dim]; eps: [..., latent_dim] -> (s [..., latent_dim], x_aug [..., D]).

    ``gemm_round`` (None = the reference's arithmetic): a function applied to BOTH operands of the ten large Linear
    products -- fc1..fc4, the activation part of fc5, fc7..fc11 -- to state what the bf16-operand configuration of the
    HIP path computes (operands rounded to bf16, exact products, wide accumulation); biases, BatchNorm, the noise branch,
    the noise part of fc5, the heads and fc6 are untouched."""
    def lin(v, name):
        if gemm_round is None:
            return _lin(v, sd, name)
        return F.linear(gemm_round(v), gemm_round(sd[name + ".weight"]), sd.get(name + ".bias"))
    z = F.elu(_bn(_lin(scale * z0, sd, "noise"), sd, "bnz", EPS_BNZ, affine=True))
    h = x
    for i in (1, 2, 3, 4):
        h = F.relu(_bn(lin(h, f"fc{i}"), sd, f"batch_fc{i}", EPS_BN))
    if gemm_round is None:
        h = torch.cat((h, z), dim=-1)
        h5 = _lin(h, sd, "fc5")
    else:
        n = h.shape[-1]
        w5 = sd["fc5.weight"]
        h5 = F.linear(gemm_round(h), gemm_round(w5[:, :n])) + F.linear(z, w5[:, n:]) + sd["fc5.bias"]
    h = F.relu(_bn(h5, sd, "batch_fc5", EPS_BN))
    mu = _bn(_lin(h, sd, "fc_mu"), sd, "batch_fc_mu", EPS_BN)
    sigma = torch.sigmoid(_lin(h, sd, "fc_sigma"))
    s = eps * sigma + mu
    h = F.relu(_bn(_lin(s, sd, "fc6"), sd, "batch_fc6", EPS_BN))
    for i in (7, 8, 9, 10):
        h = F.relu(_bn(lin(h, f"fc{i}"), sd, f"batch_fc{i}", EPS_BN))
    return s, F.relu(lin(h, "fc11"))


def random_state_dict(noise_dim, latent_dim, input_dim, n_dim, seed=0, dtype=torch.float32):
    """A state dict with the reference's keys and shapes, non-trivial running statistics (a freshly constructed module
    has mean 0 / var 1, which would hide BatchNorm mistakes)."""
    g = torch.Generator().manual_seed(seed)
    n1, n5 = input_dim // 5, n_dim // 5
    sd = {}

    def lin(name, i, o, bias=True):
        sd[name + ".weight"] = (torch.rand(o, i, generator=g, dtype=dtype) * 2 - 1) / i ** 0.5
        if bias:
            sd[name + ".bias"] = (torch.rand(o, generator=g, dtype=dtype) * 2 - 1) / i ** 0.5

    def bn(name, n, affine=False):
        if affine:
            sd[name + ".weight"] = torch.rand(n, generator=g, dtype=dtype) + 0.5
            sd[name + ".bias"] = torch.randn(n, generator=g, dtype=dtype) * 0.1
        sd[name + ".running_mean"] = torch.randn(n, generator=g, dtype=dtype) * 0.1
        sd[name + ".running_var"] = torch.rand(n, generator=g, dtype=dtype) * 0.5 + 0.05
        sd[name + ".num_batches_tracked"] = torch.tensor(7)

    lin("noise", noise_dim, noise_dim, bias=False)
    bn("bnz", noise_dim, affine=True)
    for name, i, o in [("fc1", input_dim, n1), ("fc2", n1, n1), ("fc3", n1, n_dim), ("fc4", n_dim, n_dim),
                       ("fc5", n_dim + noise_dim, n5)]:
        lin(name, i, o)
        bn("batch_" + name, o)
    lin("fc_mu", n5, latent_dim)
    lin("fc_sigma", n5, latent_dim)
    bn("batch_fc_mu", latent_dim)
    for name, i, o in [("fc6", latent_dim, n5), ("fc7", n5, n_dim), ("fc8", n_dim, n_dim), ("fc9", n_dim, n1),
                       ("fc10", n1, n1)]:
        lin(name, i, o)
        bn("batch_" + name, o)
    lin("fc11", n1, input_dim)
    return sd
