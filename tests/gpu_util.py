"""Shared helpers for the -m gpu parity tests: drive the HIP path through the Python boundary
(which calls the C ABI) on explicit noise and collect outputs + workspace intermediates."""
import torch

import distributed_vae_amd  # noqa: F401  (registers the package alias)
from distributed_vae_amd.nn_model import mixVAE_model

from oracle import restatement as R

DEV = "cuda:0"


def build_model(h: R.Hyper, sd=None, device=DEV):
    m = mixVAE_model(input_dim=h.input_dim, fc_dim=h.fc_dim, n_categories=h.n_categories, state_dim=h.state_dim,
                     lowD_dim=h.lowD_dim, x_drop=h.x_drop, s_drop=h.s_drop, n_arm=h.n_arm, lam=h.lam, lam_pc=1,
                     tau=h.tau, beta=h.beta, hard=h.hard, variational=True, device=device, eps=h.eps,
                     momentum=h.momentum, ref_prior=False, loss_mode="MSE")
    if sd is not None:
        m.load_state_dict(sd)
    return m.to(device)


def noise_to_device(noise, device=DEV):
    out = {}
    for k, v in noise.items():
        if v:
            t = torch.stack([torch.as_tensor(a) for a in v])
            t = t.to(torch.uint8) if "mask" in k else t.to(torch.float32)
            out[k] = t.contiguous().to(device)
        else:
            out[k] = None
    return out


def run_step(m, x, noise, temp=1.0, eval_flag=False, backward=True):
    """forward + loss (+ backward) through the reference-shaped API. Returns (out, loss tuple, grads)."""
    A = m.n_arm
    m.set_explicit_noise(noise_to_device(noise, x.device))
    xs = x.expand(A, -1, -1)
    out = m(xs, temp, 0.0, eval=eval_flag)
    lt = m.loss(out[0], [], [], xs, out[7], out[8], out[4], out[6], 0.0)
    grads = None
    if backward:
        m.zero_grad()
        lt[0].backward()
        grads = {k: p.grad.detach().cpu().clone() for k, p in m.named_parameters()}
    torch.cuda.synchronize()
    return out, lt, grads


def ws(m, name, width):
    return m._engine.ws_view(name, width).detach().cpu().clone()
