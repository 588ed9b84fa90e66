"""Replays single stages of the step (mmvae_debug_stage) on the benchmark workload so that rocprofv3
(kernel trace or --pmc passes) sees each kernel in isolation.
usage: python3 tools/stage_prof.py [reps] [stage ids ...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import distributed_vae_amd  # noqa: E402,F401
from distributed_vae_amd import _native as N  # noqa: E402
from distributed_vae_amd.nn_model import mixVAE_model  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
stages = [int(s) for s in sys.argv[2:]] or [0, 1, 2, 3, 4, 5, 6, 7]
A, B, D, H, L, C, S = 2, 5000, 5000, 100, 10, 92, 2
dev = torch.device("cuda", 0)
torch.manual_seed(546)
m = mixVAE_model(input_dim=D, fc_dim=H, n_categories=C, state_dim=S, lowD_dim=L, x_drop=0.5, s_drop=0.0, n_arm=A,
                 lam=1, lam_pc=1, tau=0.005, beta=1.0, hard=False, variational=True, device=dev, eps=1e-8,
                 momentum=0.01, ref_prior=False, loss_mode="MSE").to(dev)
m.train()
g = torch.Generator(device=dev).manual_seed(1)
x = (torch.rand(B, D, generator=g, device=dev) < 0.2).float() * torch.randn(B, D, generator=g, device=dev).abs() * 3
eng = m._ensure(B)
hyper = m._hyper(1.0, False)
noise = N.make_noise(None, 99, 1)
eng.forward(hyper, noise, m._flat, m._bn_flat, None, x, 0, None, True)
eng.loss(hyper)
eng.backward(hyper, noise, m._flat, x, 0, m._flat_grad)
torch.cuda.synchronize()
for sid in stages:
    for _ in range(reps):
        eng.debug_stage(sid, hyper, noise, m._flat, x, 0, m._flat_grad)
torch.cuda.synchronize()
print("done", stages, reps)
