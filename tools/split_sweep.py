"""GPU tuning aid: time the big stages (mmvae_debug_stage, HIP events) for several split factors."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import distributed_vae_amd  # noqa: E402,F401
from distributed_vae_amd import _native as N  # noqa: E402
from distributed_vae_amd.nn_model import mixVAE_model  # noqa: E402

A, B, D, H, L, C, S = 2, 5000, 5000, 100, 10, 92, 2
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
x = (torch.rand(B, D, generator=g, device=dev) < 0.2).float() * torch.randn(B, D, generator=g, device=dev).abs() * 3


def timeit(eng, sid, hyper, noise, m, reps=10):
    for _ in range(2):
        eng.debug_stage(sid, hyper, noise, m._flat, x, 0, m._flat_grad)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        eng.debug_stage(sid, hyper, noise, m._flat, x, 0, m._flat_grad)
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


which = {0: 0, 1: 1, 2: 2, 3: 3}
for stage, split_id in which.items():
    for ks in [int(v) for v in (sys.argv[1:] or ["2", "3", "4", "5", "6", "8", "10", "12", "16"])]:
        ex = N.exec_from_env()
        for w in range(4):
            ex.split[w] = 0
        ex.split[split_id] = ks
        torch.manual_seed(546)
        m = mixVAE_model(input_dim=D, fc_dim=H, n_categories=C, state_dim=S, lowD_dim=L, x_drop=0.5, s_drop=0.0,
                         n_arm=A, lam=1, lam_pc=1, tau=0.005, beta=1.0, hard=False, variational=True, device=dev,
                         eps=1e-8, momentum=0.01, ref_prior=False, loss_mode="MSE").to(dev)
        m.train()
        m._exec = ex
        eng = m._ensure(B)
        hyper = m._hyper(1.0, False)
        noise = N.make_noise(None, 99, 1)
        eng.forward(hyper, noise, m._flat, m._bn_flat, None, x, 0, None, True)
        eng.loss(hyper)
        eng.backward(hyper, noise, m._flat, x, 0, m._flat_grad)
        torch.cuda.synchronize()
        print(f"stage {stage} split {ks:3d}: {timeit(eng, stage, hyper, noise, m):8.1f} us", flush=True)
        del m, eng
