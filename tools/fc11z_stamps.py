"""Diagnostic: per-phase shader cycles of k_fc11_z (MMVAE_ABLATE_Z=8 enables the in-kernel stamps)."""
import os, sys, torch
os.environ["MMVAE_ABLATE_Z"] = "8"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C
import distributed_vae_amd  # noqa
from distributed_vae_amd import _native as N
from distributed_vae_amd.nn_model import mixVAE_model
A, B, D, H, L, Cc, S = 2, 5000, 5000, 100, 10, 92, 2
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
x = (torch.rand(B, D, generator=g, device=dev) < 0.2).float() * torch.randn(B, D, generator=g, device=dev).abs() * 3
torch.manual_seed(546)
m = mixVAE_model(input_dim=D, fc_dim=H, n_categories=Cc, state_dim=S, lowD_dim=L, x_drop=0.5, s_drop=0.0, n_arm=A, lam=1, lam_pc=1, tau=0.005, beta=1.0, hard=False, variational=True, device=dev, eps=1e-8, momentum=0.01, ref_prior=False, loss_mode="MSE").to(dev)
m.train(); eng = m._ensure(B); hyper = m._hyper(1.0, False); noise = N.make_noise(None, 99, 1)
eng.forward(hyper, noise, m._flat, m._bn_flat, None, x, 0, None, True); eng.loss(hyper); eng.backward(hyper, noise, m._flat, x, 0, m._flat_grad)
torch.cuda.synchronize()
eng.ws.zero_()
eng.forward(hyper, noise, m._flat, m._bn_flat, None, x, 0, None, True)
torch.cuda.synchronize()
# locate fc11_part: not exported -> scan for the counter block = nonzero u64 sextet; simpler: recompute offset via the C layout
# (fc11_part follows fc1_slab in make_layout; expose through ws_offset id 19 'dz11' minus GD10 slab is brittle) -> brute force:
w = eng.ws.view(torch.int64)
nz = torch.nonzero(w[: w.numel()] > (1 << 20)).flatten()
cand = [int(i) for i in nz.tolist()[:0]]
import numpy as np
wn = w.cpu().numpy()
# counters: 5 large values followed by the wave count (= 4 * blocks)
for i in range(len(wn) - 6):
    if 0 < wn[i + 5] < 100000 and all(wn[i + k] > 1000000 for k in range(5)) and wn[i+5] % 4 == 0:
        vals = wn[i:i + 6]
        nw = vals[5]
        names = ["x-load issue", "z GEMM (104 MFMA)", "W prefetch issue", "epilogue", "W->LDS + barrier"]
        tot = vals[:5].sum()
        print("waves", nw)
        for n_, v in zip(names, vals[:5]):
            print(f"  {n_:22s} {v / nw:12.0f} cycles/wave  {100.0 * v / tot:5.1f}%")
        print(f"  total {tot / nw:.0f} cycles/wave = {tot / nw / 2.35e3:.1f} us at 2.35 GHz")
        break
