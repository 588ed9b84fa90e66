"""Probe for DESIGN section 12 item 2: two single-arm train steps issued on two streams of one process against one
two-arm step (is there anything to gain from de-phasing the arms?)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import distributed_vae_amd  # noqa
from distributed_vae_amd.nn_model import mixVAE_model
from distributed_vae_amd.cpl_mixvae import FusedAdam
B, D, H, L, C, S = 5000, 5000, 100, 10, 92, 2
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
x = (torch.rand(B, D, generator=g, device=dev) < 0.2).float() * torch.randn(B, D, generator=g, device=dev).abs() * 3
def mk(A):
    torch.manual_seed(546)
    m = mixVAE_model(input_dim=D, fc_dim=H, n_categories=C, state_dim=S, lowD_dim=L, x_drop=0.5, s_drop=0.0, n_arm=A, lam=1, lam_pc=1,
                     tau=0.005, beta=1.0, hard=False, variational=True, device=dev, eps=1e-8, momentum=0.01, ref_prior=False,
                     loss_mode="MSE").to(dev)
    m.train()
    return m, FusedAdam(m, lr=1e-3)
def run(fn, n=100, w=10):
    for _ in range(w): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
m2, o2 = mk(2)
print("one two-arm step            %.1f us" % run(lambda: m2.fused_train_step(x.expand(2, -1, -1), 1.0, o2, do_adam=True)))
ma, oa = mk(1); mb, ob = mk(1)
print("one single-arm step         %.1f us" % run(lambda: ma.fused_train_step(x.expand(1, -1, -1), 1.0, oa, do_adam=True)))
s0, s1 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
def both():
    with torch.cuda.stream(s0): ma.fused_train_step(x.expand(1, -1, -1), 1.0, oa, do_adam=True)
    with torch.cuda.stream(s1): mb.fused_train_step(x.expand(1, -1, -1), 1.0, ob, do_adam=True)
print("two single-arm steps, two streams (per pair) %.1f us" % run(both))
def seq():
    ma.fused_train_step(x.expand(1, -1, -1), 1.0, oa, do_adam=True); mb.fused_train_step(x.expand(1, -1, -1), 1.0, ob, do_adam=True)
print("two single-arm steps, one stream  (per pair) %.1f us" % run(seq))
