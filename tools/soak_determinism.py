"""Soak test: the fused train step on identical state and noise must give bit-identical gradients and losses run after
run (all reductions have a fixed order).  Run-to-run differences mean a race or an unpadded hardware hazard (this is
how a sporadic dZ11 corruption would show).  Also runs the augmenter and the eval-label path repeatedly."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import distributed_vae_amd  # noqa
from distributed_vae_amd import _native as N
from distributed_vae_amd.augmentation import Augmenter_smartseq
from distributed_vae_amd.nn_model import mixVAE_model
A, B, D, H, L, C, S = int(os.environ.get("SOAK_A", 2)), 5000, 5000, 100, 10, 92, 2
n_iter = int(os.environ.get("SOAK_N", 300))
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
x = (torch.rand(B, D, generator=g, device=dev) < 0.2).float() * torch.randn(B, D, generator=g, device=dev).abs() * 3
torch.manual_seed(546)
m = mixVAE_model(input_dim=D, fc_dim=H, n_categories=C, state_dim=S, lowD_dim=L, x_drop=0.5, s_drop=0.0, n_arm=A, lam=1, lam_pc=1, tau=0.005, beta=1.0, hard=False, variational=True, device=dev, eps=1e-8, momentum=0.01, ref_prior=False, loss_mode="MSE").to(dev)
m.train(); eng = m._ensure(B); hyper = m._hyper(1.0, False); noise = N.make_noise(None, 99, 1)
bn0, nbt0 = m._bn_flat.clone(), m._nbt.clone()
ref_g = ref_l = None
bad = 0
for it in range(n_iter):
    m._bn_flat.copy_(bn0); m._nbt.copy_(nbt0)
    buf = eng.train_step(hyper, noise, m._flat, m._bn_flat, m._nbt, x, 0, m._flat_grad, False, None, None, 1, 0.0)
    gcur, lcur = m._flat_grad.clone(), buf.clone()
    if ref_g is None:
        ref_g, ref_l = gcur, lcur
    elif not (torch.equal(gcur, ref_g) and torch.equal(lcur, ref_l)):
        bad += 1
        d = (gcur != ref_g)
        print(f"iteration {it}: {int(d.sum())} gradient elements differ, loss equal: {torch.equal(lcur, ref_l)}", flush=True)
    if it % 100 == 99:
        print(f"train step: {it + 1} iterations, {bad} mismatching", flush=True)
print("train step nondeterministic iterations:", bad, "of", n_iter)
# augmenter
net = Augmenter_smartseq(50, 10, D, 500).to(dev).eval()
z0, eps = torch.randn(A, B, 50, device=dev), torch.randn(A, B, 10, device=dev)
net.set_explicit_noise(z0, eps)
ref = None; bad_a = 0
for it in range(max(n_iter // 10, 10)):
    s, xa = net(x.expand(A, -1, -1), True, 0.1)
    if ref is None: ref = (s.clone(), xa.clone())
    elif not (torch.equal(s, ref[0]) and torch.equal(xa, ref[1])): bad_a += 1
print("augmenter nondeterministic iterations:", bad_a)
# eval labels
m.eval(); ref = None; bad_e = 0
for it in range(max(n_iter // 5, 10)):
    lab = m.eval_labels(x.expand(A, -1, -1), 1.0)
    if ref is None: ref = lab.clone()
    elif not torch.equal(lab, ref): bad_e += 1
print("eval labels nondeterministic iterations:", bad_e)
sys.exit(1 if (bad or bad_a or bad_e) else 0)
