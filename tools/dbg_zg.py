"""Debug aid: dZ11 written by the fc11 kernels (forward with need_grad, and the fused train step) against the value
recomputed from x_rec = the forward's own reconstruction, at the full benchmark shape.  Used to find the 128-bit
buffer-store data hazard of k_fc11_zg."""
import os, sys, torch, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import distributed_vae_amd  # noqa
from distributed_vae_amd import _native as N
from distributed_vae_amd.nn_model import mixVAE_model
A, B, D, H, L, Cc, S = 2, int(os.environ.get("DBG_B", 5000)), int(os.environ.get("DBG_D", 5000)), 100, 10, 92, 2
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
x = (torch.rand(B, D, generator=g, device=dev) < 0.2).float() * torch.randn(B, D, generator=g, device=dev).abs() * 3
torch.manual_seed(546)
m = mixVAE_model(input_dim=D, fc_dim=H, n_categories=Cc, state_dim=S, lowD_dim=L, x_drop=0.5, s_drop=0.0, n_arm=A, lam=1, lam_pc=1, tau=0.005, beta=1.0, hard=False, variational=True, device=dev, eps=1e-8, momentum=0.01, ref_prior=False, loss_mode="MSE").to(dev)
m.train(); eng = m._ensure(B); hyper = m._hyper(1.0, False); noise = N.make_noise(None, 99, 1)
off = int(N.lib().mmvae_ws_offset(C.byref(eng.dims), C.byref(eng.ex), 19))   # MMVAE_WS_DZ11
print("dz11 offset", off)
def dz():
    return eng.ws[off: off + A * B * D].view(A, B, D).clone()
eng.forward(hyper, noise, m._flat, m._bn_flat, None, x, 0, None, True); torch.cuda.synchronize()
ref = dz()
eng.ws[off: off + A * B * D].zero_()
eng.train_step(hyper, noise, m._flat, m._bn_flat, m._nbt, x, 0, m._flat_grad, False, None, None, 1, 0.0); torch.cuda.synchronize()
got = dz()
bad = (got != ref)
print("mismatching dZ11 elements:", int(bad.sum()), "of", bad.numel())
if bad.any():
    idx = bad.nonzero()
    print("arms", idx[:, 0].unique().tolist()[:4], "rows", idx[:, 1].min().item(), idx[:, 1].max().item(), "cols", idx[:, 2].min().item(), idx[:, 2].max().item())
    print("distinct rows", idx[:, 1].unique().numel(), "distinct cols", idx[:, 2].unique().numel())
    print(idx[:10].tolist())
    i = idx[0]; print(got[i[0], i[1], i[2]].item(), ref[i[0], i[1], i[2]].item())
xr = torch.empty(A, B, D, device=dev)
eng.forward(hyper, noise, m._flat, m._bn_flat, None, x, 0, xr, True); torch.cuda.synchronize()
ref2 = dz()
coef = (A - 1) / B
exp = torch.where(xr > 0, coef * (xr - x.unsqueeze(0)), torch.zeros_like(xr))
print("forward(x_rec) vs expected max abs", float((ref2 - exp).abs().max()), " forward(no x_rec) vs expected", float((ref - exp).abs().max()), " train_step vs expected", float((got - exp).abs().max()))
print("count forward(no x_rec) != expected(>1e-6):", int(((ref - exp).abs() > 1e-6).sum()), " train_step:", int(((got - exp).abs() > 1e-6).sum()))
