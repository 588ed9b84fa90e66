"""Diagnostic: per-phase shader cycles of k_chain_fwd (MMVAE_ABLATE_C=8 enables the in-kernel stamps)."""
import os, sys, torch, ctypes as C
os.environ["MMVAE_ABLATE_C"] = "8"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
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
off = int(N.lib().mmvae_ws_debug_offset(C.byref(eng.dims), C.byref(eng.ex)))
dbg = eng.ws[off: off + 64].view(torch.int64)
names = ["input tile stage", "weight stage (issue..LDS)", "barriers", "GEMM", "epilogue (+stats)"]
for sid, label in [(20, "encoder layer fc3 (1 layer)"), (4, "decoder chain fc6..fc10 (5 layers)")]:
    dbg.zero_(); torch.cuda.synchronize()
    eng.debug_stage(sid, hyper, noise, m._flat, x, 0, m._flat_grad)
    torch.cuda.synchronize()
    v = dbg.cpu().numpy()
    nw = max(int(v[5]), 1); tot = v[:5].sum()
    print(label, "waves", nw)
    for n_, c_ in zip(names, v[:5]):
        print(f"  {n_:28s} {c_ / nw:10.0f} cycles/wave {100.0 * c_ / max(tot,1):5.1f}%")
    print(f"  total {tot / nw:.0f} cycles/wave")
    print("  stats prologue: shift load %.0f, batched loads+accumulate %.0f, barrier %.0f, combine %.0f" % tuple(v[8:12] / nw))
