"""GPU tuning aid: time the five bf16 GEMM stages (HIP events), honouring MMVAE_ABLATE_B / MMVAE_SPLIT*."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import distributed_vae_amd  # noqa
from distributed_vae_amd import _native as N
from distributed_vae_amd.nn_model import mixVAE_model
A, B, D, H, L, C, S = 2, 5000, 5000, 100, 10, 92, 2
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
x = (torch.rand(B, D, generator=g, device=dev) < 0.2).float() * torch.randn(B, D, generator=g, device=dev).abs() * 3
torch.manual_seed(546)
m = mixVAE_model(input_dim=D, fc_dim=H, n_categories=C, state_dim=S, lowD_dim=L, x_drop=0.5, s_drop=0.0, n_arm=A, lam=1, lam_pc=1, tau=0.005, beta=1.0, hard=False, variational=True, device=dev, eps=1e-8, momentum=0.01, ref_prior=False, loss_mode="MSE").to(dev)
m.train(); m.gemm_dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
eng = m._ensure(B); hyper = m._hyper(1.0, False); noise = N.make_noise(None, 99, 1)
eng.forward(hyper, noise, m._flat, m._bn_flat, None, x, 0, None, True); eng.loss(hyper); eng.backward(hyper, noise, m._flat, x, 0, m._flat_grad)
torch.cuda.synchronize()
names = {14: "fc1", 1: "fc11+gd10", 10: "fc11 alone", 11: "gd10 alone", 12: "dW1", 13: "dW11"}
out = []
for sid in (14, 1, 12, 13):
    for _ in range(3): eng.debug_stage(sid, hyper, noise, m._flat, x, 0, m._flat_grad)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): eng.debug_stage(sid, hyper, noise, m._flat, x, 0, m._flat_grad)
    e1.record(); e1.synchronize()
    out.append(f"{names[sid]} {e0.elapsed_time(e1)/20*1e3:.1f}")
print(f"{m.gemm_dtype} ABLATE_B={os.environ.get('MMVAE_ABLATE_B','0')} splits={eng.splits()}:", " | ".join(out), "us", flush=True)
