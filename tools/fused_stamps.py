"""Diagnostic: per-phase shader-clock ticks of k_enc_fwd_fused (MMVAE_ABLATE_C=8 enables the in-kernel stamps), and the
launch time of the one-launch chains replayed alone (HIP events)."""
import os, sys, torch, ctypes as C
os.environ["MMVAE_ABLATE_C"] = "8"
os.environ.setdefault("MMVAE_FUSED_CHAIN", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import distributed_vae_amd  # noqa
from distributed_vae_amd import _native as N
from distributed_vae_amd.nn_model import mixVAE_model
A, B, D, H, L, Cc, S = int(os.environ.get("ARMS", 2)), 5000, 5000, 100, 10, 92, 2
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
names = ["select / barrier wait", "batch statistics read", "input planes", "weight planes -> LDS", "GEMM", "epilogue (stores, stats, atomics issued)", "drain + workgroup barrier"]
for sid, label in [(22, "k_enc_fwd_fused")]:
    dbg.zero_(); torch.cuda.synchronize()
    eng.debug_stage(sid, hyper, noise, m._flat, x, 0, m._flat_grad)
    torch.cuda.synchronize()
    v = dbg.cpu().numpy()
    nw = max(int(v[7]), 1); tot = v[:7].sum()
    print(label, "workgroups", nw)
    for n_, c_ in zip(names, v[:7]):
        print(f"  {n_:42s} {c_ / nw:10.0f} ticks/workgroup {100.0 * c_ / max(tot,1):5.1f}%")
    print(f"  total {tot / nw:.0f} ticks/workgroup")
    print(f"  (epilogue split: bias / ReLU / 16 write-through stores {v[8] / nw:.0f}, block statistics to LDS + barrier {v[9] / nw:.0f}, "
          f"combine + accumulator adds {(v[5] - 0) / nw:.0f})")
st = torch.cuda.current_stream()
for sid, label in [(22, "k_enc_fwd_fused"), (23, "k_enc_bwd_fused"), (20, "one encoder layer forward"), (21, "one encoder layer backward")]:
    for _ in range(3):
        eng.debug_stage(sid, hyper, noise, m._flat, x, 0, m._flat_grad)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(20):
        eng.debug_stage(sid, hyper, noise, m._flat, x, 0, m._flat_grad)
    e1.record(st); e1.synchronize()
    print(f"{label:30s} {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us per launch (replayed alone, stamps on)")
