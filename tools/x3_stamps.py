"""Diagnostic: phase cycles of one wave of each group of k_x3_gemm (fc1), library built with -DX3_STAMPS (MMVAE_LIB)."""
import os, sys, torch, ctypes as C
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
m.train(); m.gemm_dtype = "fp32x3"; eng = m._ensure(B); hyper = m._hyper(1.0, False); noise = N.make_noise(None, 99, 1)
eng.forward(hyper, noise, m._flat, m._bn_flat, None, x, 0, None, True); eng.loss(hyper); eng.backward(hyper, noise, m._flat, x, 0, m._flat_grad)
torch.cuda.synchronize()
off = int(N.lib().mmvae_ws_debug_offset(C.byref(eng.dims), C.byref(eng.ex)))
dbg = eng.ws[off: off + 64].view(torch.int64)
for it in range(3):
    dbg.zero_(); torch.cuda.synchronize()
    eng.debug_stage(int(sys.argv[1]) if len(sys.argv) > 1 else 14, hyper, noise, m._flat, x, 0, m._flat_grad)
    torch.cuda.synchronize()
    v = dbg.cpu().numpy()
    if len(sys.argv) > 1 and sys.argv[1] == "1":   # fc11 stamps: waves 0 and 4 of one block
        for grp in (0, 1):
            tz, te, td, n, tot, tw, tx = [int(q) for q in v[grp * 8: grp * 8 + 7]]
            n = max(n, 1)
            print(f"wave {3 * grp}: pieces {n}  per 32-gene piece: stage 1 (z of next + epilogue) {tz / n:.0f}  DMA issue {te / n:.0f}  stage 2 (d(d10)) {td / n:.0f}  tile sync {tw / n:.0f} | total {tot} cycles")
        continue
    for grp in (0, 1):
        st, mf, bar, n, tot, ld = [int(t) for t in v[grp * 8: grp * 8 + 6]]
        n = max(n, 1)
        print(f"group {grp}: K tiles {n}  per tile: load wait {ld / n:.0f}  stage {st / n:.0f}  mfma {mf / n:.0f}  barrier wait {bar / n:.0f}  | loop total {tot} ticks (s_memtime, 100 MHz)")
