"""GPU diagnostic: the full-size case of tests/test_gpu_fullsize.py (cfg2) through the fused step under both fp32 engines
and the CPU fp32 oracle, each against the fp64 oracle: per small tensor max / median entry error and entries > 2.5e-4."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import restatement as R  # noqa: E402
from tests import gpu_util as U  # noqa: E402
A, B, D = 2, 5000, 5000
h = R.Hyper(input_dim=D, n_arm=A)
sd = R.init_state_dict(h, 546 + A)
x = R.synthetic_batch(B, D, seed=546 + D)
noise = R.draw_noise(h, B, seed=7 + A)
_, _, g32 = R.grads_autograd({k: v.clone() for k, v in sd.items()}, [x] * A, h, noise)
sd64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
n64 = {k: [t.double() if t.is_floating_point() else t for t in v] for k, v in noise.items()}
_, _, g64 = R.grads_autograd(sd64, [x.double()] * A, h, n64)
res = {}
for eng in ("fp32_mfma", "fp32x3"):
    m = U.build_model(h, sd); m.train(); m.gemm_dtype = eng
    m.set_explicit_noise(U.noise_to_device(noise))
    m.fused_train_step(x.to(U.DEV).expand(A, -1, -1), 1.0, None, do_adam=False)
    torch.cuda.synchronize()
    res[eng] = {k: gv.detach().cpu().double() for (k, _), gv in zip(m.named_parameters(), m._grad_views)}
    del m
print(f"{'tensor':20s} | {'mfma max':>9s} {'med':>9s} {'n>2.5e-4':>8s} | {'x3 max':>9s} {'med':>9s} {'n>2.5e-4':>8s} | {'cpu32 max':>9s} {'med':>9s}")
for k in g64:
    ref = g64[k].double(); sc = float(ref.abs().max()) + 1e-30
    row = []
    for g in (res["fp32_mfma"][k], res["fp32x3"][k], g32[k].double()):
        e = ((g - ref).abs() / sc).flatten()
        row.append((float(e.max()), float(e.median()), int((e > 2.5e-4).sum())))
    if ref.numel() < 1000 or row[1][0] > 1e-4:
        print(f"{k:20s} | {row[0][0]:9.2e} {row[0][1]:9.2e} {row[0][2]:8d} | {row[1][0]:9.2e} {row[1][1]:9.2e} {row[1][2]:8d} | {row[2][0]:9.2e} {row[2][1]:9.2e}")
