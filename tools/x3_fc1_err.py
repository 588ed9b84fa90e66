"""GPU diagnostic: fc1 forward (post-ReLU, pre-BN activations r1) of both fp32 engines against fp64, per unit."""
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
keep = 1.0 / (1.0 - h.x_drop)
for eng in ("fp32_mfma", "fp32x3"):
    m = U.build_model(h, sd); m.train(); m.gemm_dtype = eng
    m.set_explicit_noise(U.noise_to_device(noise))
    m.fused_train_step(x.to(U.DEV).expand(A, -1, -1), 1.0, None, do_adam=False)
    torch.cuda.synchronize()
    r1 = m._engine.ws_view("r1", h.fc_dim).cpu().double()
    for a in range(A):
        xm = (x * noise["x_mask"][a].float()).double()
        z = keep * (xm @ sd[f"fc1.{a}.weight"].double().t()) + sd[f"fc1.{a}.bias"].double()
        want = torch.relu(z)
        e = (r1[a] - want).abs()
        col_scale = want.abs().max(0).values + 1e-30
        rel_col = (e.max(0).values / col_scale)
        print(f"{eng:10s} arm {a}: max abs err {float(e.max()):.3e}  (max |r1| {float(want.max()):.3e});  worst per-unit relative error {float(rel_col.max()):.3e} (unit {int(rel_col.argmax())}, its scale {float(col_scale[rel_col.argmax()]):.3e});  median per-unit {float(rel_col.median()):.3e}")
        big = (e > 1e-4 * col_scale).sum().item()
        print(f"           entries with error > 1e-4 of their unit's scale: {big}; z range [{float(z.min()):.2f}, {float(z.max()):.2f}]")
    del m
# ReLU decisions of layer 1 against fp64, both engines
for eng in ("fp32_mfma", "fp32x3"):
    m = U.build_model(h, sd); m.train(); m.gemm_dtype = eng
    m.set_explicit_noise(U.noise_to_device(noise))
    m.fused_train_step(x.to(U.DEV).expand(A, -1, -1), 1.0, None, do_adam=False)
    torch.cuda.synchronize()
    r1 = m._engine.ws_view("r1", h.fc_dim).cpu().double()
    for a in range(A):
        xm = (x * noise["x_mask"][a].float()).double()
        z = keep * (xm @ sd[f"fc1.{a}.weight"].double().t()) + sd[f"fc1.{a}.bias"].double()
        flips = ((r1[a] > 0) != (z > 0))
        print(f"{eng:10s} arm {a}: ReLU decisions that differ from fp64: {int(flips.sum())}; |z| there: {z[flips].abs().tolist()[:8]}; entries with |z| < 1e-5: {int((z.abs() < 1e-5).sum())}, exactly 0: {int((z == 0).sum())}")
    del m
