"""GPU diagnostic: dZ1 / dZ11 / G5 of the full-size step, accumulators against partial arrays, fp32x3 engine."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import restatement as R  # noqa: E402
from tests import gpu_util as U  # noqa: E402
A, B, D = 2, 5000, 5000
h = R.Hyper(input_dim=D, n_arm=A)
seed = 546
sd = R.init_state_dict(h, seed)
x = R.synthetic_batch(B, D, seed=seed + 1)
noise = R.draw_noise(h, B, seed=seed + 2)
out = {}
for part in ("0", "1"):
    os.environ["MMVAE_BN_PARTIALS"] = part
    m = U.build_model(h, sd); m.train(); m.gemm_dtype = os.environ.get("ENG", "fp32x3")
    m.set_explicit_noise(U.noise_to_device(noise))
    m.fused_train_step(x.to(U.DEV).expand(A, -1, -1), 1.0, None, do_adam=False)
    torch.cuda.synchronize()
    e = m._engine
    out[part] = {"dz1": e.ws_view("dz1", h.fc_dim).clone(), "g5": e.ws_view("g5", h.lowD_dim).clone(),
                 "r2": e.ws_view("r2", h.fc_dim).clone(), "r4": e.ws_view("r4", h.fc_dim).clone(),
                 "x_low": e.ws_view("x_low", h.lowD_dim).clone(), "gzin": e.ws_view("gzin", h.n_categories + h.state_dim).clone(),
                 "dz11": e.ws_view("dz11", D).clone()}
    del m
for k in out["0"]:
    a0, a1 = out["0"][k].double(), out["1"][k].double()
    for arm in range(A):
        d = (a0[arm] - a1[arm]).abs()
        sc = float(a1[arm].abs().max()) + 1e-300
        print(f"{k:6s} arm {arm}: max rel diff {float(d.max())/sc:.2e}  median {float(d.median())/sc:.2e}  entries > 1e-5 of max: {int((d > 1e-5*sc).sum())} of {d.numel()}"
              f"   zero-pattern differences: {int(((a0[arm] != 0) != (a1[arm] != 0)).sum())}")
