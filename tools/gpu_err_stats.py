"""GPU diagnostic: full-size gradient error statistics of the HIP path and of the CPU fp32 oracle,
both against an fp64 evaluation of the same step (max and 90th-percentile entry error per tensor)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import restatement as R  # noqa: E402
from tests import gpu_util as U  # noqa: E402

h = R.Hyper()
B = 5000
A = h.n_arm
sd = R.init_state_dict(h, 546)
x = R.synthetic_batch(B, h.input_dim)
noise = R.draw_noise(h, B, seed=7)
_, _, g32 = R.grads_autograd({k: v.clone() for k, v in sd.items()}, [x] * A, h, noise)
sd64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
n64 = {k: [t.double() if t.is_floating_point() else t for t in v] for k, v in noise.items()}
_, _, g64 = R.grads_autograd(sd64, [x.double()] * A, h, n64)
m = U.build_model(h, sd)
m.train()
_, _, gg = U.run_step(m, x.to(U.DEV), noise)
print(f"{'tensor':22s} {'gpu max':>10s} {'gpu p90':>10s} {'gpu n>1e-4':>10s} | {'cpu max':>10s} {'cpu p90':>10s} {'cpu n>1e-4':>10s}")
for k in R.param_keys(h):
    ref = g64[k]
    sc = float(ref.abs().max()) + 1e-30
    eg = ((gg[k].double() - ref).abs() / sc).flatten()
    ec = ((g32[k].double() - ref).abs() / sc).flatten()
    q = lambda e: float(e.kthvalue(max(1, int(0.9 * e.numel()))).values)
    print(f"{k:22s} {float(eg.max()):10.2e} {q(eg):10.2e} {int((eg > 1e-4).sum()):10d} | {float(ec.max()):10.2e} {q(ec):10.2e} {int((ec > 1e-4).sum()):10d}")
