"""GPU diagnostic: do results depend on what the workspace held before the call?  Runs the API path (forward, loss,
backward) and the fused step on a workspace pre-filled with zeros, with NaN and with large finite garbage."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import restatement as R  # noqa: E402
from tests import gpu_util as U  # noqa: E402
from distributed_vae_amd import _native as N  # noqa: E402
A = int(os.environ.get("ARMS", "3"))
B, D = int(os.environ.get("BATCH", "5000")), 5000
h = R.Hyper(input_dim=D, fc_dim=100, n_categories=92, state_dim=2, lowD_dim=10, n_arm=A)
torch.manual_seed(546 + A)
m = U.build_model(h, None); m.train()
x = R.synthetic_batch(B, D, seed=A).to(U.DEV)
eng = m._ensure(B)
hyper, noise = m._hyper(1.0, False), N.make_noise(None, 11, A)
bn0, nbt0 = m._bn_flat.clone(), m._nbt.clone()
res = {}
for fill in ("zero", "nan", "big", "zero2"):
    for path in ("api", "fused"):
        if fill.startswith("zero"): eng.ws.zero_()
        elif fill == "nan": eng.ws.fill_(float("nan"))
        else: eng.ws.fill_(3.0e38)
        m._bn_flat.copy_(bn0); m._nbt.copy_(nbt0)
        g = torch.zeros_like(m._flat_grad)
        if path == "api":
            eng.forward(hyper, noise, m._flat, m._bn_flat, m._nbt, x, 0, None, True)
            l = eng.loss(hyper).clone()
            eng.backward(hyper, noise, m._flat, x, 0, g)
        else:
            l = eng.train_step(hyper, noise, m._flat, m._bn_flat, m._nbt, x, 0, g, False, None, None, 1, 0.0).clone()
        torch.cuda.synchronize()
        res[fill, path] = (l.cpu(), g.cpu())
for path in ("api", "fused"):
    l0, g0 = res["zero", path]
    for fill in ("nan", "big", "zero2"):
        l1, g1 = res[fill, path]
        print(f"{path:6s} {fill:5s}: loss identical {torch.equal(l0, l1)}  grads identical {torch.equal(g0, g1)}  finite {bool(torch.isfinite(g1).all())}"
              f"  max |dg| / max |g| {float((g1 - g0).abs().max() / g0.abs().max()):.2e}  entries that differ {int((g1 != g0).sum())}")
