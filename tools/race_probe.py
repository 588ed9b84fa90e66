"""GPU diagnostic: the A = 3 full-size comparison of the API path with the fused step, after a prelude that leaves
freed engines / workspaces of other shapes behind (the order-dependent failure of test_full_size_more_arms...)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import restatement as R  # noqa: E402
from tests import gpu_util as U  # noqa: E402
from distributed_vae_amd import _native as N  # noqa: E402

def prelude(kind):
    if kind == "none":
        return
    A, B, D = 2, 5000, 5000
    h = R.Hyper(input_dim=D, n_arm=A)
    sd = R.init_state_dict(h, 546)
    x = R.synthetic_batch(B, D, seed=547)
    noise = R.draw_noise(h, B, seed=548)
    for dt in (["fp32x3", "fp32_mfma"] if kind == "both" else [kind]):
        m = U.build_model(h, sd); m.train(); m.gemm_dtype = dt
        m.set_explicit_noise(U.noise_to_device(noise))
        m.fused_train_step(x.to(U.DEV).expand(A, -1, -1), 1.0, None, do_adam=False)
        torch.cuda.synchronize()
        del m

prelude(os.environ.get("PRELUDE", "both"))
A = 3
B, D = 5000, 5000
h = R.Hyper(input_dim=D, fc_dim=100, n_categories=92, state_dim=2, lowD_dim=10, n_arm=A)
torch.manual_seed(546 + A)
m = U.build_model(h, None); m.train()
x = R.synthetic_batch(B, D, seed=A).to(U.DEV)
eng = m._ensure(B)
hyper, noise = m._hyper(1.0, False), N.make_noise(None, 11, A)
bn0, nbt0 = m._bn_flat.clone(), m._nbt.clone()
outs = []
for rep in range(3):
    m._bn_flat.copy_(bn0); m._nbt.copy_(nbt0)
    eng.forward(hyper, noise, m._flat, m._bn_flat, m._nbt, x, 0, None, True)
    l_api = eng.loss(hyper).clone()
    g_api = torch.zeros_like(m._flat_grad)
    eng.backward(hyper, noise, m._flat, x, 0, g_api)
    m._bn_flat.copy_(bn0); m._nbt.copy_(nbt0)
    g_f = torch.zeros_like(m._flat_grad)
    buf = eng.train_step(hyper, noise, m._flat, m._bn_flat, m._nbt, x, 0, g_f, False, None, None, 1, 0.0).clone()
    torch.cuda.synchronize()
    outs.append((g_api.clone(), g_f.clone()))
    err = (g_f - g_api).abs()
    print(f"rep {rep}: max err {float(err.max()):.3e} of {float(g_api.abs().max()):.3e}; entries that differ {int((g_f != g_api).sum())}", flush=True)
    if float(err.max()) > 0:
        views = {k: v for (k, _), v in zip(m.named_parameters(), m._grad_views)}
        off = 0
        base = m._flat_grad.data_ptr()
        for k, v in views.items():
            o = (v.data_ptr() - base) // 4
            e = err[o:o + v.numel()]
            if float(e.max()) > 0:
                print(f"   {k:18s} max {float(e.max()):.3e}  differing {int((e > 0).sum())} of {v.numel()}")
print("api identical across reps:", all(torch.equal(outs[0][0], o[0]) for o in outs), " fused identical across reps:", all(torch.equal(outs[0][1], o[1]) for o in outs))
