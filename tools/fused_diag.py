"""Which results differ between the one-launch encoder chains and the per-layer launches (diagnostic)."""
import sys
import torch
sys.path.insert(0, ".")
from oracle import restatement as R
from tests import test_gpu_fused_chain as T

shape = tuple(int(v) for v in sys.argv[1:7]) if len(sys.argv) > 6 else (2, 96, 256, 32, 6, 12)
A, B, D, H, L, C = shape
h = R.Hyper(input_dim=D, fc_dim=H, n_categories=C, state_dim=2, lowD_dim=L, n_arm=A)


def run(sw):
    from distributed_vae_amd import _native as N
    from tests import gpu_util as U
    sd = R.init_state_dict(h, 11)
    x = R.synthetic_batch(B, h.input_dim, seed=12)
    m = U.build_model(h, sd)
    m.train()
    ex = N.exec_from_env(N.gemm_mode("fp32") & 0xFF)
    ex.tune[T.TUNE_FUSED] = sw
    m._exec = ex
    m.set_explicit_noise(U.noise_to_device(R.draw_noise(h, B, seed=13)))
    m.fused_train_step(x.to("cuda:0").expand(A, -1, -1), 1.0, None, do_adam=False)
    torch.cuda.synchronize()
    out = {k: gv.detach().cpu().clone() for (k, _), gv in zip(m.named_parameters(), m._grad_views)}
    for n, w in (("g1", H), ("g2", H), ("g3", H), ("g4", H), ("g5", L), ("dz2", H), ("dz3", H), ("dz4", H), ("dz5", L), ("dz1", H)):
        out["ws/" + n] = m._engine.ws_view(n, w).cpu().clone()
    return out


ref = run(0)
for sw in (5,):
    got = run(sw)
    for k in ref:
        e = float((got[k].double() - ref[k].double()).abs().max() / (ref[k].abs().max() + 1e-30))
        if e > 0:
            print(sw, k, "%.3e" % e)
