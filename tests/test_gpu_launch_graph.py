"""-m gpu: launch-graph variants of the fused train step that must not change a bit of its results.

  * the coupling terms as a ROLE of the decoder chain's launch (csrc/chain.hip k_chain_fwd_couple, the default from four arms
    up) against the coupling kernel on the side stream (the default below): same arithmetic (couple.hpp), bit-identical.

(The one-launch encoder chains this file also covered in round 3 -- bit-identical, no faster -- left the tree in round 4:
tools/experiments/fused_chain_one_launch.removed.patch.txt.)
Reference arithmetic: nn_model.py:558-569 (coupling terms) and its autograd.
"""
import pytest
import torch

from oracle import restatement as R

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TUNE_COUPLE = 13       # csrc/tune.h MMVAE_TUNE_COUPLE_SIDE: 1 coupling kernel on the side stream, 3 as a role of the decoder chain's launch


def _step(h, B, seed, fused_switch, steps=2, tune=None):
    """`steps` fused train steps (Adam included) on explicit noise; returns what must not depend on the chain's form."""
    from distributed_vae_amd import _native as N
    from distributed_vae_amd.cpl_mixvae import FusedAdam
    from tests import gpu_util as U
    sd = R.init_state_dict(h, seed)
    x = R.synthetic_batch(B, h.input_dim, seed=seed + 1)
    m = U.build_model(h, sd)
    m.train()
    ex = N.exec_from_env(N.gemm_mode("fp32") & 0xFF)
    for idx, val in (tune or {}).items():
        ex.tune[idx] = val
    m._exec = ex
    opt = FusedAdam(m, lr=1e-3)
    xs = x.to(DEV).expand(h.n_arm, -1, -1)
    bufs = []
    for s in range(steps):
        m.set_explicit_noise(U.noise_to_device(R.draw_noise(h, B, seed=seed + 2 + s)))
        bufs.append(m.fused_train_step(xs, 1.0, opt, do_adam=True).clone())
    torch.cuda.synchronize()
    eng = m._engine
    out = {"loss": torch.stack(bufs).cpu(), "params": m.flat_parameters().detach().cpu().clone(),
           "grads": m._flat_grad.detach().cpu().clone(), "bn": m._bn_flat.detach().cpu().clone()}
    for name, w in (("r1", h.fc_dim), ("r2", h.fc_dim), ("r3", h.fc_dim), ("r4", h.fc_dim), ("r5", h.lowD_dim),
                    ("dz1", h.fc_dim), ("g5", h.lowD_dim)):
        out[name] = eng.ws_view(name, w).cpu().clone()
    return out


def _same(a, b):
    for k in a:
        # NaN-safe bit comparison
        assert torch.equal(a[k].view(torch.int32) if a[k].dtype == torch.float32 else a[k],
                           b[k].view(torch.int32) if b[k].dtype == torch.float32 else b[k]), k
    assert bool(torch.isfinite(a["loss"]).all())


SHAPES = [
    # A, B, D, H, L, C
    (2, 96, 256, 32, 6, 12),       # two row blocks per arm, the second half empty
    (3, 1000, 520, 100, 10, 92),   # 16 row blocks, ragged last one (40 rows)
    (2, 5000, 1000, 100, 10, 92),  # the benchmark's chain shape: 79 row blocks per arm
]


@pytest.mark.parametrize("shape", SHAPES + [(5, 700, 256, 100, 10, 92), (4, 64, 128, 32, 6, 33)])
def test_coupling_as_a_role_of_the_decoder_launch_equals_the_side_stream_kernel(shape):
    """k_chain_fwd_couple (the fused step's coupling terms inside the decoder chain's launch, default from four arms up) against
    k_couple on the side stream: the same arithmetic (couple.hpp), exact accumulator sets -- bit-identical loss vectors,
    gradients, parameters and running statistics.  Reference: nn_model.py:558-569."""
    A, B, D, H, L, C = shape
    h = R.Hyper(input_dim=D, fc_dim=H, n_categories=C, state_dim=2, lowD_dim=L, n_arm=A)
    ref = _step(h, B, 13, 0, tune={TUNE_COUPLE: 1})
    _same(_step(h, B, 13, 0, tune={TUNE_COUPLE: 3}), ref)
    _same(_step(h, B, 13, 0), ref)      # the default choice, whichever it is for this arm count
