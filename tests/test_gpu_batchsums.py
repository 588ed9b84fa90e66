"""-m gpu: the BatchNorm batch sums (csrc/common.hpp acc_add / acc_get: per-workgroup block sums added to fixed-point
accumulators with integer atomics, three 64-bit slots per sum -- exact, hence independent of the arrival order).

Pinned here: the batch statistics against an fp64 evaluation of the activations the device stored; bit-identical results
from run to run; agreement with round 1's scheme (per-workgroup partial arrays recombined by every consumer,
``MMVAE_BN_PARTIALS=1``) to fp32 rounding; and what happens outside the accumulators' window -- a non-finite or
out-of-range block sum turns the statistics, and with them the loss, into NaN instead of wrapping silently.
Reference arithmetic: nn.BatchNorm1d in training mode (mmidas/nn_model.py:208-255, :263-271) and its autograd.
"""
import os

import pytest
import torch

from oracle import restatement as R

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _step(h, B, seed, partials, x=None, dtype="fp32", edit=None):
    from tests import gpu_util as U
    old = os.environ.get("MMVAE_BN_PARTIALS")
    os.environ["MMVAE_BN_PARTIALS"] = "1" if partials else "0"
    try:
        sd = R.init_state_dict(h, seed)
        if edit is not None:
            edit(sd)
        if x is None:
            x = R.synthetic_batch(B, h.input_dim, seed=seed + 1)
        noise = R.draw_noise(h, B, seed=seed + 2)
        m = U.build_model(h, sd)
        m.train()
        m.gemm_dtype = dtype
        m.set_explicit_noise(U.noise_to_device(noise))
        buf = m.fused_train_step(x.to(DEV).expand(h.n_arm, -1, -1), 1.0, None, do_adam=False).clone()
        torch.cuda.synchronize()
        grads = {k: gv.detach().cpu().clone() for (k, _), gv in zip(m.named_parameters(), m._grad_views)}
        return m, buf.cpu(), grads
    finally:
        if old is None:
            os.environ.pop("MMVAE_BN_PARTIALS", None)
        else:
            os.environ["MMVAE_BN_PARTIALS"] = old


@pytest.mark.parametrize("partials", [False, True])
@pytest.mark.parametrize("shape", [(2, 300, 520, 100), (3, 1100, 640, 64), (2, 5000, 1000, 100)])
def test_batch_statistics_against_fp64_of_the_stored_activations(shape, partials):
    A, B, D, H = shape
    h = R.Hyper(input_dim=D, fc_dim=H, n_categories=12, state_dim=2, lowD_dim=6, n_arm=A)
    m, _, _ = _step(h, B, 31, partials)
    e = m._engine
    r1 = e.ws_view("r1", H).cpu().double()
    pad = -(-A * H // 64) * 64
    raw = e.ws_raw("bn_mean1", pad + A * H).cpu().double()
    mean, rstd = raw[:A * H].view(A, H), raw[pad:pad + A * H].view(A, H)
    want_mean = r1.mean(1)
    want_rstd = 1.0 / torch.sqrt(r1.var(1, unbiased=False) + h.eps)
    assert float((mean - want_mean).abs().max()) <= 2e-7 * float(want_mean.abs().max())
    live = r1.var(1, unbiased=False) > 1e-12                 # a dead unit's rstd is 1 / sqrt(eps): compared absolutely below
    assert float(((rstd - want_rstd).abs() / want_rstd)[live].max()) <= 1e-6
    assert float(((rstd - want_rstd).abs() / want_rstd).max()) <= 1e-4


def test_accumulators_are_bit_reproducible_and_agree_with_the_partial_scheme():
    A, B, D, H = 2, 1100, 2600, 100
    h = R.Hyper(input_dim=D, fc_dim=H, n_categories=12, state_dim=2, lowD_dim=6, n_arm=A)
    _, b0, g0 = _step(h, B, 57, False)
    _, b1, g1 = _step(h, B, 57, False)
    assert torch.equal(b0, b1)
    for k in g0:
        assert torch.equal(g0[k], g1[k]), k
    _, bp, gp = _step(h, B, 57, True)
    assert float(((b0 - bp).abs() / (bp.abs() + 1e-30)).max()) < 1e-5
    for k in g0:
        sc = float(gp[k].abs().max()) + 1e-30
        e = ((g0[k] - gp[k]).abs() / sc).flatten()
        # typical entry at fp32 rounding (bias gradients below a BatchNorm are sums that cancel to ~1e-2 of their terms:
        # ten times the relative noise); the worst entry allows one flipped ReLU decision between the two roundings
        assert float(e.median()) < (5e-5 if e.numel() < 1000 else 2e-6), (k, float(e.median()))
        assert float(e.max()) < 5e-3, (k, float(e.max()))


@pytest.mark.parametrize("bad", ["inf", "window"])
def test_sums_outside_the_window_poison_the_statistics(bad):
    """A block sum that is not finite (here: one hidden unit with an infinite bias, so its activations are +inf), or one
    that leaves the accumulators' window (2^58), must turn the statistics and the loss into NaN -- nn.BatchNorm1d on inf
    gives nan as well; beyond the window this build stops earlier than fp32 would but never returns
    statistics from wrapped integers."""
    A, B, D, H = 2, 300, 520, 100
    h = R.Hyper(input_dim=D, fc_dim=H, n_categories=12, state_dim=2, lowD_dim=6, n_arm=A)
    x = R.synthetic_batch(B, D, seed=5)
    edit = None
    if bad == "inf":
        def edit(sd):
            sd["fc1.0.bias"][3] = float("inf")
    else:
        x = x * 3e9 + 3e9          # fc1 outputs of ~1e9 and more: 64-cell block sums of squares beyond 2.9e17
    _, buf, _ = _step(h, B, 11, False, x=x, edit=edit)
    assert bool(torch.isnan(buf[0])), buf[:5]
    # and the next, ordinary step is clean (the sets are zeroed at the start of every pass)
    _, buf2, _ = _step(h, B, 11, False)
    assert bool(torch.isfinite(buf2).all())


def _step_env(h, B, seed, env, dtype="fp32"):
    from tests import gpu_util as U
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        sd = R.init_state_dict(h, seed)
        x = R.synthetic_batch(B, h.input_dim, seed=seed + 1)
        noise = R.draw_noise(h, B, seed=seed + 2)
        m = U.build_model(h, sd)
        m.train()
        m.gemm_dtype = dtype
        m.set_explicit_noise(U.noise_to_device(noise))
        buf = m.fused_train_step(x.to(DEV).expand(h.n_arm, -1, -1), 1.0, None, do_adam=False).clone()
        torch.cuda.synchronize()
        grads = {k: gv.detach().cpu().clone() for (k, _), gv in zip(m.named_parameters(), m._grad_views)}
        return sd, x, noise, buf.cpu(), grads
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_chain_products_on_the_split_engine_agree_with_the_fp32_matrix_instruction(dtype):
    """csrc/chain.hip k_chain_fwd<true> / k_chain_bwd<true> (operands as three bf16 slice planes, six slice products per
    product) against the same kernels on v_mfma_f32_32x32x2_f32 (``MMVAE_CHAIN_FP32=1``): the fused step's loss vector and
    gradients agree to fp32 rounding -- typical entry; the worst entry allows one flipped ReLU decision -- in the fp32
    configuration and in the bf16 one (whose chain kernels take the same form)."""
    A, B, D, H = 2, 1100, 2600, 100
    h = R.Hyper(input_dim=D, fc_dim=H, n_categories=12, state_dim=2, lowD_dim=6, n_arm=A)
    _, _, _, b3, g3 = _step_env(h, B, 91, {"MMVAE_CHAIN_FP32": "0"}, dtype)
    _, _, _, b0, g0 = _step_env(h, B, 91, {"MMVAE_CHAIN_FP32": "1"}, dtype)
    assert float(((b3 - b0).abs() / (b0.abs() + 1e-30)).max()) < 1e-5
    for k in g3:
        sc = float(g0[k].abs().max()) + 1e-30
        e = ((g3[k] - g0[k]).abs() / sc).flatten()
        assert float(e.median()) < (5e-5 if e.numel() < 1000 else 2e-6), (k, float(e.median()))
        assert float(e.max()) < 5e-3, (k, float(e.max()))


@pytest.mark.parametrize("cs", [(125, 4), (92, 2), (13, 1)])
def test_chain_forms_by_width_against_the_oracle(cs):
    """Decoder input width C + S = 129 keeps the chain kernels on the fp32 form (one 128 x 128 plane per weight does not hold
    it), 94 and 14 (not multiples of four: element-wise loads of the input tile) take the split form: forward outputs, loss
    and gradients against the oracle at the fp32 gates of tests/test_gpu_parity.py."""
    from tests import golden_util as G
    C, S = cs
    A, B, D, H = 2, 150, 260, 100
    h = R.Hyper(input_dim=D, fc_dim=H, n_categories=C, state_dim=S, lowD_dim=10, n_arm=A)
    sd, x, noise, buf, grads = _step_env(h, B, 17, {})
    _, lt, gref = R.grads_autograd(sd, [x] * A, h, noise)
    lt = [v.detach() if torch.is_tensor(v) else v for v in lt]
    want = [float(lt[0]), float(lt[2]), float(lt[3]), float(lt[4]), float(lt[5])] + [float(v) for v in lt[1]]
    for got, w in zip(buf[:5 + A].tolist(), want):
        assert abs(got - w) <= 1e-5 * abs(w) + 1e-7, (got, w)
    for k, v in grads.items():
        assert G.rel_err(v, gref[k]) < 1e-3, k


@pytest.mark.parametrize("fill", ["nan", "big"])
def test_results_do_not_depend_on_what_the_workspace_held(fill):
    """The caller's workspace is uninitialised memory: every pass zeroes what it accumulates into and writes
    every padded region it later reads.  Pre-filling the workspace with NaN or with 3e38 must give bit-identical loss vectors
    and gradients, on the API path (forward, loss, backward as three calls) and in the fused step."""
    from tests import gpu_util as U
    from distributed_vae_amd import _native as N
    A, B, D = 2, 1100, 2600
    h = R.Hyper(input_dim=D, fc_dim=100, n_categories=92, state_dim=2, lowD_dim=10, n_arm=A)
    torch.manual_seed(5)
    m = U.build_model(h, None)
    m.train()
    x = R.synthetic_batch(B, D, seed=3).to(DEV)
    eng = m._ensure(B)
    hyper, noise = m._hyper(1.0, False), N.make_noise(None, 11, A)
    bn0, nbt0 = m._bn_flat.clone(), m._nbt.clone()

    def run(path, value):
        eng.ws.fill_(value)
        m._bn_flat.copy_(bn0)
        m._nbt.copy_(nbt0)
        g = torch.zeros_like(m._flat_grad)
        if path == "api":
            eng.forward(hyper, noise, m._flat, m._bn_flat, m._nbt, x, 0, None, True)
            lv = eng.loss(hyper).clone()
            eng.backward(hyper, noise, m._flat, x, 0, g)
        else:
            lv = eng.train_step(hyper, noise, m._flat, m._bn_flat, m._nbt, x, 0, g, False, None, None, 1, 0.0).clone()
        torch.cuda.synchronize()
        return lv, g

    for path in ("api", "fused"):
        l0, g0 = run(path, 0.0)
        l1, g1 = run(path, float("nan") if fill == "nan" else 3.0e38)
        assert bool(torch.isfinite(g1).all()) and bool(torch.isfinite(l1).all()), path
        assert torch.equal(l0, l1), path
        assert torch.equal(g0, g1), path
