"""CPU, build container only: oracle/restatement.py against the live reference model.

Skipped wherever /root/reference is absent (the GPU box).  Imports the reference's own
mmidas/nn_model.py (forward :297, loss :495) through oracle/ref_loader.py and checks the
restatement in fp64 (where the two must agree to rounding) and fp32 on fresh random cases, i.e.
cases that are not in the committed fixtures.
"""
import warnings

import pytest
import torch

from oracle import ref_loader as RL
from oracle import restatement as R

pytestmark = pytest.mark.skipif(not RL.reference_available(), reason="/root/reference not present")

CASES = [
    # A, B, D, H, L, C, S, hard, s_drop
    (2, 24, 40, 12, 4, 6, 2, False, 0.0),
    (3, 33, 50, 10, 3, 8, 2, False, 0.3),
    (4, 20, 36, 8, 4, 5, 1, True, 0.0),
]


def _mk(ref, cfg, dtype):
    A, B, D, H, L, C, S, hard, sdrop = cfg
    torch.manual_seed(99)
    m = ref.mixVAE_model(input_dim=D, fc_dim=H, n_categories=C, state_dim=S, lowD_dim=L, x_drop=0.5,
                         s_drop=sdrop, n_arm=A, lam=1, lam_pc=1, tau=0.005, beta=1.0, hard=hard,
                         variational=True, device="cpu", eps=1e-8, momentum=0.01, ref_prior=False,
                         loss_mode="MSE")
    h = R.Hyper(input_dim=D, fc_dim=H, n_categories=C, state_dim=S, lowD_dim=L, x_drop=0.5,
                s_drop=sdrop, n_arm=A, hard=hard)
    return m, h


@pytest.mark.parametrize("cfg", CASES)
@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-11), (torch.float32, 2e-4)])
def test_restatement_equals_reference(cfg, dtype, tol):
    warnings.simplefilter("ignore")
    ref = RL.load_reference_nn_model()
    old = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        m, h = _mk(ref, cfg, dtype)
        A, B, D = cfg[:3]
        sd = R.init_state_dict(h, 99, dtype=dtype)
        for k, v in m.state_dict().items():
            assert torch.equal(v, sd[k]), k          # same constructor RNG order
        x = R.synthetic_batch(B, D, seed=5, dtype=dtype)
        noise = R.draw_noise(h, B, seed=11)
        m.train()
        out, lo, _ = RL.reference_step(m, x.expand(A, -1, -1), 1.0, noise)
        lo[0].backward()
        out2, lt, grads = R.grads_autograd(sd, [x] * A, h, noise)
        _, _, gman, _ = R.grads_manual(R.init_state_dict(h, 99, dtype=dtype), [x] * A, h, noise)
        rel = lambda a, b: float((a - b).abs().max() / (b.abs().max() + 1e-30))
        assert abs(float(lt[0]) - float(lo[0])) <= tol * abs(float(lo[0]))
        for i in (0, 3, 4, 5, 6, 7, 8, 9):
            for a in range(A):
                assert rel(out2[i][a], out[i][a]) < max(tol, 1e-12), (i, a)
        for k, p in m.named_parameters():
            assert rel(grads[k], p.grad) < tol, k
            assert rel(gman[k], p.grad) < tol, k
        for k, v in m.state_dict().items():
            if "running" in k:
                assert rel(sd[k], v) < max(tol, 1e-12), k
    finally:
        torch.set_default_dtype(old)


def test_two_reference_snapshots_agree():
    """build/lib snapshot has identical arithmetic to the current file."""
    warnings.simplefilter("ignore")
    ref, old = RL.load_reference_nn_model(), RL.load_reference_nn_model_old()
    cfg = CASES[0]
    A, B, D = cfg[:3]
    res = []
    for mod in (ref, old):
        m, h = _mk(mod, cfg, torch.float32)
        x = R.synthetic_batch(B, D, seed=5)
        m.train()
        out, lo, _ = RL.reference_step(m, x.expand(A, -1, -1), 1.0, R.draw_noise(h, B, seed=11))
        res.append(float(lo[0]))
    assert res[0] == res[1]


def test_checkpoints_round_trip_with_the_reference(tmp_path):
    """a checkpoint written by this package's trainer (cpl_mixvae.py:783-786 layout) loads into
    the reference model and torch.optim.Adam with strict key checking, and a checkpoint written the reference's way
    loads back here -- so the reference's evaluation scripts keep working on these files."""
    warnings.simplefilter("ignore")
    import distributed_vae_amd  # noqa: F401
    from distributed_vae_amd.cpl_mixvae import FusedAdam, cpl_mixVAE
    ref = RL.load_reference_nn_model()
    cfg = (3, 8, 40, 12, 4, 6, 2, False, 0.0)
    A, B, D, H, L, C, S = cfg[:7]
    t = cpl_mixVAE(saving_folder=str(tmp_path), device="cpu", save_flag=True)
    t.init_model(n_categories=C, state_dim=S, input_dim=D, fc_dim=H, lowD_dim=L, x_drop=0.5, s_drop=0.0, n_arm=A)
    t.optimizer._bind()
    t.optimizer.step_count = 3
    t.optimizer.exp_avg.uniform_(-1, 1)
    t.optimizer.exp_avg_sq.uniform_(0, 1)
    path = str(tmp_path / "ckpt.pth")
    t.save_checkpoint(path)
    loaded = torch.load(path, map_location="cpu", weights_only=True)
    assert set(loaded) == {"model_state_dict", "optimizer_state_dict"}
    m_ref, _ = _mk(ref, cfg, torch.float32)
    missing = m_ref.load_state_dict(loaded["model_state_dict"], strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    for k, v in m_ref.state_dict().items():
        assert torch.equal(v, t.model.state_dict()[k]), k
    opt_ref = torch.optim.Adam(m_ref.parameters(), lr=1e-3)
    opt_ref.load_state_dict(loaded["optimizer_state_dict"])
    assert int(opt_ref.state[next(iter(m_ref.parameters()))]["step"]) == 3
    # the other direction: the reference's save (cpl_mixvae.py:783-786) -> this trainer's load_model
    ref_path = str(tmp_path / "ref.pth")
    torch.save({"model_state_dict": m_ref.state_dict(), "optimizer_state_dict": opt_ref.state_dict()}, ref_path)
    t2 = cpl_mixVAE(saving_folder=str(tmp_path), device="cpu", save_flag=False)
    t2.init_model(n_categories=C, state_dim=S, input_dim=D, fc_dim=H, lowD_dim=L, x_drop=0.5, s_drop=0.0, n_arm=A,
                  trained_model=ref_path)
    for k, v in t2.model.state_dict().items():
        assert torch.equal(v, m_ref.state_dict()[k]), k
    assert t2.optimizer.step_count == 3
    t3 = cpl_mixVAE(saving_folder=str(tmp_path), device="cpu", save_flag=False)
    t3.init_model(n_categories=C, state_dim=S, input_dim=D, fc_dim=H, lowD_dim=L, x_drop=0.5, s_drop=0.0, n_arm=A)
    t3.load_model(ref_path)
    assert torch.equal(t3.model.state_dict()["fc1.0.weight"], m_ref.state_dict()["fc1.0.weight"])


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-11), (torch.float32, 2e-4)])
def test_masked_forward_equals_reference(dtype, tol):
    """forward(mask=kept categories) -- the pruning-time forward, nn_model.py:332-335 -- train mode (with loss and
    gradients) and eval mode, restatement against the live reference."""
    warnings.simplefilter("ignore")
    ref = RL.load_reference_nn_model()
    old = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        cfg = (3, 33, 50, 10, 3, 8, 2, False, 0.0)
        m, h = _mk(ref, cfg, dtype)
        A, B, D = cfg[:3]
        mask = [1, 2, 4, 7]
        sd = R.init_state_dict(h, 99, dtype=dtype)
        x = R.synthetic_batch(B, D, seed=5, dtype=dtype)
        xs = x.expand(A, -1, -1)
        noise = R.draw_noise(h, B, seed=11)
        rel = lambda a, b: float((a - b).abs().max() / (b.abs().max() + 1e-30))
        m.train()
        with RL.explicit_noise(m, noise):
            out = m(xs, 1.0, 0.0, eval=False, mask=mask)
        lo = m.loss(out[0], [], [], xs, out[7], out[8], out[4], out[6], 0.0)
        lo[0].backward()
        out2, lt, grads = R.grads_autograd(sd, [x] * A, h, noise, mask=mask)
        _, _, gman, _ = R.grads_manual(R.init_state_dict(h, 99, dtype=dtype), [x] * A, h, noise, mask=mask)
        assert abs(float(lt[0]) - float(lo[0])) <= tol * abs(float(lo[0]))
        for i in (0, 3, 4, 5, 6, 7, 8, 9):
            for a in range(A):
                assert rel(out2[i][a], out[i][a]) < max(tol, 1e-12), (i, a)
        for a in range(A):
            assert float(out2[4][a][:, [0, 3, 5, 6]].abs().max()) == 0.0        # masked-out categories: exactly zero
        for k, p in m.named_parameters():
            assert rel(grads[k], p.grad) < tol, k
            assert rel(gman[k], p.grad) < tol, k
        m.eval()
        ne = R.draw_noise(h, B, seed=12, training=False, eval_flag=True)
        with torch.no_grad(), RL.explicit_noise(m, ne):
            oe = m(xs, 1.0, 0.0, eval=True, mask=mask)
        oe2 = R.forward(sd, [x] * A, h, ne, training=False, eval_flag=True, mask=mask)
        for i in (0, 3, 4, 5, 6, 7, 8, 9):
            for a in range(A):
                assert rel(oe2[i][a], oe[i][a]) < max(tol, 1e-12), (i, a)
    finally:
        torch.set_default_dtype(old)
