"""CPU: the oracle (oracle/restatement.py) against the reference-generated golden vectors.

The fixtures in tests/golden were produced by the *real* reference model
(/root/reference/mmidas/nn_model.py forward :297, loss :495 + autograd + torch.optim.Adam) with
recorded noise; see oracle/gen_golden.py.  Tolerances are fp32 reduction-order noise: both sides
are fp32 CPU, only op grouping differs.
"""
import numpy as np
import pytest
import torch

from oracle import restatement as R
from tests import golden_util as G

FWD = ["x_rec", "x_low", "c", "s_smp", "c_smp", "s_mean", "s_logvar", "c_prob"]
IDX = {"x_rec": 0, "x_low": 3, "c": 4, "s_smp": 5, "c_smp": 6, "s_mean": 7, "s_logvar": 8, "c_prob": 9}


@pytest.mark.parametrize("name", G.SMALL_CASES)
def test_forward_loss_grads_match_reference(name):
    g = G.load(name)
    h = G.hyper_of(g)
    sd = G.state_dict_of(g)
    x = torch.from_numpy(g["x"])
    noise = G.noise_of(g)
    out, lt, grads = R.grads_autograd(sd, [x] * h.n_arm, h, noise)
    for nm in FWD:
        got = torch.stack(list(out[IDX[nm]]))
        assert G.rel_err(got, g["fwd/" + nm]) < 5e-5, nm
    assert abs(float(lt[0]) - float(g["loss/total"])) <= 2e-6 * abs(float(g["loss/total"]))
    assert G.rel_err(lt[1], g["loss/rec"]) < 1e-6
    assert abs(float(lt[2]) - float(g["loss/joint"])) <= 2e-6 * abs(float(g["loss/joint"]))
    assert abs(float(lt[3]) - float(g["loss/c_ent"])) <= 1e-5 * abs(float(g["loss/c_ent"]))
    assert abs(float(lt[4]) - float(g["loss/c_dist"])) <= 2e-6 * abs(float(g["loss/c_dist"]))
    assert abs(float(lt[5]) - float(g["loss/c_l2"])) <= 1e-5 * abs(float(g["loss/c_l2"])) + 1e-7
    assert G.rel_err(torch.stack(lt[6]), g["loss/kl"]) < 1e-5
    assert G.rel_err(torch.stack(lt[8]), g["loss/ll"]) < 1e-6
    for k, v in grads.items():
        assert G.rel_err(v, g["grad/" + k]) < 1e-4, k
    for k in g.files:
        if k.startswith("bn1/"):
            assert G.rel_err(sd[k[4:]], g[k]) < 1e-5, k


@pytest.mark.parametrize("name", G.SMALL_CASES)
def test_manual_backward_matches_reference(name):
    """The analytic backward the HIP kernels implement == reference autograd."""
    g = G.load(name)
    h = G.hyper_of(g)
    sd = G.state_dict_of(g)
    x = torch.from_numpy(g["x"])
    _, lt, grads, _ = R.grads_manual(sd, [x] * h.n_arm, h, G.noise_of(g))
    for k, v in grads.items():
        assert G.rel_err(v, g["grad/" + k]) < 1e-4, k


@pytest.mark.parametrize("name", G.SMALL_CASES)
def test_eval_forward_matches_reference(name):
    g = G.load(name)
    h = G.hyper_of(g)
    sd = G.state_dict_of(g)
    sd.update(G.state_dict_of(g, "eval/sd/"))
    x = torch.from_numpy(g["x"])
    noise = G.noise_of(g, "eval/noise/")
    with torch.no_grad():
        out = R.forward(sd, [x] * h.n_arm, h, noise, training=False, eval_flag=True)
        lt = R.loss(out, [x] * h.n_arm, h)
    for nm in FWD:
        got = torch.stack(list(out[IDX[nm]]))
        assert G.rel_err(got, g["eval/fwd/" + nm]) < 5e-5, nm
    assert abs(float(lt[0]) - float(g["eval/loss_total"])) <= 1e-5 * abs(float(g["eval/loss_total"]))


@pytest.mark.parametrize("name", G.SMALL_CASES)
def test_adam_trajectory_matches_reference(name):
    """20 train steps (cpl_mixvae.py:434-463): loss trajectory and Adam state after 3 steps."""
    g = G.load(name)
    h = G.hyper_of(g)
    B = G.batch_of(g)
    sd = G.state_dict_of(g)
    n = g["traj"].shape[0]
    batches = [R.synthetic_batch(B, h.input_dim, seed=546 + 100 + s) for s in range(n)]
    noises = [R.draw_noise(h, B, seed=1000 + s) for s in range(n)]
    hist3, st = R.train_steps(sd, batches[:3], h, noises[:3], lr=1e-3)
    for k in R.param_keys(h):
        # Adam divides by sqrt(v): a gradient entry near zero turns rounding noise into an
        # O(lr) step, so parameters agree to a fraction of lr*steps = 3e-3, not to fp32 eps.
        # (worst case a sign flip: 2*lr per step).  Bound the worst entry by that and the typical
        # entry tightly.
        diff = (sd[k] - torch.from_numpy(g["adam3/p/" + k])).abs()
        assert float(diff.max()) < 3.1e-3 and float(diff.median()) < (2e-4 if h.hard else 2e-5), k
        if "adam3/m/" + k in g.files:
            assert G.rel_err(st["m"][k], g["adam3/m/" + k]) < 2e-4, k
            assert G.rel_err(st["v"][k], g["adam3/v/" + k]) < 2e-4, k
    hist, st = R.train_steps(sd, batches[3:], h, noises[3:], lr=1e-3, opt_state=st)
    got = np.array([float(t[0]) for t in hist3 + hist])
    ref = g["traj"][:, 0]
    rel = np.abs(got - ref) / np.abs(ref)
    if h.hard:
        # straight-through argmax is discontinuous: one flipped category (fp32 rounding) moves
        # the loss by O(1e-3) and compounds, so only the early steps are tight.
        assert rel[0] < 1e-5 and rel[1] < 1e-4 and np.all(rel < 0.2), rel
    else:
        assert np.all(rel < 5e-4), rel


def test_mid_case_scalars():
    g = G.load("mid_a2")
    h = G.hyper_of(g)
    B = G.batch_of(g)
    sd = R.init_state_dict(h, int(g["seed"]))
    x = R.synthetic_batch(B, h.input_dim)
    noise = R.draw_noise(h, B, seed=int(g["noise_seed"]))
    out, lt, grads = R.grads_autograd(sd, [x] * h.n_arm, h, noise)
    assert abs(float(lt[0]) - float(g["loss/total"])) <= 5e-6 * abs(float(g["loss/total"]))
    for k, v in grads.items():
        ref = g["gnorm/" + k]
        assert abs(float(v.double().norm()) - ref[0]) <= 2e-4 * ref[0], k


def test_dp_virtual_ranks():
    """rank-local statistics, averaged gradients, one Adam step."""
    g = G.load("tiny_a2")
    h = G.hyper_of(g)
    B = G.batch_of(g)
    ws = 2
    gsum = None
    for r in range(ws):
        sd = G.state_dict_of(g)
        x = R.synthetic_batch(B, h.input_dim, seed=546 + 200 + r)
        _, lt, grads = R.grads_autograd(sd, [x] * h.n_arm, h, R.draw_noise(h, B, seed=2000 + r))
        assert abs(float(lt[0]) - float(g[f"dp2/loss_rank{r}"])) <= 2e-6 * abs(float(lt[0]))
        gsum = grads if gsum is None else {k: gsum[k] + grads[k] for k in grads}
    sd = G.state_dict_of(g)
    for k in R.param_keys(h):
        gavg = gsum[k] / ws
        assert G.rel_err(gavg, g["dp2/grad/" + k]) < 1e-4, k
        p, _, _ = R.adam_step(sd[k], gavg, torch.zeros_like(gavg), torch.zeros_like(gavg), 1, 1e-3)
        assert G.rel_err(p, g["dp2/p/" + k]) < 1e-5, k


def test_single_arm_raises_like_reference():
    """nn_model.py:592-594 divides by len([]) when n_arm == 1."""
    h = R.Hyper(input_dim=16, fc_dim=8, n_categories=4, state_dim=2, lowD_dim=3, n_arm=1)
    sd = R.init_state_dict(h, 1)
    x = R.synthetic_batch(8, 16)
    out = R.forward(sd, [x], h, R.draw_noise(h, 8, 3))
    with pytest.raises(ZeroDivisionError):
        R.loss(out, [x], h)


def test_oracle_replays_the_reference_trainer_epochs():
    """tests/golden/epochs_a2.npz is a recording of the reference's own ``cpl_mixVAE.train`` (oracle/gen_golden_epochs.py):
    the oracle's train step + Adam, driven with the recorded noise, reproduces the per-epoch means the reference logged
    (cpl_mixvae.py:469-492), its validation numbers (:741-764) and the final parameters."""
    import os
    g = np.load(os.path.join(G.GOLDEN, "epochs_a2.npz"))
    A, B, D, H, L, C, S = (int(v) for v in g["cfg"])
    h = R.Hyper(input_dim=D, fc_dim=H, n_categories=C, state_dim=S, lowD_dim=L, n_arm=A)
    sd = {k[4:]: torch.from_numpy(g[k]).clone() for k in g.files if k.startswith("sd0/")}
    x_tr, x_te = torch.from_numpy(g["x_train"]), torch.from_numpy(g["x_test"])

    def noise(i):
        return {k: ([torch.from_numpy(a) for a in g[f"noise/{i}/{k}"]] if f"noise/{i}/{k}" in g.files else [])
                for k in ("x_mask", "u_gumbel", "u_state", "s_mask")}
    opt = None
    for e in range(int(g["n_epoch"])):
        base = 5 * e
        batches = [x_tr[i * B:(i + 1) * B] for i in range(3)]
        hist, opt = R.train_steps(sd, batches, h, [noise(base + k) for k in range(3)], lr=float(g["lr"]), opt_state=opt)
        tot = np.float32(0)
        for lt in hist:
            tot = np.float32(tot + np.float32(float(lt[0])))
        tol = 1e-5 if e == 0 else 2e-4
        assert abs(float(tot) / 3 - g["epoch/train/total-loss"][e]) <= tol * abs(g["epoch/train/total-loss"][e])
        for a in range(A):
            rec = sum(float(lt[1][a]) / D for lt in hist) / 3
            assert abs(rec - g[f"epoch/train/rec-loss{a}"][e]) <= tol * abs(rec)
        with torch.no_grad():
            out = R.forward(sd, [x_te] * A, h, noise(base + 4), training=False, eval_flag=True, update_running=False)
            lt = R.loss(out, [x_te] * A, h)
        assert abs(float(lt[0]) / len(x_te) - g["epoch/val/total-loss"][e]) <= tol * abs(g["epoch/val/total-loss"][e])
        val_rec = sum(float(v) / D for v in lt[1]) / len(x_te) / A
        assert abs(val_rec - g["epoch/val/rec-loss"][e]) <= tol * abs(val_rec)
    for k in R.param_keys(h):
        d = (sd[k] - torch.from_numpy(g["sdT/" + k])).abs()
        assert float(d.median()) < 1e-5 and float(d.max()) < 2.1e-3, k


def test_oracle_reproduces_the_masked_forward_fixture():
    """tests/golden/mask_a2.npz (oracle/gen_golden_mask.py: the REAL reference's forward(mask=...), nn_model.py:332-335,
    train mode with loss and gradients, then eval mode on the updated running statistics)."""
    g = G.load("mask_a2")
    h = G.hyper_of(g)
    A = h.n_arm
    mask = [int(v) for v in g["mask"]]
    x = torch.from_numpy(g["x"])
    sd = G.state_dict_of(g)
    out, lt, grads = R.grads_autograd(sd, [x] * A, h, G.noise_of(g), mask=mask)
    names = {0: "x_rec", 3: "x_low", 4: "c", 5: "s_smp", 6: "c_smp", 7: "s_mean", 8: "s_logvar", 9: "c_prob"}
    for i, nm in names.items():
        assert G.rel_err(torch.stack(list(out[i])), g["fwd/" + nm]) < 2e-4, nm
    # (fp32 against fp32: the masked-out categories put log(eps) / sqrt(eps) = -1.8e5 into both arms' coupling operands)
    assert abs(float(lt[0]) - float(g["loss/total"])) <= 1e-4 * abs(float(g["loss/total"]))
    for k, v in grads.items():
        assert G.rel_err(v, g["grad/" + k]) < 2e-4, k
    oe = R.forward(sd, [x] * A, h, G.noise_of(g, "noise_eval/"), training=False, eval_flag=True, mask=mask)
    for i, nm in names.items():
        assert G.rel_err(torch.stack(list(oe[i])), g["eval/" + nm]) < 2e-4, nm
