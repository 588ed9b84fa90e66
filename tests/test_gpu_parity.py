"""-m gpu: parity of the HIP path (through the Python boundary -> C ABI -> kernels) with
 (1) the committed golden vectors generated from the REAL reference model (tests/golden, made by
     oracle/gen_golden.py from /root/reference/mmidas/nn_model.py + autograd + torch.optim.Adam),
 (2) the oracle restatement (oracle/restatement.py) on shapes the fixtures do not cover,
 (3) size-independent properties at BASELINE.json's full size (B=5000, D=5000, A=2).

Tolerances (fp32; stated per SURVEY.md section 8c).  All are relative to the largest magnitude of the
compared tensor, because tau=0.005 and inv_var ~ 1e4 put losses at 1e8..1e11 and gradients at 1e10+:
  forward activations 1e-4, loss scalars 1e-5, gradients 1e-3, Adam parameters: see test.
The reference-vs-oracle fp32 noise floor measured on CPU is ~1e-5 on gradients (tests/test_oracle_*).
"""
import numpy as np
import pytest
import torch

from oracle import restatement as R
from tests import golden_util as G

pytestmark = pytest.mark.gpu

FWD_TOL, LOSS_TOL, GRAD_TOL = 1e-4, 1e-5, 1e-3
FWD = {"x_rec": 0, "x_low": 3, "c": 4, "s_smp": 5, "c_smp": 6, "s_mean": 7, "s_logvar": 8, "c_prob": 9}


def _U():
    from tests import gpu_util as U
    return U


def _loss_close(got, ref, tol=LOSS_TOL):
    got, ref = float(got), float(ref)
    assert abs(got - ref) <= tol * abs(ref) + 1e-7, (got, ref)


@pytest.mark.parametrize("name", G.SMALL_CASES)
def test_golden_forward_loss_grads(name):
    U = _U()
    g = G.load(name)
    h = G.hyper_of(g)
    m = U.build_model(h, G.state_dict_of(g))
    m.train()
    x = torch.from_numpy(g["x"]).to(U.DEV)
    out, lt, grads = U.run_step(m, x, G.noise_of(g))
    for nm, i in FWD.items():
        got = torch.stack([t.cpu() for t in out[i]])
        assert G.rel_err(got, g["fwd/" + nm]) < FWD_TOL, nm
    _loss_close(lt[0], g["loss/total"])
    assert G.rel_err(lt[1].cpu(), g["loss/rec"]) < LOSS_TOL
    _loss_close(lt[2], g["loss/joint"])
    _loss_close(lt[3], g["loss/c_ent"], 1e-4)
    _loss_close(lt[4], g["loss/c_dist"])
    _loss_close(lt[5], g["loss/c_l2"], 1e-4)
    assert G.rel_err(torch.stack([t.cpu() for t in lt[6]]), g["loss/kl"]) < 1e-4
    assert G.rel_err(torch.stack([t.cpu() for t in lt[8]]), g["loss/ll"]) < LOSS_TOL
    assert not lt[1].requires_grad                      # loss_recs is detached (nn_model.py:590)
    for k, v in grads.items():
        assert G.rel_err(v, g["grad/" + k]) < GRAD_TOL, k
    sd = m.state_dict()
    for k in g.files:
        if k.startswith("bn1/"):
            if "num_batches" in k:
                assert int(sd[k[4:]]) == int(g[k]), k
            else:
                assert G.rel_err(sd[k[4:]].cpu(), g[k]) < 1e-5, k


@pytest.mark.parametrize("name", G.SMALL_CASES)
def test_golden_eval_forward(name):
    """model.eval() + forward(eval=True): running-stat BN, no dropout, no Gumbel noise, hard sample."""
    U = _U()
    g = G.load(name)
    h = G.hyper_of(g)
    sd = G.state_dict_of(g)
    sd.update(G.state_dict_of(g, "eval/sd/"))
    m = U.build_model(h, sd)
    m.eval()
    x = torch.from_numpy(g["x"]).to(U.DEV)
    with torch.no_grad():
        out, lt, _ = U.run_step(m, x, G.noise_of(g, "eval/noise/"), eval_flag=True, backward=False)
    for nm, i in FWD.items():
        got = torch.stack([t.cpu() for t in out[i]])
        assert G.rel_err(got, g["eval/fwd/" + nm]) < FWD_TOL, nm
    _loss_close(lt[0], g["eval/loss_total"], 1e-4)
    for k, v in m.state_dict().items():                 # eval must not touch the running statistics
        if "running" in k:
            assert torch.equal(v.cpu(), sd[k]), k


@pytest.mark.parametrize("name", G.SMALL_CASES)
def test_golden_adam_trajectory(name):
    """20 fused train steps (cpl_mixvae.py:434-463) on seeded explicit noise vs the reference's
    loss trajectory, and parameters after 3 Adam steps."""
    U = _U()
    from distributed_vae_amd.cpl_mixvae import FusedAdam
    g = G.load(name)
    h = G.hyper_of(g)
    B = G.batch_of(g)
    m = U.build_model(h, G.state_dict_of(g))
    m.train()
    opt = FusedAdam(m, lr=1e-3)
    n = g["traj"].shape[0]
    got = []
    for s in range(n):
        x = R.synthetic_batch(B, h.input_dim, seed=546 + 100 + s).to(U.DEV)
        m.set_explicit_noise(U.noise_to_device(R.draw_noise(h, B, seed=1000 + s)))
        buf = m.fused_train_step(x.expand(h.n_arm, -1, -1), 1.0, opt, do_adam=True)
        got.append(buf.cpu().numpy().copy())
        if s == 2:
            for k, p in m.named_parameters():
                diff = (p.detach().cpu() - torch.from_numpy(g["adam3/p/" + k])).abs()
                # Adam turns rounding noise on near-zero gradients into O(lr) steps: bound the worst
                # entry by lr * steps and the typical entry tightly (same rule as the CPU oracle test)
                assert float(diff.max()) < 3.1e-3 and float(diff.median()) < (2e-4 if h.hard else 2e-5), k
    got = np.array(got)
    ref = g["traj"]
    rel = np.abs(got[:, 0] - ref[:, 0]) / np.abs(ref[:, 0])
    if h.hard:   # straight-through argmax is discontinuous; only early steps are tight
        assert rel[0] < 1e-5 and rel[1] < 1e-3 and np.all(rel < 0.3), rel
    else:
        assert np.all(rel < 1e-3), rel
        A = h.n_arm
        assert np.all(np.abs(got[:, 1] - ref[:, 1]) <= 1e-3 * np.abs(ref[:, 1]))                  # joint
        assert np.all(np.abs(got[:, 5:5 + A] - ref[:, 5:5 + A]) <= 1e-3 * np.abs(ref[:, 5:5 + A]) + 1e-4)  # rec


def test_golden_mid_case():
    """B=512, D=1024, H=100, L=10, C=92, S=2: the reference's layer widths."""
    U = _U()
    g = G.load("mid_a2")
    h = G.hyper_of(g)
    B = G.batch_of(g)
    m = U.build_model(h, R.init_state_dict(h, int(g["seed"])))
    m.train()
    x = R.synthetic_batch(B, h.input_dim).to(U.DEV)
    out, lt, grads = U.run_step(m, x, R.draw_noise(h, B, seed=int(g["noise_seed"])))
    _loss_close(lt[0], g["loss/total"])
    assert G.rel_err(lt[1].cpu(), g["loss/rec"]) < LOSS_TOL
    _loss_close(lt[4], g["loss/c_dist"])
    for nm, i in [("c", 4), ("c_smp", 6), ("s_mean", 7), ("s_logvar", 8), ("x_low", 3)]:
        got = torch.stack([t.cpu() for t in out[i]])[:, :16]
        assert G.rel_err(got, g["fwd/" + nm]) < FWD_TOL, nm
    xs = np.array([float(t.double().sum()) for t in out[0]])
    assert np.all(np.abs(xs - g["fwd/x_rec_sum"]) <= 1e-5 * np.abs(g["fwd/x_rec_sum"]))
    for k, v in grads.items():
        ref = g["gnorm/" + k]
        assert abs(float(v.double().norm()) - ref[0]) <= GRAD_TOL * ref[0], k
        assert abs(float(v.double().abs().max()) - ref[2]) <= GRAD_TOL * ref[2], k
        if "grad/" + k in g.files:
            assert G.rel_err(v, g["grad/" + k]) < GRAD_TOL, k


def test_fused_step_matches_api_path_fc100():
    """fc_dim = 100 takes the specialised kernels in the fused train step (fc11 + loss + dZ11 + d(d10) in one kernel, the
    96 + 4 column GEMMs, coupling / loss scalars / dW11 on the side stream, Adam inside the slab reduction).  Its loss
    vector and gradients must equal what forward() / loss() / backward() give on the same inputs (that path is pinned
    to the reference by test_golden_mid_case), and its Adam step what torch.optim.Adam does with those gradients."""
    U = _U()
    from distributed_vae_amd.cpl_mixvae import FusedAdam
    g = G.load("mid_a2")
    h = G.hyper_of(g)
    B = G.batch_of(g)
    assert h.fc_dim == 100
    sd = R.init_state_dict(h, int(g["seed"]))
    x = R.synthetic_batch(B, h.input_dim).to(U.DEV)
    noise = R.draw_noise(h, B, seed=int(g["noise_seed"]))
    m1 = U.build_model(h, sd); m1.train()
    _, lt, grads = U.run_step(m1, x, noise)
    # gradients only
    m2 = U.build_model(h, sd); m2.train()
    m2.set_explicit_noise(U.noise_to_device(noise))
    buf = m2.fused_train_step(x.expand(h.n_arm, -1, -1), 1.0, None, do_adam=False)
    torch.cuda.synchronize()
    _loss_close(buf[0], lt[0], 1e-6)
    for (k, _), gv in zip(m2.named_parameters(), m2._grad_views):   # the fused step leaves the gradient in the flat buffer
        assert G.rel_err(gv.cpu(), grads[k]) < 1e-5, k               # same arithmetic up to the d(d10) slab order
    # with the Adam update riding on the reduction
    m3 = U.build_model(h, sd); m3.train()
    m3.set_explicit_noise(U.noise_to_device(noise))
    opt = FusedAdam(m3, lr=1e-3)
    m3.fused_train_step(x.expand(h.n_arm, -1, -1), 1.0, opt, do_adam=True)
    torch.cuda.synchronize()
    ref = {k: torch.nn.Parameter(v.clone()) for k, v in sd.items() if k in grads}
    topt = torch.optim.Adam(list(ref.values()), lr=1e-3)
    for k, p in ref.items():
        p.grad = grads[k].clone()
    topt.step()
    for k, p in m3.named_parameters():
        diff = (p.detach().cpu() - ref[k].detach()).abs()
        assert float(diff.max()) < 1.1e-3 and float(diff.median()) < 1e-5, (k, float(diff.max()), float(diff.median()))


ODD = [
    # A, B, D, H, L, C, S, hard, s_drop, x_drop
    (2, 33, 50, 10, 3, 8, 2, False, 0.0, 0.5),       # D % 4 != 0, ragged batch
    (3, 70, 131, 20, 7, 17, 3, False, 0.1, 0.3),     # odd everything, three row blocks
    (2, 129, 257, 100, 10, 92, 2, True, 0.0, 0.0),   # no input dropout, hard samples
    (4, 64, 96, 128, 16, 128, 4, False, 0.0, 0.5),   # kernel limits H = C = 128
]


@pytest.mark.parametrize("cfg", ODD)
def test_vs_oracle_odd_shapes(cfg):
    """Ragged / limit shapes against the oracle (not covered by the committed fixtures)."""
    U = _U()
    A, B, D, H, L, C, S, hard, sdrop, xdrop = cfg
    h = R.Hyper(input_dim=D, fc_dim=H, n_categories=C, state_dim=S, lowD_dim=L, x_drop=xdrop, s_drop=sdrop, n_arm=A,
                hard=hard)
    sd = R.init_state_dict(h, 11)
    x = R.synthetic_batch(B, D, seed=3)
    noise = R.draw_noise(h, B, seed=5)
    _, lt_r, g_r = R.grads_autograd({k: v.clone() for k, v in sd.items()}, [x] * A, h, noise)
    m = U.build_model(h, sd)
    m.train()
    out, lt, grads = U.run_step(m, x.to(U.DEV), noise)
    _loss_close(lt[0], lt_r[0])
    for k, v in grads.items():
        assert G.rel_err(v, g_r[k]) < GRAD_TOL, k


EDGE = [
    # A, B, D, H, L, C, S, hard, s_drop, x_drop
    (2, 2, 8, 4, 1, 2, 1, False, 0.0, 0.5),          # smallest legal everything (B = 2 rows for batch statistics)
    (8, 16, 32, 8, 2, 4, 1, False, 0.0, 0.5),        # MMVAE_MAX_ARMS arms: 28 coupled pairs
    (2, 300, 64, 128, 64, 128, 32, False, 0.5, 0.9), # every kernel limit at once, heavy dropout
    (3, 65, 260, 36, 9, 30, 2, True, 0.0, 0.5),      # one cell past a 64-row block, 4 genes past a 64-gene tile
    (2, 70, 64, 32, 32, 96, 16, False, 0.2, 0.5),    # the limits of the half-wave latent kernels (C = 96, L = 32, 2S = 32)
    (2, 70, 64, 32, 33, 97, 2, True, 0.0, 0.5),      # one past them: the one-wave-per-cell kernels
]


@pytest.mark.parametrize("cfg", EDGE)
def test_vs_oracle_edge_shapes(cfg):
    """Minimum sizes, maximum supported sizes and off-by-one tile boundaries against the oracle."""
    U = _U()
    A, B, D, H, L, C, S, hard, sdrop, xdrop = cfg
    h = R.Hyper(input_dim=D, fc_dim=H, n_categories=C, state_dim=S, lowD_dim=L, x_drop=xdrop, s_drop=sdrop, n_arm=A,
                hard=hard)
    sd = R.init_state_dict(h, 21)
    x = R.synthetic_batch(B, D, seed=8)
    noise = R.draw_noise(h, B, seed=13)
    out_r, lt_r, g_r = R.grads_autograd({k: v.clone() for k, v in sd.items()}, [x] * A, h, noise)
    m = U.build_model(h, sd)
    m.train()
    out, lt, grads = U.run_step(m, x.to(U.DEV), noise)
    _loss_close(lt[0], lt_r[0], 5e-5 if B == 2 else LOSS_TOL)
    assert G.rel_err(torch.stack([t.cpu() for t in out[0]]), torch.stack(list(out_r[0]))) < FWD_TOL
    gmax = max(float(v.abs().max()) for v in g_r.values())
    for k, v in grads.items():
        if float(g_r[k].abs().max()) < 1e-6 * gmax:
            # B = 2: BatchNorm of two rows is +-1, its backward cancels exactly; the reference only has
            # rounding noise upstream of a BatchNorm there.  Require "negligible", not "equal noise".
            assert float(v.abs().max()) < 1e-5 * gmax, k
        else:
            assert G.rel_err(v, g_r[k]) < GRAD_TOL, k


def test_unsupported_shapes_are_rejected():
    """Beyond the kernels' limits the engine refuses (NotImplementedError), it never degrades silently."""
    U = _U()
    for kw in (dict(fc_dim=129), dict(n_categories=129), dict(lowD_dim=65), dict(n_arm=9)):
        base = dict(input_dim=32, fc_dim=8, n_categories=4, state_dim=2, lowD_dim=3, n_arm=2)
        base.update(kw)
        h = R.Hyper(**base)
        m = U.build_model(h)
        x = torch.zeros(8, 32, device=U.DEV)
        with pytest.raises(NotImplementedError):
            m(x.expand(h.n_arm, -1, -1), 1.0)


def test_distinct_input_per_arm():
    """x given as a list of different tensors (augmenter-style input, cpl_mixvae.py:422-423)."""
    U = _U()
    h = R.Hyper(input_dim=64, fc_dim=16, n_categories=7, state_dim=2, lowD_dim=5, n_arm=2)
    B = 40
    sd = R.init_state_dict(h, 2)
    xs = [R.synthetic_batch(B, 64, seed=s) for s in (1, 2)]
    noise = R.draw_noise(h, B, seed=9)
    _, lt_r, g_r = R.grads_autograd({k: v.clone() for k, v in sd.items()}, xs, h, noise)
    m = U.build_model(h, sd)
    m.train()
    m.set_explicit_noise(U.noise_to_device(noise))
    xd = [t.to(U.DEV) for t in xs]
    out = m(xd, 1.0, 0.0)
    lt = m.loss(out[0], [], [], xd, out[7], out[8], out[4], out[6], 0.0)
    lt[0].backward()
    _loss_close(lt[0], lt_r[0])
    for k, p in m.named_parameters():
        assert G.rel_err(p.grad.cpu(), g_r[k]) < GRAD_TOL, k


def test_philox_step_equals_explicit_replay():
    """The in-kernel Philox mode and the explicit-noise mode run the same arithmetic: dump the noise
    a Philox step uses, replay it explicitly, results must be bit-identical.  Also sanity-checks
    the generator's first moments."""
    U = _U()
    from distributed_vae_amd import _native as N
    h = R.Hyper(input_dim=512, fc_dim=32, n_categories=20, state_dim=2, lowD_dim=6, n_arm=3, s_drop=0.25)
    B = 200
    sd = R.init_state_dict(h, 4)
    x = R.synthetic_batch(B, h.input_dim).to(U.DEV)
    m = U.build_model(h, sd)
    m.train()
    m._noise_seed, m._noise_offset = 1234567, 41
    out = m(x.expand(3, -1, -1), 1.0, 0.0)
    lt = m.loss(out[0], [], [], None, out[7], out[8], out[4], out[6], 0.0)
    lt[0].backward()
    g1 = m.flat_grad().clone()
    l1 = float(lt[0])
    noise = m._engine.dump_noise(m._hyper(1.0, False), N.make_noise(None, 1234567, 42))
    keep = float(noise["x_mask"].float().mean())
    assert abs(keep - 0.5) < 0.01
    assert abs(float(noise["s_mask"].float().mean()) - 0.75) < 0.05
    assert abs(float(noise["u_gumbel"].mean()) - 0.5) < 0.02 and float(noise["u_gumbel"].max()) < 1.0
    assert float(noise["u_gumbel"].min()) >= 0.0
    # arms draw different masks
    assert float((noise["x_mask"][0] != noise["x_mask"][1]).float().mean()) > 0.4
    m2 = U.build_model(h, sd)
    m2.train()
    m2.set_explicit_noise(noise)
    out2 = m2(x.expand(3, -1, -1), 1.0, 0.0)
    lt2 = m2.loss(out2[0], [], [], None, out2[7], out2[8], out2[4], out2[6], 0.0)
    lt2[0].backward()
    assert float(lt2[0]) == l1
    assert torch.equal(m2.flat_grad(), g1)
    # a different offset gives different noise
    n3 = m._engine.dump_noise(m._hyper(1.0, False), N.make_noise(None, 1234567, 43))
    assert float((n3["x_mask"] != noise["x_mask"]).float().mean()) > 0.4


@pytest.mark.parametrize("D,H", [(50, 10), (100, 12)])   # general kernels (D % 4 != 0) / fast path with D % 8 != 0
def test_philox_replay_other_paths(D, H):
    """Same replay check on the general (unaligned) kernels and on the fast path's D % 8 != 0 branch."""
    U = _U()
    from distributed_vae_amd import _native as N
    h = R.Hyper(input_dim=D, fc_dim=H, n_categories=9, state_dim=2, lowD_dim=4, n_arm=2)
    B = 70
    sd = R.init_state_dict(h, 4)
    x = R.synthetic_batch(B, D).to(U.DEV)
    res = []
    noise = None
    for mode in ("philox", "explicit"):
        m = U.build_model(h, sd)
        m.train()
        if mode == "philox":
            m._noise_seed, m._noise_offset = 77, 4
        else:
            m.set_explicit_noise(noise)
        out = m(x.expand(2, -1, -1), 1.0, 0.0)
        lt = m.loss(out[0], [], [], None, out[7], out[8], out[4], out[6], 0.0)
        lt[0].backward()
        res.append((float(lt[0]), m.flat_grad().clone()))
        if mode == "philox":
            noise = m._engine.dump_noise(m._hyper(1.0, False), N.make_noise(None, 77, 5))
            assert 0.3 < float(noise["x_mask"].float().mean()) < 0.7
    assert res[0][0] == res[1][0]
    assert torch.equal(res[0][1], res[1][1])


@pytest.mark.parametrize("x_drop,bits", [(0.5, 1), (0.25, 2), (0.8125, 4), (0.98828125, 8), (0.3, 16), (0.0, 1)])
@pytest.mark.parametrize("D,H", [(1000, 100), (50, 10)])   # bit-image fast path (D % 32 != 0) / general kernels
def test_philox_mask_field_widths(x_drop, bits, D, H):
    """The input-dropout mask spends 1, 2, 4, 8 or 16 random bits per element, the fewest that represent the keep
    probability exactly (csrc/common.hpp xmask_keep).  For every width: keep fraction, no correlation between
    neighbouring genes / cells / arms, and the Philox step is bit-identical to the explicit replay of its dump."""
    U = _U()
    from distributed_vae_amd import _native as N
    h = R.Hyper(input_dim=D, fc_dim=H, n_categories=9, state_dim=2, lowD_dim=4, n_arm=2, x_drop=x_drop)
    B = 300
    sd = R.init_state_dict(h, 4)
    x = R.synthetic_batch(B, D).to(U.DEV)
    res = []
    noise = None
    for mode in ("philox", "explicit"):
        m = U.build_model(h, sd)
        m.train()
        if mode == "philox":
            m._noise_seed, m._noise_offset = 99, 10
        else:
            m.set_explicit_noise(noise)
        out = m(x.expand(2, -1, -1), 1.0, 0.0)
        lt = m.loss(out[0], [], [], None, out[7], out[8], out[4], out[6], 0.0)
        lt[0].backward()
        res.append((float(lt[0]), m.flat_grad().clone()))
        if mode == "philox":
            noise = m._engine.dump_noise(m._hyper(1.0, False), N.make_noise(None, 99, 11))
            k = noise["x_mask"].float()            # [A, B, D]
            keep = 1.0 - x_drop
            n = k[0].numel()
            tol = 5.0 * (keep * (1 - keep) / n) ** 0.5 + 1e-12
            assert abs(float(k.mean()) - keep) < tol, (float(k.mean()), keep)
            if 0.0 < keep < 1.0:
                pair_tol = 6.0 / n ** 0.5
                assert abs(float((k[:, :, 1:] * k[:, :, :-1]).mean()) - keep * keep) < pair_tol   # neighbouring genes
                assert abs(float((k[:, 1:] * k[:, :-1]).mean()) - keep * keep) < pair_tol         # neighbouring cells
                assert abs(float((k[0] * k[1]).mean()) - keep * keep) < pair_tol                  # arms
                assert abs(float(k.mean(dim=(0, 1)).std()) - (keep * (1 - keep) / (2 * B)) ** 0.5) < 0.01   # per gene
    assert res[0][0] == res[1][0]
    assert torch.equal(res[0][1], res[1][1])


def test_adam_kernel_matches_torch():
    U = _U()
    from distributed_vae_amd import _native as N
    torch.manual_seed(0)
    n = 100003
    for wd, dec, Opt in [(0.0, False, torch.optim.Adam), (0.01, True, torch.optim.AdamW), (0.02, False, torch.optim.Adam)]:
        p0 = torch.randn(n)
        p_ref = p0.clone().requires_grad_(True)
        opt = Opt([p_ref], lr=1e-3, weight_decay=wd)
        p = p0.clone().to(U.DEV)
        m = torch.zeros_like(p)
        v = torch.zeros_like(p)
        for step in range(1, 6):
            g = torch.randn(n) * (10.0 ** (step - 3))
            p_ref.grad = g.clone()
            opt.step()
            N.adam_step(p, g.to(U.DEV), m, v, step, 1e-3, wd=wd, decoupled=dec)
        assert float((p.cpu() - p_ref.detach()).abs().max()) < 2e-6
        st = opt.state[p_ref]
        # (1-b2)*g*g is associated differently from torch's addcmul: a few ulp
        assert G.rel_err(m.cpu(), st["exp_avg"]) < 1e-5 and G.rel_err(v.cpu(), st["exp_avg_sq"]) < 5e-5


def test_torch_optimizer_interop_and_state_dict():
    """Reference-style loop: zero_grad / forward / loss / backward / torch.optim.Adam.step on
    model.parameters() (cpl_mixvae.py:274, :434-463) and a state_dict round trip."""
    U = _U()
    h = R.Hyper(input_dim=64, fc_dim=16, n_categories=7, state_dim=2, lowD_dim=5, n_arm=2)
    B = 32
    sd = R.init_state_dict(h, 546)
    x = R.synthetic_batch(B, 64)
    noises = [R.draw_noise(h, B, seed=70 + s) for s in range(3)]
    ref_sd = {k: v.clone() for k, v in sd.items()}
    hist, _ = R.train_steps(ref_sd, [x] * 3, h, noises, lr=1e-3)
    m = U.build_model(h, sd)
    m.train()
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    xd = x.to(U.DEV).expand(2, -1, -1)
    for s in range(3):
        m.set_explicit_noise(U.noise_to_device(noises[s]))
        opt.zero_grad()
        out = m(xd, 1.0, 0.0)
        lt = m.loss(out[0], [], [], xd, out[7], out[8], out[4], out[6], 0.0)
        lt[0].backward()
        opt.step()
        _loss_close(lt[0], hist[s][0], 1e-4)
    assert m._is_packed()                                 # optimizer updated the flat views in place
    sd2 = {k: v.cpu().clone() for k, v in m.state_dict().items()}
    assert list(sd2.keys()) == list(ref_sd.keys())
    m3 = U.build_model(h, sd2)
    for k, v in m3.state_dict().items():
        assert torch.equal(v.cpu(), sd2[k]), k


def test_single_arm_raises_like_reference():
    U = _U()
    h = R.Hyper(input_dim=32, fc_dim=8, n_categories=4, state_dim=2, lowD_dim=3, n_arm=1)
    m = U.build_model(h, R.init_state_dict(h, 1))
    m.train()
    x = R.synthetic_batch(8, 32).to(U.DEV)
    out = m(x.expand(1, -1, -1), 1.0, 0.0)
    with pytest.raises(ZeroDivisionError):
        m.loss(out[0], [], [], None, out[7], out[8], out[4], out[6], 0.0)


def test_c_abi_error_codes():
    _U()
    import ctypes as C
    from distributed_vae_amd import _native as N
    L = N.lib()
    bad = N.Dims(2, 32, 64, 200, 5, 7, 2)               # fc_dim > 128
    assert L.mmvae_check_dims(C.byref(bad)) == -2
    assert b"unsupported" in L.mmvae_last_error_string()
    assert L.mmvae_check_dims(C.byref(N.Dims(2, 0, 64, 16, 5, 7, 2))) == -1   # B < 1
    d = N.Dims(2, 32, 64, 16, 5, 7, 2)
    hy = N.Hyper(0.005, 1.0, 1.0, 1.0, 1e-8, 0.01, 0.5, 0.0, 0, 1, 0)
    nz = N.make_noise(None, 1, 1)
    ws = torch.empty(1024, device="cuda:0")
    x = torch.zeros(32, 64, device="cuda:0")
    p = torch.zeros(100000, device="cuda:0")
    rc = L.mmvae_forward(C.byref(d), C.byref(hy), C.byref(nz), p.data_ptr(), None, None, x.data_ptr(), 0, None, 0,
                         ws.data_ptr(), ws.numel() * 4, None, None)
    assert rc == -4 and b"workspace" in L.mmvae_last_error_string()
    with pytest.raises(NotImplementedError):
        N.Engine(2, 32, 64, 200, 5, 7, 2, "cuda:0")


# ---------------------------------------------------------------------------------------------------
# BASELINE.json configs[1] full size: A=2, B=5000 cells x D=5000 genes, fp32
# ---------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def full():
    U = _U()
    h = R.Hyper()                                       # defaults = the benchmark configuration
    B = 5000
    sd = R.init_state_dict(h, 546)
    x = R.synthetic_batch(B, h.input_dim)
    noise = R.draw_noise(h, B, seed=7)
    m = U.build_model(h, sd)
    m.train()
    return U, h, sd, x, noise, m


def test_full_size_against_oracle(full):
    """One full-size step against the oracle (seconds of CPU), both compared with an fp64 evaluation
    of the same step.  Bulk accuracy must match the fp32 CPU oracle's (90th-percentile entry error
    within 3x or below 1e-4; measured 1e-7..1e-6 for both, up to 5e-5 on a bias gradient downstream of
    a flipped decision).  The *maximum* entry error is looser: at B*D = 25 M
    ReLU / 0.1-threshold decisions a few pre-activations sit within fp32 rounding of zero, and a
    flipped decision moves one row of a weight gradient by ~1e-4..1e-3 of its scale in ANY fp32
    evaluation order (tools/gpu_err_stats.py prints the statistics for both sides)."""
    U, h, sd, x, noise, m = full
    out, lt, grads = U.run_step(m, x.to(U.DEV), noise)
    # the oracle on the device's own ReLU decisions (tests/gpu_util.py::flip_aware_oracle: a decision may differ from the
    # fp64 oracle's only at a pre-activation within fp32 rounding of zero, at most 8 hidden ones do)
    fo = U.flip_aware_oracle(h, sd, x, noise, U.device_relu_patterns(m._engine, h))
    _loss_close(lt[0], fo["lt_64"][0])
    _loss_close(lt[0], fo["lt_32"][0])
    assert G.rel_err(lt[1].cpu(), fo["lt_64"][1]) < LOSS_TOL
    U.assert_gradients_tight(grads, fo, GRAD_TOL)


def test_full_size_properties(full):
    """Size-independent properties: determinism, gradient linearity in the upstream scale,
    invariance to the GEMM split factors, arm-permutation equivariance."""
    U, h, sd, x, noise, m = full
    from distributed_vae_amd import _native as N
    xd = x.to(U.DEV)
    _, lt1, g1 = U.run_step(m, xd, noise)
    m2 = U.build_model(h, sd); m2.train()
    _, lt2, g2 = U.run_step(m2, xd, noise)
    assert float(lt1[0]) == float(lt2[0])
    for k in g1:
        assert torch.equal(g1[k], g2[k]), k               # bitwise reproducible (no float atomics)
    # linearity: backward of 3 * loss
    m3 = U.build_model(h, sd); m3.train()
    m3.set_explicit_noise(U.noise_to_device(noise))
    xs = xd.expand(h.n_arm, -1, -1)
    out = m3(xs, 1.0, 0.0)
    lt = m3.loss(out[0], [], [], xs, out[7], out[8], out[4], out[6], 0.0)
    (3.0 * lt[0]).backward()
    for k, p in m3.named_parameters():
        assert G.rel_err(p.grad.cpu(), 3.0 * g1[k]) < 1e-6, k
    # split factors change only the summation order (they travel in the engine's mmvae_exec)
    ex = N.exec_from_env()
    for which, val in [(0, 3), (1, 5), (2, 7), (3, 9)]:
        ex.split[which] = val
    m4 = U.build_model(h, sd); m4.train()
    m4._exec = ex
    _, lt4, g4 = U.run_step(m4, xd, noise)
    assert m4._engine.splits()[:4] == [3, 5, 7, 9]
    _loss_close(lt4[0], lt1[0], 1e-6)
    # summation-order noise only.  A pre-activation within fp32 rounding of zero may be decided differently under another
    # summation order, and one differing hidden decision moves every bias gradient below it: when the two runs took the
    # same hidden decisions everywhere, EVERY tensor is held to the tight bound against the first run; otherwise the
    # second run is held to the flip-aware oracle gate by itself (as the first is in test_full_size_against_oracle)
    p1, p4 = U.device_relu_patterns(m._engine, h), U.device_relu_patterns(m4._engine, h)
    k_diff = sum(int((p1[s_] != p4[s_]).sum()) for s_ in U.HIDDEN_SITES)
    print("hidden ReLU decisions that differ between the two split settings:", k_diff)
    assert k_diff <= 8
    if k_diff == 0:
        for k in g1:
            e = ((g4[k].double() - g1[k].double()).abs() / (float(g1[k].abs().max()) + 1e-30)).flatten()
            q = float(e.kthvalue(max(1, int(0.9 * e.numel()))).values)
            assert q < 1e-4 and float(e.max()) < 5 * GRAD_TOL, (k, q, float(e.max()))
    else:
        U.assert_gradients_tight(g4, U.flip_aware_oracle(h, sd, x, noise, p4), GRAD_TOL)
    del p1, p4
    # the fused train step (fused fc11 / d(d10) kernel, side stream) against the separately-called API path
    m6 = U.build_model(h, sd); m6.train()
    m6.set_explicit_noise(U.noise_to_device(noise))
    buf6 = m6.fused_train_step(xs, 1.0, None, do_adam=False)
    torch.cuda.synchronize()
    _loss_close(buf6[0], lt1[0], 1e-6)
    for (k, _), gv in zip(m6.named_parameters(), m6._grad_views):
        e = ((gv.cpu().double() - g1[k].double()).abs() / (float(g1[k].abs().max()) + 1e-30)).flatten()
        p90 = float(e.kthvalue(max(1, int(0.9 * e.numel()))).values)
        assert p90 < 1e-4 and float(e.max()) < 5 * GRAD_TOL, (k, p90, float(e.max()))
    # swapping the two arms (parameters and noise) swaps their gradients
    sw = {}
    for k, v in sd.items():
        name, a, rest = k.split(".")
        sw[f"{name}.{1 - int(a)}.{rest}"] = v
    nsw = {k: (v[::-1] if v else v) for k, v in noise.items()}
    m5 = U.build_model(h, sw); m5.train()
    _, lt5, g5 = U.run_step(m5, xd, nsw)
    _loss_close(lt5[0], lt1[0], 1e-6)
    for k in g1:
        name, a, rest = k.split(".")
        assert G.rel_err(g5[f"{name}.{1 - int(a)}.{rest}"], g1[k]) < 1e-5, k


def test_trainer_loop_and_checkpoint(tmp_path):
    """cpl_mixVAE.init_model / train / checkpoint (cpl_mixvae.py:193-286, :397-492, :777-788)."""
    _U()
    from distributed_vae_amd.cpl_mixvae import cpl_mixVAE
    torch.manual_seed(546)
    N_, Dm = 640, 128
    x = R.synthetic_batch(N_, Dm)
    ds = torch.utils.data.TensorDataset(x, torch.arange(N_))
    tr = torch.utils.data.DataLoader(ds, batch_size=128, shuffle=False, drop_last=True)
    te = torch.utils.data.DataLoader(ds, batch_size=320, shuffle=False)
    t = cpl_mixVAE(saving_folder=str(tmp_path), device=0)
    t.init_model(n_categories=12, state_dim=2, input_dim=Dm, fc_dim=32, lowD_dim=6, x_drop=0.5, s_drop=0.0,
                 lr=1e-3, n_arm=2, temp=1.0, tau=0.005)
    hist = t.train(tr, te, n_epoch=12, n_epoch_p=0, good_enuf_consensus=2.0)
    assert len(hist["losses"]) == 12 and np.all(np.isfinite(hist["losses"]))
    assert hist["losses"][-1] < hist["losses"][0]
    ck = tmp_path / "model" / "cpl_mixVAE_model_epoch_10.pth"
    assert ck.exists()
    loaded = torch.load(ck, map_location="cpu", weights_only=True)
    assert set(loaded) == {"model_state_dict", "optimizer_state_dict"}
    assert len(loaded["model_state_dict"]) == 92
    # the optimizer state loads into a stock torch.optim.Adam over the same parameters
    t2 = cpl_mixVAE(saving_folder="", device=0, save_flag=False)
    t2.init_model(n_categories=12, state_dim=2, input_dim=Dm, fc_dim=32, lowD_dim=6, n_arm=2,
                  trained_model=str(ck))
    ref_opt = torch.optim.Adam(t2.model.parameters(), lr=1e-3)
    ref_opt.load_state_dict(loaded["optimizer_state_dict"])
    assert t2.optimizer.step_count == 55                  # 11 epochs x 5 batches
    for k, v in t2.model.state_dict().items():
        assert torch.equal(v.cpu(), loaded["model_state_dict"][k]), k


def test_full_size_step_is_bit_reproducible():
    """Every reduction of the step has a fixed order: identical state and noise give bit-identical gradients and
    losses run after run at the benchmark shape.  A run-to-run difference means a race or an unpadded hardware hazard
    (tools/soak_determinism.py runs the same check for hundreds of iterations)."""
    U = _U()
    from distributed_vae_amd import _native as N
    A, B, D = 2, 5000, 5000
    h = R.Hyper(input_dim=D, fc_dim=100, n_categories=92, state_dim=2, lowD_dim=10, n_arm=A)
    torch.manual_seed(546)
    m = U.build_model(h, None)
    m.train()
    x = R.synthetic_batch(B, D).to(U.DEV)
    eng = m._ensure(B)
    hyper, noise = m._hyper(1.0, False), N.make_noise(None, 7, 3)
    bn0, nbt0 = m._bn_flat.clone(), m._nbt.clone()
    ref = None
    for it in range(12):
        m._bn_flat.copy_(bn0)
        m._nbt.copy_(nbt0)
        buf = eng.train_step(hyper, noise, m._flat, m._bn_flat, m._nbt, x, 0, m._flat_grad, False, None, None, 1, 0.0)
        cur = (m._flat_grad.clone(), buf.clone())
        if ref is None:
            ref = cur
        else:
            assert torch.equal(cur[0], ref[0]) and torch.equal(cur[1], ref[1]), f"iteration {it} differs"
    assert torch.isfinite(ref[0]).all() and float(ref[0].abs().max()) > 0


def test_early_gradient_event_path_gives_the_same_gradients():
    """Data-parallel overlap (mmvae_set_early_grad_event): with the event set, train_step(do_adam=0) reduces the fc11
    gradients early on the side stream and the rest at the end -- the gradient buffer must be bit-identical to the
    single reduction, the event must have been recorded, and waiting on it must be enough to read the fc11 ranges."""
    U = _U()
    from distributed_vae_amd import _native as N
    g = G.load("mid_a2")
    h = G.hyper_of(g)
    B = G.batch_of(g)
    m = U.build_model(h, R.init_state_dict(h, int(g["seed"])))
    m.train()
    x = R.synthetic_batch(B, h.input_dim).to(U.DEV)
    eng = m._ensure(B)
    hyper, noise = m._hyper(1.0, False), N.make_noise(None, 3, 1)
    bn0, nbt0 = m._bn_flat.clone(), m._nbt.clone()

    def run(early):
        m._bn_flat.copy_(bn0)
        m._nbt.copy_(nbt0)
        m._flat_grad.zero_()
        eng.enable_early_grad_event(early)
        eng.train_step(hyper, noise, m._flat, m._bn_flat, m._nbt, x, 0, m._flat_grad, False, None, None, 1, 0.0)
        rec = eng.early_recorded()
        part = None
        if early:
            lay = m._layout
            per_arm, o26 = int(lay.per_arm), int(lay.offset[26])
            side = torch.cuda.Stream()
            side.wait_event(eng.early_event)
            with torch.cuda.stream(side):
                part = m._flat_grad[o26:per_arm].clone()          # arm 0's fc11 range, read behind the event only
            side.synchronize()
        torch.cuda.synchronize()
        return m._flat_grad.clone(), rec, part

    g0, rec0, _ = run(False)
    g1, rec1, part = run(True)
    assert not rec0 and rec1
    assert torch.equal(g0, g1)
    lay = m._layout
    assert torch.equal(part, g0[int(lay.offset[26]):int(lay.per_arm)]) and float(part.abs().max()) > 0
    eng.enable_early_grad_event(False)
    # the two ranges tile an arm's segment exactly: [0, o26) and [o26, per_arm)
    assert int(lay.offset[26]) % 4 == 0 and int(lay.offset[27]) > int(lay.offset[26])


@pytest.mark.parametrize("A", [3, 5])
def test_full_size_more_arms_fused_step_matches_api_path(A):
    """BASELINE.json's A = 3 and A = 5 configurations at the full batch / gene size: the fused train step (fused fc11
    kernel, two-round grids, side-stream overlaps) gives the loss vector and gradients of forward() / loss() /
    backward() -- the path the reference-generated fixtures pin at small sizes (tiny_a3, tiny_a5)."""
    U = _U()
    from distributed_vae_amd import _native as N
    B, D = 5000, 5000
    h = R.Hyper(input_dim=D, fc_dim=100, n_categories=92, state_dim=2, lowD_dim=10, n_arm=A)
    torch.manual_seed(546 + A)
    m = U.build_model(h, None)
    m.train()
    x = R.synthetic_batch(B, D, seed=A).to(U.DEV)
    eng = m._ensure(B)
    hyper, noise = m._hyper(1.0, False), N.make_noise(None, 11, A)
    bn0, nbt0 = m._bn_flat.clone(), m._nbt.clone()
    # API path: forward(need_grad) + loss + backward
    eng.forward(hyper, noise, m._flat, m._bn_flat, m._nbt, x, 0, None, True)
    l_api = eng.loss(hyper).clone()
    g_api = torch.zeros_like(m._flat_grad)     # zeros: the flat buffer's alignment gaps are never written (the model's own hold zeros)
    eng.backward(hyper, noise, m._flat, x, 0, g_api)
    # fused path from the same state
    m._bn_flat.copy_(bn0)
    m._nbt.copy_(nbt0)
    buf = eng.train_step(hyper, noise, m._flat, m._bn_flat, m._nbt, x, 0, m._flat_grad, False, None, None, 1, 0.0).clone()
    torch.cuda.synchronize()
    assert torch.isfinite(buf).all() and torch.isfinite(m._flat_grad).all()
    assert float((buf - l_api).abs().max()) <= 1e-5 * float(l_api.abs().max())
    err = (m._flat_grad - g_api).abs()
    scale = float(g_api.abs().max())
    assert float(err.max()) <= 5 * GRAD_TOL * scale                      # a few ReLU flips at 25 M decisions per arm
    assert float(torch.quantile(err[::97].float(), 0.9)) <= 1e-4 * scale


# ---------------------------------------------------------------------------------------------------
# forward(mask=...): the pruning-time forward (scope row f-4; nn_model.py:332-335)
# ---------------------------------------------------------------------------------------------------
def test_masked_forward_against_the_reference_fixture():
    """tests/golden/mask_a2.npz, generated by the REAL reference (oracle/gen_golden_mask.py): train mode -- forward outputs,
    loss and every gradient through ``loss.backward()`` -- then eval mode on the updated running statistics (what
    ``eval_model`` runs on a checkpoint with pruned categories)."""
    U = _U()
    g = G.load("mask_a2")
    h = G.hyper_of(g)
    A = h.n_arm
    mask = [int(v) for v in g["mask"]]
    off = sorted(set(range(h.n_categories)) - set(mask))
    m = U.build_model(h, G.state_dict_of(g))
    m.train()
    x = torch.from_numpy(g["x"]).to(U.DEV)
    xs = x.expand(A, -1, -1)
    m.set_explicit_noise(U.noise_to_device(G.noise_of(g)))
    out = m(xs, 1.0, 0.0, eval=False, mask=mask)
    lt = m.loss(out[0], [], [], xs, out[7], out[8], out[4], out[6], 0.0)
    m.zero_grad()
    lt[0].backward()
    torch.cuda.synchronize()
    names = {0: "x_rec", 3: "x_low", 4: "c", 5: "s_smp", 6: "c_smp", 7: "s_mean", 8: "s_logvar", 9: "c_prob"}
    for i, nm in names.items():
        assert G.rel_err(torch.stack(list(out[i])).cpu(), g["fwd/" + nm]) < FWD_TOL, nm
    for a in range(A):
        assert float(out[4][a][:, off].abs().max()) == 0.0                     # masked-out categories: exactly zero
        assert float((out[4][a].sum(1) - 1).abs().max()) < 1e-5
    assert abs(float(lt[0]) - float(g["loss/total"])) <= 1e-4 * abs(float(g["loss/total"]))
    for k, p in m.named_parameters():
        assert G.rel_err(p.grad.cpu(), g["grad/" + k]) < GRAD_TOL, k
    # eval mode (boolean mask form), running statistics after the one training step
    for k in g.files:
        if k.startswith("sd1/"):
            ref = torch.from_numpy(np.asarray(g[k]))
            if ref.dtype.is_floating_point:
                assert G.rel_err(m.state_dict()[k[4:]].cpu(), ref) < 1e-4, k
    m.eval()
    bmask = torch.zeros(h.n_categories, dtype=torch.bool)
    bmask[mask] = True
    m.set_explicit_noise(U.noise_to_device(G.noise_of(g, "noise_eval/")))
    with torch.no_grad():
        oe = m(xs, 1.0, 0.0, eval=True, mask=bmask)
    for i, nm in names.items():
        assert G.rel_err(torch.stack(list(oe[i])).cpu(), g["eval/" + nm]) < FWD_TOL, nm
    # the full mask and no mask are the same forward
    m.set_explicit_noise(U.noise_to_device(G.noise_of(g, "noise_eval/")))
    with torch.no_grad():
        o1 = m(xs, 1.0, 0.0, eval=True, mask=list(range(h.n_categories)))
    m.set_explicit_noise(U.noise_to_device(G.noise_of(g, "noise_eval/")))
    with torch.no_grad():
        o2 = m(xs, 1.0, 0.0, eval=True)
    for i in names:
        assert torch.equal(torch.stack(list(o1[i])), torch.stack(list(o2[i]))), i
    with pytest.raises(IndexError):
        m(xs, 1.0, 0.0, eval=True, mask=[0, h.n_categories])
