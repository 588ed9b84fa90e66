"""-m gpu: BASELINE.json's configurations at FULL size against the ORACLE (not against this engine's other path).

  cfg2  A = 2, B = 5000, D = 5000   (the benchmark line; also tests/test_gpu_parity.py::test_full_size_against_oracle)
  cfg4  A = 5, B = 5000, D = 5000   (ten arm pairs in the coupling terms)
  cfg5  A = 3, B = 5000, D = 5032   (the SmartSeq gene panel stand-in: D % 64 = 40, edge tiles of every big GEMM)
  and   A = 3, B = 5000, D = 5000

Per case, on one seeded input and explicit noise (oracle/restatement.py, the reference's arithmetic):
  * the FUSED train step (mmvae_train_step, the path the trainer and bench.py run) against the oracle evaluated in fp64,
    with the oracle's own fp32 evaluation as the noise floor.  The gate is FLIP-AWARE: the device's ReLU decision patterns
    (r1..r5, d6..d10 from the workspace, fc11's from dZ11) are compared with the fp64 oracle's pre-activations; a
    decision may differ only where the fp64 pre-activation is within fp32 rounding of zero, at most max(8, 4 A) hidden
    ones do, and the oracle is then evaluated on exactly the device's decisions (restatement.forward(relu_override=)),
    so that EVERY tensor -- bias gradients included -- is held to the tight gate: 90th percentile < max(3 x CPU, 1e-4),
    worst entry < 5e-3, at most max(3, 1 %) entries above a quarter of the gradient tolerance;
  * a stage-level pin of the dominant kernel's two outputs (k_fc11_zg: dZ11 and the gene-split slabs of d(d10)):
    every element whose fp64 pre-activation is more than MARGIN from the ReLU threshold must agree with the fp64 oracle
    to max(STAGE_TOL, 4 x the fp32 CPU oracle's own distance from fp64) of the tensor's largest magnitude -- no allowance for a fraction of bad elements (round 1 had a store
    hazard that corrupted 81 k of 50 M dZ11 elements and sat inside a 1 % allowance); the number of excluded elements
    is asserted small; the d(d10) GEMM is pinned exactly against the dZ11 this engine produced.
"""
import gc
import math

import pytest
import torch

from oracle import restatement as R
from tests import golden_util as G

pytestmark = pytest.mark.gpu

LOSS_TOL, GRAD_TOL = 1e-5, 1e-3
STAGE_TOL = 1e-5          # of the largest magnitude of the compared tensor
MARGIN = 1e-4             # |fp64 pre-activation| below this: the ReLU decision may legitimately differ in fp32
CASES = {
    "cfg2_a2_d5000": (2, 5000, 5000, "fp32"),
    # the same configuration on the OTHER fp32 engine (the fp32 matrix instruction: gemm_big.hip / gemm_fast.hip, the fallback
    # of every shape the split engine does not take), so that both engines are verified by the default `pytest -m gpu`
    "cfg2_a2_d5000_fp32_mfma": (2, 5000, 5000, "fp32_mfma"),
    "a3_d5000": (3, 5000, 5000, "fp32"),
    "cfg4_a5_d5000": (5, 5000, 5000, "fp32"),
    "cfg5_a3_d5032": (3, 5000, 5032, "fp32"),
}


def _U():
    from tests import gpu_util as U
    return U


@pytest.fixture(scope="module", params=list(CASES))
def case(request):
    """Oracle results (fp32 autograd, fp64 autograd, fp64 stage tensors) and one fused step on the GPU."""
    U = _U()
    A, B, D, gemm = CASES[request.param]
    h = R.Hyper(input_dim=D, n_arm=A)
    sd = R.init_state_dict(h, 546 + A)
    x = R.synthetic_batch(B, D, seed=546 + D)
    noise = R.draw_noise(h, B, seed=7 + A)
    # --- one fused train step (no Adam) on the GPU from the same state: gradients, stage tensors, ReLU decision patterns
    m = U.build_model(h, sd)
    m.train()
    m.gemm_dtype = gemm
    m.set_explicit_noise(U.noise_to_device(noise))
    xd = x.to(U.DEV)
    buf = m.fused_train_step(xd.expand(A, -1, -1), 1.0, None, do_adam=False).clone()
    torch.cuda.synchronize()
    grads = {k: gv.detach().cpu().clone() for (k, _), gv in zip(m.named_parameters(), m._grad_views)}
    eng = m._engine
    ns = eng.splits()[4]
    stage = {
        "dz11": eng.ws_view("dz11", D).cpu(),
        "gd10": eng.ws_raw("gd10_slab", ns * A * B * h.fc_dim).view(ns, A, B, h.fc_dim).cpu(),
        "d10": eng.ws_view("d10", h.fc_dim).cpu(),
    }
    patterns = U.device_relu_patterns(eng, h)
    del m, eng, xd
    gc.collect()
    torch.cuda.empty_cache()
    # --- oracle on the device's ReLU decisions (tests/gpu_util.py::flip_aware_oracle): fp32 (noise floor) and fp64
    fo = U.flip_aware_oracle(h, sd, x, noise, patterns)
    del patterns
    lt_32, g_32, lt_64, g_64, saved = fo["lt_32"], fo["g_32"], fo["lt_64"], fo["g_64"], fo["saved64"]
    sd64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    am1 = float(max(A - 1, 1))
    d10 = torch.stack([s_["d10"] for s_ in saved])                                # [A,B,H] fp64 (the un-forced forward)
    z11, gz11, gd = [], [], []
    for a in range(A):
        z = saved[a]["zx_rec"]                                                    # fc11 pre-activation (nn_model.py:286)
        gz = am1 * (torch.relu(z) - x.double()) / B * (z > 0)                     # d total / d z11 (nn_model.py:544, :587)
        z11.append(z.float())
        gz11.append(gz)
        gd.append(gz @ sd64[f"fc11.{a}.weight"])
    del saved
    fo["saved64"] = None
    # the fp32 oracle's own stage values: its distance from fp64 is the noise floor of ANY fp32 evaluation (BatchNorm
    # divides by the batch deviation of every unit, a nearly dead unit amplifies rounding noise by 1/sqrt(var + 1e-8))
    d10_32 = fo.pop("d10_32")
    floor_d10 = float((d10_32.double() - d10).abs().max() / d10.abs().max())
    floor_dz = 0.0
    for a in range(A):
        z32 = d10_32[a] @ sd[f"fc11.{a}.weight"].t() + sd[f"fc11.{a}.bias"]
        gz32 = am1 * (torch.relu(z32) - x) / B * (z32 > 0)
        far = z11[a].abs() > MARGIN
        floor_dz = max(floor_dz, float(((gz32.double() - gz11[a]).abs() * far).max() / gz11[a].abs().max()))
        del z32, gz32, far
    del d10_32
    yield dict(A=A, B=B, D=D, h=h, sd=sd, lt_32=lt_32, lt_64=lt_64, g_32=g_32, g_64=g_64, buf=buf.cpu(), grads=grads,
               stage=stage, z11=z11, x=x, floor_d10=floor_d10, floor_dz=floor_dz, gz11=gz11, gd=gd, d10=d10, fo=fo)
    gc.collect()


def test_fused_step_against_oracle(case):
    """Loss vector and every parameter gradient of the fused step against the fp64 oracle."""
    c = case
    A = c["A"]
    U = _U()
    from distributed_vae_amd import _native as N
    lt = c["lt_64"]
    lt = [v.detach() if torch.is_tensor(v) else v for v in lt]
    want = [float(lt[0]), float(lt[2]), float(lt[3]), float(lt[4]), float(lt[5])]
    want += [float(v) for v in lt[1]] + [float(v) for v in lt[6]] + [float(v) for v in lt[8]]
    got = c["buf"].double().tolist()
    tol = [LOSS_TOL, LOSS_TOL, 1e-4, LOSS_TOL, 1e-4] + [LOSS_TOL] * A + [1e-4] * A + [LOSS_TOL] * A
    assert len(got) == N.LOSS_REC0 + 3 * A
    for i, (g_, w_, t_) in enumerate(zip(got, want, tol)):
        assert abs(g_ - w_) <= t_ * abs(w_) + 1e-7, (i, g_, w_)
    # every tensor, bias gradients included, to the round-1 gate -- against the oracle evaluated on the decisions the
    # device took (at most a handful differ from the fp64 oracle's own, each at a pre-activation within rounding of zero:
    # asserted in the fixture)
    U.assert_gradients_tight(c["grads"], c["fo"], GRAD_TOL)


def test_dominant_kernel_outputs_against_oracle(case):
    """k_fc11_zg's outputs element for element: dZ11 [A,B,D] and the summed d(d10) slabs [A,B,H]."""
    c = case
    A, B, D = c["A"], c["B"], c["D"]
    st = c["stage"]
    # the kernel's input first: a wrong d10 would show up as a wrong dZ11
    sc = float(c["d10"].abs().max())
    tol_d10 = max(STAGE_TOL, 4.0 * c["floor_d10"])
    tol_dz = max(STAGE_TOL, 4.0 * c["floor_dz"])
    assert tol_d10 < 2e-4 and tol_dz < 2e-4, (c["floor_d10"], c["floor_dz"])      # the floor itself stays small
    e_d10 = float((st["d10"].double() - c["d10"]).abs().max()) / sc
    assert e_d10 <= tol_d10, (e_d10, c["floor_d10"])
    n_excl = 0
    for a in range(A):
        ref = c["gz11"][a]
        sc = float(ref.abs().max())
        err = (st["dz11"][a].double() - ref).abs()
        near = c["z11"][a].abs() <= MARGIN               # decisions fp32 may take either way
        n_excl += int(near.sum())
        bad = (err > tol_dz * sc) & ~near
        assert not bool(bad.any()), (a, int(bad.sum()), float(err[~near].max() / sc), c["floor_dz"])
        # an excluded element is either right or on the other side of the ReLU (0 <-> coef (0 - x)): nothing else
        flipped = near & (err > tol_dz * sc)
        if bool(flipped.any()):
            coef = max(A - 1, 1) / B
            alt = torch.where(c["z11"][a] > 0, torch.zeros_like(ref), -coef * c["x"].double())
            assert float((st["dz11"][a].double() - alt)[flipped].abs().max()) <= tol_dz * sc + MARGIN * coef
    assert n_excl <= 2e-3 * A * B * D, n_excl            # measured: a few 1e-4 of the elements
    # d(d10) = dZ11 W11: (1) the GEMM itself, exactly, on the dZ11 this engine produced; (2) against the oracle
    got = st["gd10"].double().sum(0)                      # fixed-order sum of the gene-split slabs (decoder prologue)
    for a in range(A):
        w = c["sd"][f"fc11.{a}.weight"].double()
        own = st["dz11"][a].double() @ w
        sc = float(own.abs().max())
        assert float((got[a] - own).abs().max()) <= STAGE_TOL * sc, a
        e = ((got[a] - c["gd"][a]).abs() / float(c["gd"][a].abs().max())).flatten()
        p90 = float(e.kthvalue(int(0.9 * e.numel())).values)
        # (the maximum: one flipped ReLU decision of a gene with a large x moves a whole cell's row; the GEMM itself is
        # pinned exactly above, the flips are classified element by element in the dZ11 check)
        assert p90 <= tol_dz and float(e.max()) < 2e-2, (a, p90, float(e.max()))


def test_loss_rec_counts_every_threshold_decision(case):
    """loss_rec = 0.5 sum (x_rec - x)^2 / B + 50 * mismatch fraction (nn_model.py:544-546): the mismatch count is an
    integer over B * D decisions; a wrong count of n elements moves loss_rec by 50 n / (B D).  Bound: the number of
    elements whose fp64 |x_rec - 0.1| is within fp32 rounding cannot explain more than a 1e-6 relative difference."""
    c = case
    A = c["A"]
    rec_gpu = c["buf"][5:5 + A].double()
    rec_ref = torch.as_tensor([float(v) for v in c["lt_64"][1]]).double()
    assert float(((rec_gpu - rec_ref).abs() / rec_ref.abs()).max()) < 2e-6
    assert math.isfinite(float(c["buf"][0]))


def test_full_size_adam_trajectory():
    """cfg2 (A = 2, B = D = 5000) through FIVE consecutive fused steps WITH Adam on the device -- a fresh batch and fresh
    explicit noise per step, as the training loop draws them (cpl_mixvae.py:434-463); parameters, both Adam moments, the
    step count and the BatchNorm running buffers evolve on the device only.  What one step without Adam (the tests above)
    cannot see: the optimiser with non-trivial moments, its interaction with the batch-sum accumulators, the running
    buffers' momentum updates and the split engine's dropped terms over consecutive steps at the benchmark's size.

    The oracle is TEACHER-FORCED: before every step it takes the device's state (parameters, moments, running statistics,
    read back) and takes the same step in fp64; the step's loss vector, the parameters behind it, the new moments and the
    new running statistics are compared.  A free-running comparison cannot be gated at this size: the coupling term
    (inv_var ~ 1e4, tau = 0.005) amplifies fp32 rounding from step to step -- the CPU oracle's OWN fp32 trajectory leaves
    its fp64 one by 6e-5 / 1.2e-3 / 2.8e-3 / 3.6e-3 at steps 2 .. 5 in the build container and by 0.8e-3 at step 5 on the GPU
    box's cores (another thread count, another summation order), the device's by 4.5e-3 -- while every single step from a
    common state agrees to 1e-5."""
    U = _U()
    from distributed_vae_amd.cpl_mixvae import FusedAdam
    A, B, D, steps, lr = 2, 5000, 5000, 5, 1e-3
    h = R.Hyper(input_dim=D, n_arm=A)
    m = U.build_model(h, R.init_state_dict(h, 546))
    m.train()
    opt = FusedAdam(m, lr=lr)
    names = [k for k, _ in m.named_parameters()]
    keys = R.param_keys(h)
    assert set(names) == set(keys)

    def adam_state():
        st = opt.state_dict()["state"]
        if not st:
            return None
        return {"t": int(float(st[0]["step"])), "m": {k: st[i]["exp_avg"].cpu().double() for i, k in enumerate(names)},
                "v": {k: st[i]["exp_avg_sq"].cpu().double() for i, k in enumerate(names)}}

    for s in range(steps):
        sd64 = {k: (v.detach().cpu().double() if v.is_floating_point() else v.detach().cpu().clone()) for k, v in m.state_dict().items()}
        st64 = adam_state()
        assert (st64 is None) == (s == 0) and (st64 is None or st64["t"] == s)
        x = R.synthetic_batch(B, D, seed=546 + 300 + s)
        nz = R.draw_noise(h, B, seed=3000 + s)
        m.set_explicit_noise(U.noise_to_device(nz))
        buf = m.fused_train_step(x.to(U.DEV).expand(A, -1, -1), 1.0, opt, do_adam=True).cpu().double()
        torch.cuda.synchronize()
        n64 = {k: [t.double() if t.is_floating_point() else t for t in v] for k, v in nz.items()}
        hist, st_new = R.train_steps(sd64, [x.double()], h, [n64], lr=lr, opt_state=st64)
        lt = hist[0]
        want = [float(lt[0]), float(lt[2])] + [float(v) for v in lt[1]]
        got = [float(buf[0]), float(buf[1])] + [float(buf[5 + a]) for a in range(A)]
        for i, name in enumerate(["total", "joint"] + [f"rec{a}" for a in range(A)]):
            tol = 1e-5 if name.startswith("rec") else 1e-4
            assert abs(got[i] - want[i]) <= tol * abs(want[i]), (s, name, got[i], want[i])
        dev_sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
        new = adam_state()
        assert new["t"] == s + 1 == st_new["t"]
        for k in keys:
            diff = (dev_sd[k].double() - sd64[k]).abs()
            # Adam turns rounding noise on near-zero gradients into O(lr) steps (at t = 1 the update is lr x sign(g): a gradient
            # entry that is zero up to rounding may step the other way): the worst entry of ONE step is bounded by 2 lr, the
            # typical entry tightly (the rule of tests/test_gpu_parity.py::test_golden_adam_trajectory)
            assert float(diff.max()) < 2.06 * lr and float(diff.median()) < 2e-5, (s, k, float(diff.max()), float(diff.median()))
            for name, dv, ov in (("m", new["m"][k], st_new["m"][k]), ("v", new["v"][k], st_new["v"][k])):
                scale = float(ov.abs().max()) + 1e-30
                assert float((dv - ov).abs().max()) <= GRAD_TOL * scale, (s, k, name)
        for k, v in dev_sd.items():
            if "running" in k:
                ref = sd64[k]
                assert float((v.double() - ref).abs().max()) <= 1e-5 * max(1.0, float(ref.abs().max())), (s, k)
            elif "num_batches" in k:       # (batch_s is never applied, nn_model.py: its buffers exist and never change)
                assert int(v.reshape(-1)[0]) == (0 if k.startswith("batch_s") else s + 1), k
    del m, opt
    gc.collect()
    torch.cuda.empty_cache()
