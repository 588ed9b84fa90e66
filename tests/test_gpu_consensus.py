"""GPU parity of the evaluation-label / consensus path against oracle/consensus.py
(numpy restatement of mmidas/_utils.py:79-129, pinned by the reference's own KATs) and the reference-generated
eval-mode golden vectors.  Integer work is compared bit-exactly; so are the fp64 normalisation and mean."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import consensus as OC
from oracle import restatement as R
from tests import golden_util as G

pytestmark = pytest.mark.gpu
KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "consensus_kat.json")))


def _U():
    from tests import gpu_util as U
    return U


def test_reference_kats_through_the_device_utils():
    """The reference's tests/test_utils.py cases, through the same-named functions of distributed_vae_amd._utils."""
    U = _U()
    from distributed_vae_amd import _utils as DU
    dev = U.DEV
    for k in KAT["compute_confmat"]:
        got = DU.compute_confmat(torch.tensor(k["labels1"], device=dev), torch.tensor(k["labels2"], device=dev))
        assert got.dtype == torch.float64 and np.array_equal(got.cpu().numpy(), np.array(k["expected"], dtype=float))
    for k in KAT["confmat_normalize"]:
        got = DU.confmat_normalize(torch.tensor(k["cm"], dtype=torch.float64, device=dev))
        assert np.array_equal(got.cpu().numpy(), np.array(k["expected"], dtype=float))
    for k in KAT["confmat_mean"]:
        assert float(DU.confmat_mean(torch.tensor(k["cm"], dtype=torch.float64, device=dev))) == k["expected"]
    for k in KAT["classify"]:
        assert DU.classify(torch.tensor(k["probs"], dtype=torch.float32, device=dev)).cpu().tolist() == k["expected"]


@pytest.mark.parametrize("A,K,n", [(2, 4, 50), (2, 92, 5000), (3, 92, 777), (5, 128, 20000), (2, 7, 1), (8, 33, 1000)])
def test_counts_normalisation_and_mean_bit_exact(A, K, n):
    from distributed_vae_amd import _native as N
    U = _U()
    rng = np.random.default_rng(A * 1000 + K)
    base = rng.integers(0, K, n)
    labels = np.stack([np.where(rng.random(n) < 0.6, base, rng.integers(0, max(K - 2, 1), n)) for _ in range(A)]).astype(np.int64)
    lab_d = torch.from_numpy(labels).to(torch.int32).to(U.DEV)
    counts = N.confmat_accumulate(lab_d, K)
    counts = N.confmat_accumulate(lab_d, K, counts)            # accumulation: every count doubles
    cons, norm = N.consensus(counts, want_norm=True)
    pair = 0
    vals = []
    for a in range(A):
        for b in range(a + 1, A):
            cm = OC.compute_confmat(labels[a], labels[b], K) * 2
            assert np.array_equal(counts[pair].cpu().numpy(), cm.astype(np.int64))
            nm = OC.confmat_normalize(cm)
            assert np.array_equal(norm[pair].cpu().numpy(), nm)          # bit-exact fp64
            v = OC.confmat_mean(nm)
            assert float(cons[pair]) == v, (float(cons[pair]), v)        # numpy's summation order reproduced
            vals.append(v)
            pair += 1
    assert pair == A * (A - 1) // 2
    assert float(cons.mean()) == pytest.approx(OC.epoch_consensus(labels, K)[1], rel=1e-15)


def test_classify_ties_and_shapes():
    from distributed_vae_amd import _native as N
    U = _U()
    p = torch.tensor([[0.4, 0.4, 0.2], [0.1, 0.45, 0.45], [0.0, 0.0, 0.0]], device=U.DEV)
    assert N.classify(p).cpu().tolist() == [0, 1, 0]
    g = torch.Generator(device="cpu").manual_seed(5)
    for C in (1, 63, 64, 65, 92, 128, 200):
        q = torch.rand(3, 37, C, generator=g)
        q[0, 5, C // 2] = 2.0
        got = N.classify(q.to(U.DEV))
        assert got.shape == (3, 37) and got.dtype == torch.int32
        assert np.array_equal(got.cpu().numpy(), OC.classify(q.numpy()))


@pytest.mark.parametrize("name", G.SMALL_CASES)
def test_eval_labels_against_reference_golden(name):
    """Labels of the encoder-only eval path == classify(c) of the reference's eval forward (tests/golden, generated
    by the real reference) wherever the reference's own top-2 margin exceeds the fp32 forward tolerance, and ==
    classify(c) of this engine's full eval forward everywhere (same kernels)."""
    U = _U()
    g = G.load(name)
    h = G.hyper_of(g)
    sd = G.state_dict_of(g)
    sd.update(G.state_dict_of(g, "eval/sd/"))
    m = U.build_model(h, sd)
    m.eval()
    x = torch.from_numpy(g["x"]).to(U.DEV)
    A = h.n_arm
    counts = torch.zeros(max(A * (A - 1) // 2, 1), h.n_categories, h.n_categories, dtype=torch.int64, device=U.DEV)
    labels = m.eval_labels(x.expand(A, -1, -1), 1.0, counts).cpu().numpy()
    c_ref = g["eval/fwd/c"]                                   # [A, B, C] from the reference
    top2 = np.sort(c_ref, axis=-1)[..., -2:]
    clear = (top2[..., 1] - top2[..., 0]) > 1e-3
    assert clear.mean() > 0.5
    assert np.array_equal(labels[clear], OC.classify(c_ref)[clear])
    with torch.no_grad():
        out, _, _ = U.run_step(m, x, G.noise_of(g, "eval/noise/"), eval_flag=True, backward=False)
    full = np.stack([OC.classify(t.cpu().numpy()) for t in out[4]])
    assert np.array_equal(labels, full)
    vals, mean = OC.epoch_consensus(labels.astype(np.int64), h.n_categories)
    from distributed_vae_amd._utils import consensus_from_counts
    got = consensus_from_counts(counts).cpu().numpy()
    assert np.array_equal(got, np.array(vals))
    for k, v in m.state_dict().items():                      # nothing touched
        if "running" in k or "num_batches" in k:
            assert torch.equal(v.cpu(), sd[k]), k


def test_trainer_consensus_over_a_loader_matches_oracle():
    """cpl_mixVAE.consensus over a ragged loader (last batch smaller) == the oracle's epoch_consensus on the labels of
    the oracle's own eval forward (fp32 CPU), allowing label flips only where c has no clear maximum."""
    U = _U()
    from distributed_vae_amd.cpl_mixvae import cpl_mixVAE
    A, D, C = 3, 64, 6
    h = R.Hyper(input_dim=D, fc_dim=16, n_categories=C, state_dim=2, lowD_dim=4, n_arm=A)
    sd = R.init_state_dict(h, 21)
    g = torch.Generator().manual_seed(2)
    for k in sd:                                              # non-trivial running statistics
        if "running_mean" in k:
            sd[k] = torch.randn(sd[k].shape, generator=g) * 0.1
        if "running_var" in k:
            sd[k] = torch.rand(sd[k].shape, generator=g) + 0.5
    t = cpl_mixVAE(saving_folder="", device=U.DEV, save_flag=False)
    t.init_model(n_categories=C, state_dim=2, input_dim=D, fc_dim=16, lowD_dim=4, x_drop=0.5, s_drop=0.0, n_arm=A,
                 temp=1.0, tau=0.005)
    t.model.load_state_dict(sd)
    X = R.synthetic_batch(150, D, seed=9)
    loader = [(X[i:i + 64],) for i in range(0, 150, 64)]      # 64, 64, 22
    got = t.consensus(loader)
    assert t.model.training                                    # mode restored
    # oracle labels from the oracle's eval forward of the whole set (running statistics: batch-size independent)
    noise = R.draw_noise(h, 150, seed=1)
    out = R.forward(sd, [X] * A, h, noise, training=False, eval_flag=True, update_running=False)
    c = np.stack([t_.numpy() for t_ in out[4]])
    top2 = np.sort(c, axis=-1)[..., -2:]
    assert ((top2[..., 1] - top2[..., 0]) > 1e-3).all(), "test case has near-ties; pick another seed"
    vals, mean = OC.epoch_consensus(OC.classify(c).astype(np.int64), C)
    assert got == mean


def test_validation_block_matches_oracle_including_the_batch_size_one_branch():
    """cpl_mixVAE.validate mirrors cpl_mixvae.py:665-775: with the reference's default test loader (batch_size 1) the
    whole test set is ONE batch taken from loader.dataset.tensors and the sums are divided by len(loader) (= the number
    of rows); with batch_size > 1 the loader is walked.  Both against the oracle's eval forward + loss + consensus."""
    U = _U()
    from distributed_vae_amd.cpl_mixvae import cpl_mixVAE
    from distributed_vae_amd.utils import dataloader as DL
    A, D, C = 3, 64, 6
    h = R.Hyper(input_dim=D, fc_dim=16, n_categories=C, state_dim=2, lowD_dim=4, n_arm=A)
    sd = R.init_state_dict(h, 5)
    g = torch.Generator().manual_seed(8)
    for k in sd:
        if "running_mean" in k:
            sd[k] = torch.randn(sd[k].shape, generator=g) * 0.1
        if "running_var" in k:
            sd[k] = torch.rand(sd[k].shape, generator=g) + 0.5
    t = cpl_mixVAE(saving_folder="", device=U.DEV, save_flag=False)
    t.init_model(n_categories=C, state_dim=2, input_dim=D, fc_dim=16, lowD_dim=4, x_drop=0.5, s_drop=0.0, n_arm=A)
    t.model.load_state_dict(sd)
    X = R.synthetic_batch(200, D, seed=4)
    _, te, _ = DL.get_loaders(X.numpy(), seed=546, batch_size=64, device=U.DEV)
    assert te.batch_size == 1 and len(te) == 20
    _, te_idx = DL.split_indices(200, 180, 546)
    Xt = X[torch.from_numpy(te_idx)]

    def oracle(batches, denom):
        tot = rec = 0.0
        labs = []
        for xb in batches:
            noise = R.draw_noise(h, xb.shape[0], seed=1)
            out = R.forward(sd, [xb] * A, h, noise, training=False, eval_flag=True, update_running=False)
            lt = R.loss(out, [xb] * A, h)
            tot += float(lt[0])
            rec += float(lt[1].double().sum()) / D
            labs.append(np.stack([OC.classify(c.numpy()) for c in out[4]]))
            top2 = np.sort(np.stack([c.numpy() for c in out[4]]), axis=-1)[..., -2:]
            assert ((top2[..., 1] - top2[..., 0]) > 1e-3).all(), "near-tie in the test case; pick another seed"
        cons = OC.epoch_consensus(np.concatenate(labs, axis=1).astype(np.int64), C)[1]
        return tot / denom, rec / denom / A, cons

    # the state sample enters the eval loss through the noise: feed the engine the oracle's draws
    got = []
    for loader, batches, denom in ((te, [Xt], 20), (None, [Xt[:8], Xt[8:16], Xt[16:]], 3)):
        if loader is None:
            loader = DL.DeviceLoader(te.data, torch.from_numpy(te_idx), 8, False, False)
            assert len(loader) == 3
        want = oracle(batches, denom)
        t.model.set_explicit_noise(None)
        # explicit noise per batch: the oracle draws with seed 1 at every batch size
        vals = []
        orig_forward = t.model.forward

        def fwd(xs, temp, prior_c=[], eval=False, mask=None):
            nz = R.draw_noise(h, xs.shape[1], seed=1)
            t.model.set_explicit_noise(U.noise_to_device(nz))
            return orig_forward(xs, temp, prior_c, eval=eval, mask=mask)
        t.model.forward = fwd
        try:
            res = t.validate(loader, full=True)
        finally:
            t.model.forward = orig_forward
            t.model.set_explicit_noise(None)
        assert abs(res[0] - want[0]) <= 1e-4 * abs(want[0]), (res, want)
        assert abs(res[1] - want[1]) <= 1e-4 * abs(want[1]), (res, want)
        assert res[2] == want[2], (res, want)
        assert abs(t.validate(loader) - res[1]) <= 1e-3 * abs(res[1])     # other state noise, same reconstruction scale
