"""CPU: the index logic of the device-resident loaders (distributed-vae_amd/utils/dataloader.py) against the libraries
the reference calls: sklearn's train_test_split (mmidas/utils/dataloader.py:73-83) and torch's DistributedSampler
(:116-121).  No GPU: only index tensors are compared."""
import numpy as np
import pytest
import torch

import distributed_vae_amd  # noqa: F401
from distributed_vae_amd.utils import dataloader as DL


@pytest.mark.parametrize("n,frac,seed", [(100, 0.9, 546), (50000, 0.9, 546), (22365, 0.9, 0), (17, 0.5, 3), (10, 0.9, 1)])
def test_split_equals_sklearn(n, frac, seed):
    sk = pytest.importorskip("sklearn.model_selection")
    tt = int(frac * n)
    data = np.arange(n * 2).reshape(n, 2)
    tr, te, tr_i, te_i = sk.train_test_split(data, np.arange(n), train_size=tt, test_size=n - tt, random_state=seed)
    a, b, tr2, te2 = DL.data_gen(data, tt, seed)
    assert np.array_equal(tr_i, tr2) and np.array_equal(te_i, te2)
    assert np.array_equal(tr, a) and np.array_equal(te, b)
    assert len(set(tr2) | set(te2)) == n


def test_split_rejects_what_sklearn_rejects():
    with pytest.raises(ValueError):
        DL.split_indices(10, 10, 0)        # empty test set
    with pytest.raises(ValueError):
        DL.split_indices(10, 0, 0)


class _FakeCuda(torch.Tensor):
    pass


def _loader(n, bs, shuffle, drop_last, seed, ws=1, rk=0):
    """A DeviceLoader without a GPU: bypass the constructor's device check (index logic only)."""
    ld = DL.DeviceLoader.__new__(DL.DeviceLoader)
    ld.data = None
    ld.index = torch.arange(n)
    ld.batch_size, ld.shuffle, ld.drop_last = bs, shuffle, drop_last
    ld.seed, ld.world_size, ld.rank, ld.epoch, ld._auto_epoch = seed, ws, rk, 0, 0
    ld._pin, ld._pin_ev, ld._pin_k = None, None, 0
    return ld


@pytest.mark.parametrize("n,ws", [(103, 2), (100, 4), (5, 8), (45000, 8)])
@pytest.mark.parametrize("shuffle", [True, False])
def test_distributed_order_equals_torch_sampler(n, ws, shuffle):
    from torch.utils.data import DistributedSampler, TensorDataset
    ds = TensorDataset(torch.arange(n))
    for rank in range(min(ws, 3)):
        s = DistributedSampler(ds, num_replicas=ws, rank=rank, shuffle=shuffle, seed=546)
        ld = _loader(n, 16, shuffle, True, 546, ws, rank)
        for epoch in (0, 1, 5):
            s.set_epoch(epoch)
            ld.set_epoch(epoch)
            assert list(s) == ld.epoch_order().tolist()
            assert len(ld) == len(list(s)) // 16


def test_lengths_and_epoch_shuffling():
    ld = _loader(103, 10, True, True, 7)
    assert len(ld) == 10
    o0 = ld.epoch_order()
    assert sorted(o0.tolist()) == list(range(103))
    ld._auto_epoch = 1
    assert not torch.equal(o0, ld.epoch_order())
    ld2 = _loader(103, 10, False, False, 7)
    assert len(ld2) == 11 and ld2.epoch_order().tolist() == list(range(103))


def test_label_branch_keeps_the_reference_arithmetic():
    """:97-110 splits the WHOLE dataset with the label's train size and indexes the label's rows with the result; with
    more than one label that indexes past the label's rows, in the reference as here."""
    ref_err = None
    n = 40
    label = np.array([0] * 30 + [1] * 10)
    try:
        for ll in np.unique(label):
            indx = np.where(label == ll)[0]
            tr, te = DL.split_indices(n, int(0.9 * (label == ll).sum()), 546)
            indx[tr]
    except IndexError as e:
        ref_err = e
    assert ref_err is not None
