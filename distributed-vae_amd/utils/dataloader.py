"""Device-resident drop-in for ``mmidas/utils/dataloader.py::get_loaders`` (:86-168).

The reference builds three ``torch.utils.data.DataLoader`` objects over host ``TensorDataset``s (2 worker processes,
pinned memory, one H2D copy per step).  The cells x genes matrix is 1 GB (50 k x 5000) to 10 GB (500 k x 5000) of fp32
and an MI355X has 288 GB of HBM: here the matrix is uploaded once, the loaders hold *index* tensors, and a batch is one
row gather on the device (``mmvae_gather_rows``).  What is reproduced exactly:

* the 90/10 split: ``sklearn.model_selection.train_test_split(dataset, arange(N), train_size, test_size,
  random_state=seed)`` (:73-83) = ``RandomState(seed).permutation(N)``, the first ``n_test`` entries are the test set,
  the next ``n_train`` the training set (sklearn's ShuffleSplit), restated with numpy -- no sklearn import here;
* the label-stratified variant (:97-110) with its arithmetic as written;
* ``drop_last=True`` / ``batch_size`` for training, ``batch_size=1, shuffle=False`` for the test loader,
  ``shuffle=False, drop_last=False`` for the all-data loader; every batch is ``(x, n)`` with ``n`` the float32 row
  indices, as ``TensorDataset(data, indices)`` yields them;
* ``DistributedSampler`` sharding (:116-121): the same index sequence per (seed, epoch, rank) as torch's sampler
  (seeded ``torch.randperm`` on the host, padded to a multiple of the world size, strided by rank).  The reference
  passes ``shuffle=True`` together with a sampler, which torch rejects; the sharded loader here simply works.

What differs on purpose: the shuffle order of the non-distributed training loader comes from a generator owned by the
loader (``seed``), not from torch's global RNG via DataLoader's ``RandomSampler``.
"""
from __future__ import annotations

import math
from typing import Optional, Sequence

import numpy as np
import torch

from .. import _native as N


def data_gen(dataset, train_size: int, seed):
    """dataloader.py:73-83.  Returns (train rows, test rows, train_ind, test_ind); ``dataset`` may be a numpy array
    or a tensor (rows are taken with the same indices either way)."""
    n = dataset.shape[0]
    train_ind, test_ind = split_indices(n, train_size, seed)
    return dataset[train_ind], dataset[test_ind], train_ind, test_ind


def split_indices(n: int, train_size: int, seed):
    """sklearn's ``train_test_split(..., train_size=int, test_size=n - train_size, random_state=seed)`` index logic."""
    test_size = n - train_size
    if train_size <= 0 or test_size < 0 or train_size > n:
        raise ValueError(f"train_size={train_size} should be in (0, {n}]")
    if test_size == 0:
        raise ValueError("test_size=0 should be a positive integer")     # sklearn rejects an empty test set
    rng = np.random.RandomState(seed) if not isinstance(seed, np.random.RandomState) else seed
    perm = rng.permutation(n)
    return perm[test_size:test_size + train_size], perm[:test_size]


class _Tensors:
    """``loader.dataset.tensors`` of the reference's TensorDataset (used at cpl_mixvae.py:617): (rows, indices)."""

    def __init__(self, loader):
        self._loader = loader

    @property
    def tensors(self):
        ld = self._loader
        return N.gather_rows(ld.data, ld.index), ld.index.to(torch.float32)

    def __len__(self):
        return int(self._loader.index.numel())


class DeviceLoader:
    """Iterable over ``(x [b, D] float32, n [b] float32)`` device batches of the resident matrix.

    index: int64 rows of ``data`` this loader serves, in their base order.  shuffle: a fresh permutation per epoch
    (``set_epoch`` or one per ``__iter__``).  world_size > 1: DistributedSampler semantics."""

    def __init__(self, data: torch.Tensor, index: torch.Tensor, batch_size: int, shuffle: bool, drop_last: bool,
                 seed: Optional[int] = None, world_size: int = 1, rank: int = 0):
        if data.device.type != "cuda":
            raise N.NativeError("DeviceLoader needs the matrix on the GPU (the data path has no CPU fallback)")
        assert data.dim() == 2 and data.dtype == torch.float32
        self.data = data
        self.index = index.to(device=data.device, dtype=torch.int64)
        if self.index.numel() and (int(self.index.min()) < 0 or int(self.index.max()) >= data.shape[0]):
            raise IndexError("row index out of range")
        self.batch_size, self.shuffle, self.drop_last = int(batch_size), bool(shuffle), bool(drop_last)
        self.seed = 0 if seed is None else int(seed)
        self.world_size, self.rank = int(world_size), int(rank)
        self.epoch = 0
        self._auto_epoch = 0
        self.dataset = _Tensors(self)

    # torch.utils.data.DistributedSampler.set_epoch
    def set_epoch(self, epoch: int):
        self.epoch = int(epoch)
        self._auto_epoch = None

    def _n_local(self) -> int:
        n = int(self.index.numel())
        if self.world_size > 1:
            return math.ceil(n / self.world_size)       # DistributedSampler(drop_last=False): padded
        return n

    def __len__(self) -> int:
        n = self._n_local()
        return n // self.batch_size if self.drop_last else math.ceil(n / self.batch_size)

    def epoch_order(self) -> torch.Tensor:
        """Positions into ``self.index`` this rank visits this epoch (host int64 tensor)."""
        n = int(self.index.numel())
        epoch = self.epoch if self._auto_epoch is None else self._auto_epoch
        if self.world_size > 1:
            # torch/utils/data/distributed.py: seeded randperm, pad by wrapping, stride by rank
            if self.shuffle:
                g = torch.Generator()
                g.manual_seed(self.seed + epoch)
                order = torch.randperm(n, generator=g)
            else:
                order = torch.arange(n)
            total = math.ceil(n / self.world_size) * self.world_size
            pad = total - n
            if pad > 0:
                reps = math.ceil(pad / max(n, 1))
                order = torch.cat([order, order.repeat(reps)[:pad]])
            return order[self.rank:total:self.world_size]
        if self.shuffle:
            g = torch.Generator()
            g.manual_seed(self.seed + epoch)
            return torch.randperm(n, generator=g)
        return torch.arange(n)

    def __iter__(self):
        order = self.epoch_order().to(self.index.device)
        if self._auto_epoch is not None:
            self._auto_epoch += 1
        rows = self.index[order]
        nb = len(self)
        for i in range(nb):
            r = rows[i * self.batch_size:(i + 1) * self.batch_size]
            yield N.gather_rows(self.data, r), r.to(torch.float32)


def get_loaders(dataset, label: Sequence = [], seed=None, batch_size=128, train_size=0.9, use_dist_sampler=False,
                world_size=1, rank=0, device=None):
    """dataloader.py:86-168 on a device-resident matrix.  ``dataset``: [N, D] numpy array or tensor (uploaded once if
    it is not on ``device`` already).  Returns ``(train_loader, test_loader, alldata_loader)``."""
    if device is None:
        device = dataset.device if isinstance(dataset, torch.Tensor) and dataset.device.type == "cuda" else "cuda"
    data = torch.as_tensor(dataset, dtype=torch.float32).to(device).contiguous()
    n = data.shape[0]
    if len(label) > 0:
        label = np.asarray(label)
        train_ind, test_ind = [], []
        for ll in np.unique(label):
            indx = np.where(label == ll)[0]
            tt_size = int(train_size * sum(label == ll))
            tr_sub, te_sub = split_indices(n, tt_size, seed)     # the reference splits the WHOLE set here (:102)
            train_ind.append(indx[tr_sub])                        # ... and indexes the label's rows with it (:103)
            test_ind.append(indx[te_sub])
        train_ind = np.concatenate(train_ind)
        test_ind = np.concatenate(test_ind)
    else:
        tt_size = int(train_size * n)
        train_ind, test_ind = split_indices(n, tt_size, seed)
    dist = world_size > 1 and use_dist_sampler
    ws, rk = (world_size, rank) if dist else (1, 0)
    tr = DeviceLoader(data, torch.from_numpy(np.ascontiguousarray(train_ind)), batch_size, True, True, seed, ws, rk)
    te = DeviceLoader(data, torch.from_numpy(np.ascontiguousarray(test_ind)), 1, dist, False, seed, ws, rk)
    al = DeviceLoader(data, torch.arange(n), batch_size, False, False, seed)
    return tr, te, al
