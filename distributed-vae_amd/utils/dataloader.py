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
* ``DistributedSampler`` sharding (:116-121): pad to a multiple of the world size by wrapping, stride by rank, one
  seeded permutation per epoch shared by all ranks.  With ``host_order = True`` the sequence is torch's sampler's,
  index for index; by default the permutation is drawn by the device generator (same sharding, no upload per epoch).
  The reference passes ``shuffle=True`` together with a sampler, which torch rejects; the sharded loader here works.

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
                 seed: Optional[int] = None, world_size: int = 1, rank: int = 0, ring: int = 4):
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
        # Batches are gathered into a ring of `ring` persistent buffers: a fresh 100 MB tensor per batch makes the caching
        # allocator grow and hipMalloc in the middle of an epoch as soon as two streams are involved (measured: epochs
        # of 70-90 ms instead of 11).  A yielded batch stays valid until `ring` - 1 more batches have been drawn --
        # training loops consume a batch at once; ring = 0 returns a new tensor per batch, as a DataLoader does.
        self.ring = int(ring)
        self._bufs = None
        self._slot = 0
        self.prefetch_order = True       # see _epoch_rows
        self._order_stream, self._ahead = None, None
        self._data16 = None
        self._pin, self._pin_ev, self._pin_k, self._copy_stream = None, None, 0, None
        self.host_order = False      # True: torch's CPU permutation (DistributedSampler's exact sequence), uploaded per epoch

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

    def _base_order(self, epoch: int, device) -> torch.Tensor:
        n = int(self.index.numel())
        if not self.shuffle:
            return torch.arange(n, device=device)
        g = torch.Generator(device=device)
        g.manual_seed(self.seed + epoch)
        return torch.randperm(n, generator=g, device=device)

    def _shard(self, order: torch.Tensor) -> torch.Tensor:
        """torch/utils/data/distributed.py: pad by wrapping to a multiple of the world size, stride by rank."""
        if self.world_size <= 1:
            return order
        n = order.numel()
        total = math.ceil(n / self.world_size) * self.world_size
        pad = total - n
        if pad > 0:
            reps = math.ceil(pad / max(n, 1))
            order = torch.cat([order, order.repeat(reps)[:pad]])
        return order[self.rank:total:self.world_size]

    def _epoch(self) -> int:
        return self.epoch if self._auto_epoch is None else self._auto_epoch

    def epoch_order(self) -> torch.Tensor:
        """Positions into ``self.index`` this rank visits this epoch, computed on the HOST: for a sharded loader this is
        the exact sequence of ``torch.utils.data.DistributedSampler(seed=seed)`` (tests/test_dataloader_cpu.py)."""
        return self._shard(self._base_order(self._epoch(), "cpu"))

    def epoch_order_device(self) -> torch.Tensor:
        """The same logic with the permutation drawn on the device (seeded device generator: identical on every rank):
        no host-to-device copy per epoch.  Such a copy (400 KB, even pinned and on a stream of its own) stalled for
        50-90 ms every few epochs while train steps were queued -- an epoch is 11 ms."""
        return self._shard(self._base_order(self._epoch(), self.index.device))

    def _upload(self, order: torch.Tensor) -> torch.Tensor:
        """Host permutation -> device through pinned staging and an asynchronous copy.  A plain ``.to(device)`` of
        pageable memory while train steps are queued stalled for 25-75 ms now and then (measured: epochs of 35-85 ms
        instead of 10.6)."""
        n = order.numel()
        if self._pin is None or self._pin[0].numel() < n:
            self._pin = [torch.empty(max(n, 1), dtype=torch.int64).pin_memory() for _ in range(2)]
            self._pin_ev = [None, None]
        k = self._pin_k
        self._pin_k ^= 1
        if self._pin_ev[k] is not None:
            self._pin_ev[k].synchronize()                       # the copy issued two epochs ago has long finished
        self._pin[k][:n].copy_(order)
        dev = torch.empty(n, dtype=torch.int64, device=self.index.device)
        if self._copy_stream is None:
            self._copy_stream = torch.cuda.Stream(device=self.index.device)
        cur = torch.cuda.current_stream(self.index.device)
        self._copy_stream.wait_stream(cur)                      # `dev` was just handed out by the allocator on `cur`
        with torch.cuda.stream(self._copy_stream):              # a stream of its own: not behind the queued train steps
            dev.copy_(self._pin[k][:n], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self._copy_stream)
        cur.wait_event(ev)
        dev.record_stream(self._copy_stream)
        self._pin_ev[k] = ev
        return dev

    def row_chunks(self, rows: int):
        """The loader's rows in base order (no shuffle, no sharding, nothing consumed), gathered ``rows`` at a time:
        ``loader.dataset.tensors[0]`` piece by piece, for per-cell evaluation over sets too large to gather at once."""
        n = int(self.index.numel())
        for i in range(0, n, int(rows)):
            yield N.gather_rows(self.data, self.index[i:i + int(rows)])

    def data_bf16(self):
        """The bf16 copy of ``self.data`` for the bf16 configuration's row-indexed step (made on first use, once per data
        set: ``_native.to_bf16``), or None where that step cannot use it (gene count or row pitch not a multiple of 8)."""
        if self.data.device.type != "cuda" or self.data.shape[1] % 8 != 0 or self.data.stride(0) % 8 != 0:
            return None
        ver = self.data._version                      # an in-place edit of the matrix makes the copy stale
        if self._data16 is None or getattr(self, "_data16_version", None) != ver:
            self._data16 = N.to_bf16(self.data)
            self._data16_version = ver
        return self._data16

    def data_planes(self, n_planes: int):
        """``self.data`` as the planes x planes GEMM engine's tiled slice planes (``_native.tp_planes``; made on first use, once
        per data set and plane count) for the augmenter's row-indexed forward, or None where the library does not offer it."""
        if self.data.device.type != "cuda":
            return None
        ver = self.data._version                      # an in-place edit of the matrix makes the planes stale
        cache = getattr(self, "_planes", None)
        if cache is None or cache[0] != (ver, n_planes):
            self._planes = ((ver, n_planes), N.tp_planes(self.data, n_planes))
        return self._planes[1]

    def unread_epoch(self):
        """Hand an epoch back unconsumed: the caller took this epoch's rows (``iter_rows``) but cannot use them and will draw
        the same epoch again through another iterator (the trainer's fallback from row-indexed steps to gathered batches)."""
        if self._auto_epoch is not None:
            self._auto_epoch -= 1

    def _epoch_rows(self) -> torch.Tensor:
        """Row indices (into ``self.data``) of this rank's epoch, in visiting order, and the epoch bookkeeping of one pass.
        With the permutation drawn on the device the NEXT epoch's rows are prepared right away on a stream of their own (the
        order is a function of seed + epoch): drawn at the head of its epoch, the permutation's five small launches sit
        between two train steps -- 70 us per epoch at the benchmark shape, 7 us per step of a ten-step epoch."""
        e = self._epoch()
        dev = self.index.device
        ahead, self._ahead = getattr(self, "_ahead", None), None
        if self.host_order:
            rows = self.index[self._upload(self.epoch_order())]
        elif ahead is not None and ahead[0] == e:
            rows = ahead[1]
            cur = torch.cuda.current_stream(dev)
            cur.wait_event(ahead[2])
            rows.record_stream(cur)
        else:
            rows = self.index[self.epoch_order_device()]
        if self._auto_epoch is not None:
            self._auto_epoch += 1
        if not self.host_order and self.shuffle and dev.type == "cuda" and self.prefetch_order:
            if self._order_stream is None:
                self._order_stream = torch.cuda.Stream(device=dev)
                self._order_stream.wait_stream(torch.cuda.current_stream(dev))     # self.index may just have been written
            # (no wait on the current stream after that: the host runs up to an epoch ahead of the device, and a prefetch
            # ordered behind everything queued so far would run at the very end of the epoch it was meant to overlap)
            with torch.cuda.stream(self._order_stream):
                nxt = self.index[self._shard(self._base_order(e + 1, dev))]
                ev = torch.cuda.Event()
                ev.record(self._order_stream)
            self._ahead = (e + 1, nxt, ev)
        return rows

    def iter_rows(self):
        """One epoch as ROW INDICES instead of gathered batches: yields int64 device tensors ``rows`` (the batch is
        ``self.data[rows]``), same order, sharding, ``drop_last`` and epoch bookkeeping as ``__iter__``.  For consumers that
        read the resident matrix through the indices (``mixVAE_model.fused_train_step_rows``): nothing is copied."""
        rows = self._epoch_rows()
        for i in range(len(self)):
            yield rows[i * self.batch_size:(i + 1) * self.batch_size]

    def __iter__(self):
        rows = self._epoch_rows()
        nb = len(self)
        if self.ring > 0 and (self._bufs is None or self._bufs[0][0].shape[0] != self.batch_size):
            self._bufs = [(torch.empty(self.batch_size, self.data.shape[1], dtype=torch.float32, device=self.data.device),
                           torch.empty(self.batch_size, dtype=torch.float32, device=self.data.device))
                          for _ in range(self.ring)]
        for i in range(nb):
            r = rows[i * self.batch_size:(i + 1) * self.batch_size]
            if self.ring > 0:
                xb, ib = self._bufs[self._slot]
                self._slot = (self._slot + 1) % self.ring
                xb, ib = xb[:r.numel()], ib[:r.numel()]
                N.gather_rows(self.data, r, xb)
                ib.copy_(r)
                yield xb, ib
            else:
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
