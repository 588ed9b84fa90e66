"""GPU: the device-resident data path: row gather bit-exact against numpy indexing (the
oracle of a byte copy), loaders yield the rows the index logic says, and the trainer runs an epoch from them."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _N():
    import distributed_vae_amd  # noqa: F401
    from distributed_vae_amd import _native as N
    return N


@pytest.mark.parametrize("n_rows,D,n", [(1000, 5000, 5000), (37, 52, 100), (64, 1006, 64), (5, 3, 9), (300, 257, 1)])
def test_gather_rows_bit_exact(n_rows, D, n):
    N = _N()
    rng = np.random.default_rng(D)
    data = rng.standard_normal((n_rows, D)).astype(np.float32)
    idx = rng.integers(0, n_rows, n)
    got = N.gather_rows(torch.from_numpy(data).to(DEV), torch.from_numpy(idx).to(DEV))
    assert np.array_equal(got.cpu().numpy(), data[idx])
    # strided source rows (a column window of a wider matrix)
    if D > 8:
        wide = torch.from_numpy(data).to(DEV)
        view = wide[:, 4:D - 3]
        got = N.gather_rows(view, torch.from_numpy(idx).to(DEV))
        assert np.array_equal(got.cpu().numpy(), data[idx][:, 4:D - 3])


def test_loaders_serve_the_split_without_loss():
    from distributed_vae_amd.utils import dataloader as DL
    n, D, bs = 1003, 64, 100
    data = np.arange(n * D, dtype=np.float32).reshape(n, D)
    tr, te, al = DL.get_loaders(data, seed=546, batch_size=bs, train_size=0.9, device=DEV)
    tr_i, te_i = DL.split_indices(n, int(0.9 * n), 546)
    assert len(tr) == len(tr_i) // bs and len(te) == len(te_i) and len(al) == -(-n // bs)
    seen = []
    for x, idx in tr:
        assert x.shape == (bs, D) and idx.dtype == torch.float32 and x.device.type == "cuda"
        ii = idx.cpu().numpy().astype(np.int64)
        assert np.array_equal(x.cpu().numpy(), data[ii])
        seen.append(ii)
    seen = np.concatenate(seen)
    assert len(np.unique(seen)) == len(seen) == len(tr) * bs and set(seen) <= set(tr_i)     # drop_last, no repeats
    first = next(iter(tr))[1]
    assert not torch.equal(first, torch.from_numpy(seen[:bs]).float().to(DEV))              # next epoch: new order
    got = np.concatenate([i.cpu().numpy() for _, i in te]).astype(np.int64)
    assert np.array_equal(got, te_i)                                                       # batch_size 1, in order
    rows = np.concatenate([x.cpu().numpy() for x, _ in al])
    assert np.array_equal(rows, data)                                                      # all data, in order
    xt, it = tr.dataset.tensors
    assert np.array_equal(xt.cpu().numpy(), data[tr_i]) and np.array_equal(it.cpu().numpy(), tr_i.astype(np.float32))


def test_distributed_shards_partition_the_training_set():
    from distributed_vae_amd.utils import dataloader as DL
    n, D = 400, 16
    data = torch.rand(n, D, device=DEV)
    per_rank = []
    for rank in range(4):
        tr, _, _ = DL.get_loaders(data, seed=1, batch_size=10, use_dist_sampler=True, world_size=4, rank=rank)
        tr.set_epoch(3)
        per_rank.append(np.concatenate([i.cpu().numpy() for _, i in tr]))
    allv = np.concatenate(per_rank)
    assert len(np.unique(allv)) == len(allv) == 360                                        # 90 rows per rank, disjoint


def test_device_and_host_orders_share_the_sharding_logic():
    from distributed_vae_amd.utils import dataloader as DL
    data = torch.rand(103, 8, device=DEV)
    for ws in (1, 4):
        parts = []
        for rank in range(ws):
            ld = DL.DeviceLoader(data, torch.arange(103), 10, True, True, seed=3, world_size=ws, rank=rank)
            ld.set_epoch(2)
            od = ld.epoch_order_device().cpu()
            assert od.numel() == ld.epoch_order().numel()
            parts.append(od)
            ld2 = DL.DeviceLoader(data, torch.arange(103), 10, True, True, seed=3, world_size=ws, rank=rank)
            ld2.set_epoch(2)
            assert torch.equal(ld2.epoch_order_device().cpu(), od)              # deterministic in (seed, epoch, rank)
            ld2.host_order = True
            got = torch.cat([i.clone() for _, i in ld2]).cpu().long()     # ring buffers: copy before the next draw
            assert torch.equal(got, ld2.index.cpu()[ld2.epoch_order()][:got.numel()])
        allv = torch.cat(parts)
        assert set(allv.tolist()) == set(range(103))                            # every row served, padding by wrapping


def test_trainer_epoch_from_device_loaders():
    from distributed_vae_amd.cpl_mixvae import cpl_mixVAE
    from distributed_vae_amd.utils import dataloader as DL
    from oracle import restatement as R
    D = 64
    X = R.synthetic_batch(330, D, seed=3)
    tr, te, al = DL.get_loaders(X.numpy(), seed=546, batch_size=64, device=DEV)
    t = cpl_mixVAE(saving_folder="", device=DEV, save_flag=False)
    t.init_model(n_categories=6, state_dim=2, input_dim=D, fc_dim=16, lowD_dim=4, x_drop=0.5, s_drop=0.0, n_arm=2)
    te.batch_size = 16                                         # validation in batches (the reference's 1 also works)
    hist = t.train(tr, te, n_epoch=2)
    assert len(hist["losses"]) == 2 and np.isfinite(hist["losses"]).all()
    assert np.isfinite(hist["validation_loss"]).all() and 0.0 <= hist["consensus_train"][-1] <= 1.0


# ---------------------------------------------------------------------------------------------------
# The batch as rows of the resident matrix (mmvae_train_step_rows): never materialised, bit-identical to gather + step
# ---------------------------------------------------------------------------------------------------
def _rows_case(A, B, D, H, n_rows, seed, dtype="fp32"):
    from oracle import restatement as R
    from tests import gpu_util as U
    from distributed_vae_amd.cpl_mixvae import FusedAdam
    h = R.Hyper(input_dim=D, fc_dim=H, n_categories=12, state_dim=2, lowD_dim=6, n_arm=A)
    sd = R.init_state_dict(h, seed)
    data = R.synthetic_batch(n_rows, D, seed=seed + 1).to(U.DEV)
    g = torch.Generator().manual_seed(seed + 2)
    rows = torch.randint(0, n_rows, (B,), generator=g)
    rows[:3] = torch.tensor([n_rows - 1, 0, n_rows - 1])           # repeated rows, both ends
    res = []
    for indexed in (False, True):
        m = U.build_model(h, sd)
        m.train()
        m.gemm_dtype = dtype
        opt = FusedAdam(m, lr=1e-3)
        bufs = []
        for s in range(2):
            m.set_explicit_noise(U.noise_to_device(R.draw_noise(h, B, seed=seed + 10 + s)))
            r = torch.roll(rows, s).to(U.DEV)
            if indexed:
                bufs.append(m.fused_train_step_rows(data, r, 1.0, opt, do_adam=True).clone())
            else:
                x = data[r].contiguous()
                bufs.append(m.fused_train_step(x.expand(A, -1, -1), 1.0, opt, do_adam=True).clone())
        torch.cuda.synchronize()
        res.append((torch.stack(bufs).cpu(), m.flat_parameters().detach().cpu().clone(), m._flat_grad.detach().cpu().clone(),
                    m._bn_flat.detach().cpu().clone(), opt.step_count))
    return res


@pytest.mark.parametrize("shape", [(2, 300, 520, 100, 1000), (3, 257, 1000, 64, 700), (2, 1100, 2600, 100, 4000)])
def test_row_indexed_step_is_bit_identical_to_gather_then_step(shape):
    A, B, D, H, n_rows = shape
    a, b = _rows_case(A, B, D, H, n_rows, 31)
    assert a[4] == b[4] == 2
    for u, v in zip(a[:4], b[:4]):
        assert torch.equal(u.view(torch.int32), v.view(torch.int32))
    assert bool(torch.isfinite(a[0]).all())


def test_row_indexed_step_on_the_bf16_engine():
    a, b = _rows_case(2, 300, 520, 100, 1000, 33, dtype="bf16")
    for u, v in zip(a[:4], b[:4]):
        assert torch.equal(u.view(torch.int32), v.view(torch.int32))


def test_row_indexed_step_is_refused_where_it_is_not_built():
    from oracle import restatement as R
    from tests import gpu_util as U
    h = R.Hyper(input_dim=256, fc_dim=32, n_categories=12, state_dim=2, lowD_dim=6, n_arm=2)
    m = U.build_model(h, R.init_state_dict(h, 1))
    m.train()
    m.gemm_dtype = "fp32_mfma"
    data = R.synthetic_batch(500, 256, seed=2).to(U.DEV)
    rows = torch.arange(96, device=U.DEV)
    off0 = m._noise_offset
    with pytest.raises(NotImplementedError):
        m.fused_train_step_rows(data, rows, 1.0, None, do_adam=False)
    assert m._noise_offset == off0                                  # the refused call consumed no noise
    m.fused_train_step(data[rows].expand(2, -1, -1), 1.0, None, do_adam=False)   # the gathered batch runs


def test_shuffled_epoch_through_row_indices_equals_gathered_batches(monkeypatch):
    """The trainer's epoch on a DeviceLoader: row-indexed steps (default) against gathered batches (MMVAE_ROWS=0) -- the same
    permutation, the same Philox noise offsets, bit-identical parameters and epoch history."""
    from distributed_vae_amd.cpl_mixvae import cpl_mixVAE
    from distributed_vae_amd.utils.dataloader import DeviceLoader
    from oracle import restatement as R
    from tests import gpu_util as U
    data = R.synthetic_batch(1300, 520, seed=9).to(U.DEV)
    out = []
    # gathered batches through loader rings of 4 (default), 2 and 1 slots (the last cannot be pipelined: the trainer must
    # notice) and fresh tensors per batch (ring 0): all the same numbers
    for mode, ring in (("1", 4), ("0", 4), ("0", 2), ("0", 1), ("0", 0)):
        monkeypatch.setenv("MMVAE_ROWS", mode)
        torch.manual_seed(77)
        t = cpl_mixVAE(saving_folder="", device=U.DEV, save_flag=False)
        t.init_model(n_categories=12, state_dim=2, input_dim=520, fc_dim=100, lowD_dim=6, x_drop=0.5, s_drop=0.0, n_arm=2)
        ld = DeviceLoader(data, torch.arange(1300), 256, True, True, seed=5, ring=ring)
        hist = t.train(ld, None, n_epoch=2, good_enuf_consensus=2.0)
        torch.cuda.synchronize()
        out.append((t.model.flat_parameters().detach().cpu().clone(), hist["losses"], getattr(t, "_rows_ok", True)))
    assert out[0][2] is True                                         # the row-indexed path was taken, not refused
    for o in out[1:]:
        assert torch.equal(out[0][0], o[0]) and out[0][1] == o[1]


def test_bf16_trainer_epoch_on_bf16_storage_equals_fp32_storage_of_the_rounded_matrix(monkeypatch):
    """``cpl_mixVAE.train`` in the bf16 configuration: the loader's bf16 copy feeds the row-indexed steps (default) -- same
    parameters and history, bit for bit, as with MMVAE_BF16_STORAGE=0 when the matrix holds bf16-representable values."""
    from distributed_vae_amd.cpl_mixvae import cpl_mixVAE
    from distributed_vae_amd.utils.dataloader import DeviceLoader
    from oracle import restatement as R
    from tests import gpu_util as U
    data = R.synthetic_batch(1300, 520, seed=9).to(torch.bfloat16).float().to(U.DEV)
    out = []
    for mode in ("1", "0"):
        monkeypatch.setenv("MMVAE_BF16_STORAGE", mode)
        torch.manual_seed(77)
        t = cpl_mixVAE(saving_folder="", device=U.DEV, save_flag=False)
        t.init_model(n_categories=12, state_dim=2, input_dim=520, fc_dim=100, lowD_dim=6, x_drop=0.5, s_drop=0.0, n_arm=2,
                     gemm_dtype="bf16")
        ld = DeviceLoader(data, torch.arange(1300), 256, True, True, seed=5)
        hist = t.train(ld, None, n_epoch=2, good_enuf_consensus=2.0)
        torch.cuda.synchronize()
        out.append((t.model.flat_parameters().detach().cpu().clone(), hist["losses"], ld._data16 is not None))
    assert out[0][2] is True and out[1][2] is False
    assert torch.equal(out[0][0], out[1][0]) and out[0][1] == out[1][1]
