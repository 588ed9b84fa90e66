"""CPU, world_size = 2 over gloo: the data-parallel host logic of distributed-vae_amd/dist.py.

The HIP engine cannot run here, so each rank computes its rank-local gradients with the oracle
(rank-local BatchNorm / inv_var statistics), packs them into the flat
gradient buffer in the engine's layout, and the package's collectives are exercised for real:
broadcast of the flat parameter buffer, ONE all-reduce(mean) of the flat gradients, the folded
per-epoch scalar reduction.  Expected values: the reference-generated "virtual rank" fixture
(tests/golden/tiny_a2.npz, dp2/*)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _worker(rank, ws, port, out_dir):
    sys.path.insert(0, ROOT)
    import distributed_vae_amd  # noqa: F401
    from distributed_vae_amd import dist as D
    from distributed_vae_amd.cpl_mixvae import FusedAdam
    from distributed_vae_amd.nn_model import mixVAE_model
    from oracle import restatement as R
    from tests import golden_util as G

    torch.set_num_threads(2)
    D.init_dist_env(rank, ws, "127.0.0.1", port, backend="gloo")
    g = G.load("tiny_a2")
    h = G.hyper_of(g)
    B = G.batch_of(g)
    # replicas start different on purpose; rank 0's parameters must win the broadcast
    torch.manual_seed(1000 + rank)
    m = mixVAE_model(input_dim=h.input_dim, fc_dim=h.fc_dim, n_categories=h.n_categories, state_dim=h.state_dim,
                     lowD_dim=h.lowD_dim, x_drop=0.5, s_drop=0.0, n_arm=h.n_arm, lam=1, lam_pc=1, tau=0.005, beta=1.0,
                     hard=False, variational=True, device="cpu", eps=1e-8, momentum=0.01, ref_prior=False,
                     loss_mode="MSE")
    if rank == 0:
        m.load_state_dict(G.state_dict_of(g))
    flat = m.flat_parameters()
    D.broadcast_flat(flat)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    ref0 = G.state_dict_of(g)
    ok_bcast = all(torch.equal(sd[k], ref0[k]) for k in R.param_keys(h))
    # rank-local step through the oracle on this rank's shard
    x = R.synthetic_batch(B, h.input_dim, seed=546 + 200 + rank)
    _, lt, grads = R.grads_autograd(sd, [x] * h.n_arm, h, R.draw_noise(h, B, seed=2000 + rank))
    with torch.no_grad():
        for k, p in m.named_parameters():
            gv = m._grad_views[list(dict(m.named_parameters())).index(k)]
            gv.copy_(grads[k])
    D.allreduce_mean_(m.flat_grad())          # the ONE collective of a DP step
    red = torch.tensor([float(lt[0]), 1.0])
    D.allreduce_sum_(red)                      # per-epoch scalars (cpl_mixvae.py:480-483)
    lo, hi = D.shard_rows(1000, rank, ws)
    if rank == 0:
        np.savez(os.path.join(out_dir, "dp.npz"), ok_bcast=ok_bcast, red=red.numpy(), lo=lo, hi=hi,
                 **{"g/" + k: p.grad.numpy() if p.grad is not None else
                    m._grad_views[i].numpy() for i, (k, p) in enumerate(m.named_parameters())})
    assert D.is_dist()
    D.dist.barrier()
    D.dist.destroy_process_group()


def test_dp_two_ranks_gloo(tmp_path):
    from distributed_vae_amd import dist as D
    from tests import golden_util as G

    port = D.find_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    out = np.load(os.path.join(str(tmp_path), "dp.npz"))
    g = G.load("tiny_a2")
    assert bool(out["ok_bcast"])
    for k in g.files:
        if k.startswith("dp2/grad/"):
            name = k[len("dp2/grad/"):]
            assert G.rel_err(out["g/" + name], g[k]) < 1e-4, name
    want = float(g["dp2/loss_rank0"]) + float(g["dp2/loss_rank1"])
    assert abs(float(out["red"][0]) - want) <= 1e-5 * abs(want) and float(out["red"][1]) == 2.0
    assert (int(out["lo"]), int(out["hi"])) == (0, 500)


def test_single_process_helpers_are_noops():
    import distributed_vae_amd  # noqa: F401
    from distributed_vae_amd import dist as D
    t = torch.arange(4.0)
    assert not D.is_dist()
    assert torch.equal(D.allreduce_mean_(t.clone()), t)
    assert torch.equal(D.allreduce_sum_(t.clone()), t)
    D.broadcast_flat(t)
    assert D.shard_rows(10, 1, 2) == (5, 10)


def _solo(rank, ws, port, out_dir):
    """BASELINE.json configs[0] as it is worded: a process group of world_size = 1 on the CPU over gloo."""
    sys.path.insert(0, ROOT)
    import distributed_vae_amd  # noqa: F401
    from distributed_vae_amd import dist as D
    D.init_dist_env(rank, ws, "127.0.0.1", port, backend="gloo")
    assert D.dist.is_initialized() and D.dist.get_world_size() == 1 and D.dist.get_backend() == "gloo"
    assert not D.is_dist()                       # one rank: every helper is the identity, no collective is issued
    t = torch.arange(6.0)
    assert torch.equal(D.allreduce_mean_(t.clone()), t) and torch.equal(D.allreduce_sum_(t.clone()), t)
    D.broadcast_flat(t)
    assert D.shard_rows(1000, 0, 1) == (0, 1000)
    # the real collectives of a one-rank group are identities too
    u = t.clone()
    D.dist.all_reduce(u)
    assert torch.equal(u, t)
    D.dist.barrier()
    D.dist.destroy_process_group()
    open(os.path.join(out_dir, "ok"), "w").write("1")


def test_world_size_one_gloo_group(tmp_path):
    from distributed_vae_amd import dist as D
    mp.spawn(_solo, args=(1, D.find_port(), str(tmp_path)), nprocs=1, join=True)
    assert os.path.exists(os.path.join(str(tmp_path), "ok"))
