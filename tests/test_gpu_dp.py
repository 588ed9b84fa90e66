"""-m gpu: the trainer's data-parallel path with TWO rank processes on one GPU (scope row (e); both ranks use cuda:0 and
the collectives go through gloo, as ``bench.py --share-gpu`` does -- RCCL refuses two ranks on one device, and a one-GPU
box is all these tests get; this is NOT a multi-GPU measurement).

Each child runs ``cpl_mixVAE.train`` for one epoch on ``DeviceLoader`` shards with DistributedSampler semantics
(``world_size = 2``), explicit noise per rank and step.  Checked against the virtual-rank oracle (SURVEY.md section 8e:
same weights, the ranks' batches through the oracle separately with rank-local BatchNorm / inv_var statistics, gradients
averaged, one Adam step):
  * the parameters are bit-equal across the ranks after the epoch;
  * they equal the oracle's averaged-gradient Adam trajectory to the trajectory tolerance of tests/test_gpu_trainer.py;
  * the folded epoch scalars (cpl_mixvae.py:480-492) equal the two-rank sums.
Also: ``tools/train_dp.py --gpus 2 --share-gpu`` (the entry that does what the reference's dead ``mp.spawn`` branch would,
train.py:269-288) runs end to end with Philox noise and leaves bit-equal replicas.
"""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import restatement as R

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WS = 2
CFG = dict(A=2, B=64, D=256, H=32, L=6, C=12, S=2, steps=2)


def _hyper():
    c = CFG
    return R.Hyper(input_dim=c["D"], fc_dim=c["H"], n_categories=c["C"], state_dim=c["S"], lowD_dim=c["L"], n_arm=c["A"])


def _data():
    c = CFG
    return R.synthetic_batch(WS * c["steps"] * c["B"], c["D"], seed=991)


def _noise(rank, step):
    return R.draw_noise(_hyper(), CFG["B"], seed=5000 + 100 * rank + step)


def _rank(rank, port, out_dir, gemm="fp32", storage="1"):
    """Child process: one rank of the two (fresh interpreter, started before it touches the GPU)."""
    sys.path.insert(0, ROOT)
    os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    os.environ["MMVAE_BF16_STORAGE"] = storage
    import distributed_vae_amd  # noqa: F401
    from distributed_vae_amd import dist as D
    from distributed_vae_amd.cpl_mixvae import cpl_mixVAE
    from distributed_vae_amd.utils.dataloader import DeviceLoader
    from tests import gpu_util as U

    c = CFG
    h = _hyper()
    torch.cuda.set_device(0)
    D.init_dist_env(rank, WS, "127.0.0.1", port, backend="gloo")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    data = _data().to(dev)
    if gemm == "bf16":
        data = data.to(torch.bfloat16).float()        # bf16-representable values: storage on / off must agree bit for bit
    loader = DeviceLoader(data, torch.arange(data.shape[0]), c["B"], True, True, seed=546, world_size=WS, rank=rank)
    order = loader.index[loader.epoch_order_device()].cpu().numpy()     # the rows this rank visits in epoch 0
    t = cpl_mixVAE(saving_folder="", device=dev, save_flag=False)
    torch.manual_seed(7 + rank)                  # replicas start different: train() broadcasts rank 0's parameters
    t.init_model(n_categories=c["C"], state_dim=c["S"], input_dim=c["D"], fc_dim=c["H"], lowD_dim=c["L"], x_drop=0.5,
                 s_drop=0.0, n_arm=c["A"], gemm_dtype=gemm)
    if rank == 0:
        t.model.load_state_dict(R.init_state_dict(h, 546))
    t.model.set_explicit_noise([U.noise_to_device(_noise(rank, s), dev) for s in range(c["steps"])])
    loader.set_epoch(0)
    hist = t.train(loader, None, n_epoch=1, rank=rank, ws=WS, good_enuf_consensus=2.0)
    torch.cuda.synchronize()
    assert not t.model._explicit_noise           # every scheduled draw was consumed by a train step
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), used16=np.array(loader._data16 is not None), params=t.model.flat_parameters().detach().cpu().numpy(), order=order,
             losses=np.array(hist["losses"]), c_dists=np.array(hist["c_dists"]), loss_joints=np.array(hist["loss_joints"]),
             loss_recs=np.array(hist["loss_recs"]), bn=t.model._bn_flat.detach().cpu().numpy(),
             **{"p/" + k: v.detach().cpu().numpy() for k, v in t.model.state_dict().items()})
    D.dist.barrier()
    D.dist.destroy_process_group()


def test_two_ranks_on_one_gpu_against_the_virtual_rank_oracle(tmp_path):
    import multiprocessing as mp
    from distributed_vae_amd import dist as D
    c, h = CFG, _hyper()
    port = D.find_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_rank, args=(r, port, str(tmp_path))) for r in range(WS)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0, p.exitcode
    out = [np.load(os.path.join(str(tmp_path), f"rank{r}.npz")) for r in range(WS)]
    # 1. replicas bit-equal (same averaged gradients, same Adam arithmetic on every rank)
    assert np.array_equal(out[0]["params"], out[1]["params"])
    # 2. disjoint shards of one permutation (DistributedSampler semantics)
    o0, o1 = out[0]["order"], out[1]["order"]
    assert len(o0) == len(o1) == c["steps"] * c["B"] and not set(o0.tolist()) & set(o1.tolist())
    # 3. the virtual-rank oracle: rank-local statistics, averaged gradients, one Adam step per step
    X = _data()
    sd = R.init_state_dict(h, 546)
    keys = R.param_keys(h)
    st = {"t": 0, "m": {k: torch.zeros_like(sd[k]) for k in keys}, "v": {k: torch.zeros_like(sd[k]) for k in keys}}
    bn_local = [{k: v.clone() for k, v in sd.items() if k not in keys} for _ in range(WS)]
    sums = np.zeros(5 + 3 * c["A"])
    for s in range(c["steps"]):
        grads = []
        for r in range(WS):
            rows = torch.from_numpy(out[r]["order"][s * c["B"]:(s + 1) * c["B"]])
            work = {**{k: sd[k].clone() for k in keys}, **bn_local[r]}
            _, lt, g = R.grads_autograd(work, [X[rows]] * c["A"], h, _noise(r, s))
            bn_local[r] = {k: v for k, v in work.items() if k not in keys}
            grads.append(g)
            sums[0] += float(lt[0]); sums[1] += float(lt[2]); sums[3] += float(lt[4])
            sums[5:5 + c["A"]] += np.array([float(v) for v in lt[1]])
        st["t"] += 1
        for k in keys:
            gavg = sum(g[k] for g in grads) / WS
            sd[k], st["m"][k], st["v"][k] = R.adam_step(sd[k], gavg, st["m"][k], st["v"][k], st["t"], 1e-3)
    for k in keys:
        d = np.abs(out[0]["p/" + k] - sd[k].numpy())
        assert float(np.median(d)) < 2e-5 and float(d.max()) < 2.1e-3, (k, float(np.median(d)), float(d.max()))
    # rank-local BatchNorm running statistics (rank 0's are the ones a checkpoint would hold)
    for k, v in bn_local[0].items():
        if v.dtype.is_floating_point:
            assert np.abs(out[0]["p/" + k] - v.numpy()).max() <= 1e-4 * (np.abs(v.numpy()).max() + 1e-6), k
    # 4. folded epoch scalars: all-reduced sums over both ranks' steps / the all-reduced step count (cpl_mixvae.py:480-492)
    nsteps = WS * c["steps"]
    for r in range(WS):
        assert abs(out[r]["losses"][0] - sums[0] / nsteps) <= 2e-4 * abs(sums[0] / nsteps)
        assert abs(out[r]["c_dists"][0] - sums[3] / nsteps) <= 2e-4 * abs(sums[3] / nsteps)
        for a in range(c["A"]):
            want = sums[5 + a] / c["D"] / nsteps
            assert abs(out[r]["loss_recs"][a][0] - want) <= 2e-4 * abs(want)
    assert out[0]["losses"][0] == out[1]["losses"][0]


def test_two_ranks_in_the_bf16_configuration_on_bf16_storage(tmp_path):
    """The data-parallel trainer in the bf16 configuration: the row-indexed steps read each rank's loader through its bf16 copy
    (``dp_train_step(rows=(data, rows, data16))``); replicas bit-equal, and the same parameters as with the copy switched off."""
    import multiprocessing as mp
    from distributed_vae_amd import dist as D
    ctx = mp.get_context("spawn")
    res = {}
    for storage in ("1", "0"):
        d = tmp_path / storage
        d.mkdir()
        port = D.find_port()
        procs = [ctx.Process(target=_rank, args=(r, port, str(d), "bf16", storage)) for r in range(WS)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(300)
            assert p.exitcode == 0, p.exitcode
        res[storage] = [np.load(os.path.join(str(d), f"rank{r}.npz")) for r in range(WS)]
    assert bool(res["1"][0]["used16"]) and not bool(res["0"][0]["used16"])
    assert np.array_equal(res["1"][0]["params"], res["1"][1]["params"])
    assert np.array_equal(res["1"][0]["params"], res["0"][0]["params"])
    assert np.isfinite(res["1"][0]["params"]).all() and res["1"][0]["losses"][0] == res["0"][0]["losses"][0]


@pytest.mark.parametrize("extra", [[], ["--no-augmentation", "--hard", "False", "--loss_mode", "MSE", "--n_epoch_p", "0",
                                        "--min_con", "0.99", "--max_prun_it", "0"]],
                         ids=["augmenter_on_as_the_reference_default", "no_augmenter_hard"])
def test_train_dp_entry_runs_two_ranks_on_one_gpu(tmp_path, extra):
    """tools/train_dp.py spawns its ranks before touching a GPU and never re-execs; with Philox noise (different streams per
    rank would be wrong here: the replicas must stay identical, which they do because only gradients are averaged).  The
    flag set is the reference's (train.py:172-267): ``--augmentation`` defaults to True (random-init Augmenter_smartseq in
    front of every step: the pretrained file is not shipped), ``--hard False`` is True (``type=bool``)."""
    out = str(tmp_path / "o")
    cmd = [sys.executable, os.path.join(ROOT, "tools", "train_dp.py"), "--gpus", "2", "--share-gpu", "--cells", "2048",
           "--genes", "256", "--fc_dim", "32", "--latent_dim", "6", "--n_categories", "12", "--batch_size", "128",
           "--n_epoch", "2", "--good-enuf-consensus", "2.0", "--out", out] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    a, b = torch.load(os.path.join(out, "rank0.pt"), weights_only=False), torch.load(os.path.join(out, "rank1.pt"), weights_only=False)
    assert a["backend"] == "gloo" and a["steps_per_epoch"] == (2048 * 9 // 10 + 1) // 2 // 128
    assert torch.equal(a["params"], b["params"]) and bool(torch.isfinite(a["params"]).all())
    assert len(a["hist"]["losses"]) == 2 and np.isfinite(a["hist"]["losses"]).all()
    assert a["hist"]["losses"] == b["hist"]["losses"]        # all-reduced epoch means: the same number on every rank
    assert not torch.equal(a["bn"], b["bn"])                  # BatchNorm running statistics stay rank-local


def test_library_allreduce_entry_on_one_rank():
    """mmvae_dp_unique_id / mmvae_dp_init / mmvae_allreduce_grads / mmvae_dp_destroy (include/mmvae.h) on a one-rank
    communicator -- all a one-GPU box can run (RCCL refuses two ranks on one device): the average over one rank is the
    identity, the call is ordered on the stream it is given, and a destroyed communicator is refused."""
    import distributed_vae_amd  # noqa: F401
    from distributed_vae_amd import dist as D
    dev = torch.device("cuda", 0)
    comm = D.DirectComm(0, 1, dev)
    g = torch.randn(1 << 20, device=dev)
    want = g.clone() * 3.0
    s = torch.cuda.Stream(device=dev)
    s.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(s):
        g.mul_(3.0)                       # queued in front of the collective on the same stream
        comm.allreduce_mean_(g)
        out = g + 0.0
    s.synchronize()
    assert torch.equal(out, want)
    comm.close()
    comm.close()
