#!/usr/bin/env python3
"""Data-parallel training entry for the cpl-mixVAE trainer: what the reference's ``train.py`` would do with ``--gpus N``
if its distributed branch were live (``/root/reference/train.py:269-288`` raises NotImplementedError at :274-275; the
branch below it spawns one process per GPU with ``mp.spawn(main, args=(ws, args), nprocs=ws)`` and ``main`` brings the
process group up through ``init_dist_env``, :81-83).

  python tools/train_dp.py --gpus 8 --n_arm 2 --n_epoch 10                # one process per GPU, RCCL over xGMI
  python tools/train_dp.py --gpus 2 --share-gpu --cells 4096 --genes 256  # rehearsal: both ranks on cuda:0, gloo

* The parent starts the ranks BEFORE it touches any GPU (spawn start method: fresh interpreters, never a re-exec of a
  process that has initialised HIP) and exits with their status.
* Every rank: process group (RCCL = torch.distributed "nccl"; gloo for the one-GPU rehearsal, where RCCL refuses two ranks
  on one device), the cells x genes matrix resident on its GPU, ``get_loaders(use_dist_sampler=True, world_size, rank)``
  (DistributedSampler semantics: disjoint shards of every epoch's permutation), ``cpl_mixVAE.init_model``, parameter
  broadcast from rank 0, ``cpl_mixVAE.train``: per step one all-reduce (mean) of the flat gradient buffer, per epoch one
  folded scalar all-reduce (cpl_mixvae.py:480-483).  BatchNorm / inv_var statistics stay rank-local (no SyncBatchNorm in
  the reference).
* Data: synthetic-10x-v1 (the h5ad loader needs ``anndata``, which this image lacks) or ``--npy FILE``
  with a dense [cells, genes] float32 matrix.

Flags keep the reference's names and defaults where it has them (train.py:174-266).
"""
from __future__ import annotations

import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEED = 546          # train.py:27


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--n_categories", type=int, default=92)
    p.add_argument("--state_dim", type=int, default=2)
    p.add_argument("--n_arm", type=int, default=2)
    p.add_argument("--temp", type=float, default=1.0)
    p.add_argument("--tau", type=float, default=0.005)
    p.add_argument("--beta", type=float, default=1.0)
    p.add_argument("--lam", type=float, default=1.0)
    p.add_argument("--lam_pc", type=float, default=1.0)
    p.add_argument("--latent_dim", type=int, default=10)
    p.add_argument("--n_epoch", type=int, default=10)
    p.add_argument("--fc_dim", type=int, default=100)
    p.add_argument("--batch_size", type=int, default=5000)
    p.add_argument("--lr", type=float, default=1e-3)
    p.add_argument("--p_drop", type=float, default=0.5)
    p.add_argument("--s_drop", type=float, default=0.0)
    p.add_argument("--optimizer", type=str, default="adam", choices=["adam", "adamw"])
    p.add_argument("--gpus", type=int, default=1)
    # the rest of the reference's flag set (train.py:172-267), with its defaults and its type quirks: ``type=bool`` flags are
    # True for ANY non-empty value ("--hard False" is True), exactly as argparse gives them to the reference
    p.add_argument("--n_epoch_p", default=0, type=int, help="epochs of the pruning phase (train.py:196-200; disabled upstream)")
    p.add_argument("--min_con", default=0.99, type=float, help="minimum consensus (train.py:202)")
    p.add_argument("--max_prun_it", default=0, type=int, help="train.py:203-208")
    p.add_argument("--variational", default=True, type=bool, help="train.py:213-215")
    p.add_argument("--augmentation", default=True, type=bool,
                   help="VAE-GAN augmenter in front of every step (train.py:216-217; the production default).  The pretrained "
                        "file is not shipped: --aug_file names a checkpoint in the reference's layout, else the architecture "
                        "runs on random-init weights (results unpinned, the arithmetic and the cost are the production path's)")
    p.add_argument("--aug_file", default="", type=str, help="augmenter checkpoint (mmidas.toml 'aug'; cpl_mixvae.py:128-149)")
    p.add_argument("--no-augmentation", dest="augmentation", action="store_const", const=False,
                   help="not in the reference: the only way to switch a type=bool flag off from the command line")
    p.add_argument("--ref_pc", default=False, type=bool, help="train.py:225-227 (rejected by loss(), nn_model.py:578)")
    p.add_argument("--pretrained_model", default=False, type=bool, help="train.py:229-231: start from --trained_model")
    p.add_argument("--trained_model", default="", type=str, help="checkpoint for --pretrained_model (mmidas.toml 'trained')")
    p.add_argument("--n_pr", default=0, type=int, help="train.py:232-237")
    p.add_argument("--loss_mode", default="MSE", type=str, help="train.py:241-243 (ZINB is rejected by forward, nn_model.py:315)")
    p.add_argument("--hard", default=False, type=bool, help="train.py:245: straight-through one-hot samples")
    p.add_argument("--log-dir", dest="log_dir", default="", help="directory for the ranks' rank<r>.err files")
    p.add_argument("--gemm-dtype", default="fp32", choices=["fp32", "bf16", "fp32_mfma"])
    p.add_argument("--good-enuf-consensus", type=float, default=0.75)
    # data
    p.add_argument("--cells", type=int, default=50000, help="synthetic cells in the whole data set (split 90/10, then sharded)")
    p.add_argument("--genes", type=int, default=5000)
    p.add_argument("--npy", type=str, default="", help="dense float32 [cells, genes] matrix instead of the synthetic one")
    p.add_argument("--saving-folder", type=str, default="")
    p.add_argument("--out", type=str, default="", help="directory for rank<r>.pt result files (parameters, history)")
    p.add_argument("--share-gpu", action="store_true",
                   help="rehearsal on a box with fewer GPUs than ranks: every rank uses cuda:0, collectives through gloo")
    return p.parse_args(argv)


def main(rank: int, ws: int, args, port: int) -> None:
    """One rank (reference: train.py::main(rank, ws, args), :81-166)."""
    sys.path.insert(0, ROOT)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import numpy as np
    import torch
    import torch.distributed as dist

    import distributed_vae_amd  # noqa: F401
    from distributed_vae_amd import dist as D
    from distributed_vae_amd.cpl_mixvae import cpl_mixVAE
    from distributed_vae_amd.utils.dataloader import get_loaders

    local = 0 if args.share_gpu else rank
    if not torch.cuda.is_available() or local >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: no GPU {local} (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if ws > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        D.init_dist_env(rank, ws, "127.0.0.1", port, backend="gloo" if args.share_gpu else "nccl")
        torch.cuda.set_device(local)
    # the data set: identical on every rank (the shards are cut from it by the sampler logic)
    if args.npy:
        data = torch.from_numpy(np.load(args.npy).astype(np.float32)).to(dev)
    else:
        g = torch.Generator(device=dev).manual_seed(SEED)
        data = (torch.rand(args.cells, args.genes, generator=g, device=dev) < 0.2).float()
        data *= torch.randn(args.cells, args.genes, generator=g, device=dev).abs() * 3.0
    n, d = data.shape
    if rank == 0:
        print(f"# cells: {n}, # genes: {d}, world size: {ws}", flush=True)
    trainer = cpl_mixVAE(args.saving_folder, args.aug_file if args.augmentation else "", dev,
                         save_flag=bool(args.saving_folder) and rank == 0)
    if args.augmentation and not args.aug_file:
        # the authors' pretrained augmenter is not shipped (mmidas.toml:27): the Augmenter_smartseq architecture on
        # random-init weights, identical on every rank (it is frozen: no gradient ever reaches it)
        from distributed_vae_amd.augmentation import Augmenter_smartseq
        torch.manual_seed(SEED)
        trainer.set_augmenter(Augmenter_smartseq(50, 10, d, 500))
    train_loader, test_loader, _ = get_loaders(dataset=data, seed=SEED, batch_size=args.batch_size, world_size=ws,
                                               rank=rank, use_dist_sampler=True, device=dev)
    torch.manual_seed(SEED + rank)     # replicas start different on purpose: rank 0's parameters win the broadcast in train()
    trainer.init_model(n_categories=args.n_categories, state_dim=args.state_dim, input_dim=d, fc_dim=args.fc_dim,
                       lowD_dim=args.latent_dim, x_drop=args.p_drop, s_drop=args.s_drop, lr=args.lr, n_arm=args.n_arm,
                       temp=args.temp, hard=args.hard, tau=args.tau, lam=args.lam, lam_pc=args.lam_pc, beta=args.beta,
                       ref_prior=args.ref_pc, variational=args.variational,
                       trained_model=args.trained_model if args.pretrained_model else "", n_pr=args.n_pr,
                       mode=args.loss_mode, gemm_dtype=args.gemm_dtype)
    if args.optimizer == "adamw":      # train.py:146-147
        trainer.optimizer = torch.optim.AdamW(trainer.model.parameters(), lr=args.lr)
    hist = trainer.train(train_loader=train_loader, test_loader=None, n_epoch=args.n_epoch, n_epoch_p=args.n_epoch_p,
                         min_con=args.min_con, max_prun_it=args.max_prun_it, rank=rank, ws=ws,
                         good_enuf_consensus=args.good_enuf_consensus)
    torch.cuda.synchronize()
    if args.out:
        os.makedirs(args.out, exist_ok=True)
        torch.save({"params": trainer.model.flat_parameters().detach().cpu(), "bn": trainer.model._bn_flat.detach().cpu(),
                    "hist": {k: v for k, v in hist.items() if k != "loss_recs"}, "loss_recs": hist.get("loss_recs"),
                    "steps_per_epoch": len(train_loader), "backend": dist.get_backend() if ws > 1 else None},
                   os.path.join(args.out, f"rank{rank}.pt"))
    if ws > 1:
        dist.barrier()
        dist.destroy_process_group()


def launch(args) -> int:
    """Start the ranks (no GPU call has been made in this process) and return the exit status to leave with."""
    import multiprocessing as mp
    import socket
    import tempfile
    import time

    ws = args.gpus
    if ws <= 1:
        main(0, 1, args, 0)
        return 0
    import torch    # counting devices does not initialise HIP
    have = torch.cuda.device_count()
    if have < ws and not (args.share_gpu and have >= 1):
        print(f"train_dp.py: --gpus {ws} but only {have} GPU(s) visible", file=sys.stderr, flush=True)
        return 2
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    log_dir = args.log_dir or tempfile.mkdtemp(prefix="mmvae_train_dp_")
    os.makedirs(log_dir, exist_ok=True)
    procs = [ctx.Process(target=_rank_entry, args=(r, ws, args, port, log_dir)) for r in range(ws)]
    for p in procs:
        p.start()
    # supervise: the first rank that exits non-zero stops the others (they would otherwise wait in their next collective
    # until the process-group timeout); only the children started here are signalled, by handle
    rc = 0
    while True:
        codes = [p.exitcode for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            r, c = bad[0]
            rc = abs(c) or 1
            tail = ""
            try:
                tail = "\n".join(open(os.path.join(log_dir, f"rank{r}.err"), errors="replace").read().splitlines()[-30:])
            except OSError:
                pass
            print(f"rank {r} exited with status {c}; stopping the other ranks.  Its stderr tail:\n{tail}", file=sys.stderr, flush=True)
            break
        if all(c == 0 for c in codes):
            break
        time.sleep(0.2)
    for p in procs:
        if p.exitcode is None:
            p.terminate()
    for p in procs:
        p.join(5)
        if p.exitcode is None:
            p.kill()
            p.join()
    return rc


def _rank_entry(rank: int, ws: int, args, port: int, log_dir: str) -> None:
    """A rank's stderr goes to <log_dir>/rank<r>.err (the parent prints its tail when the rank fails)."""
    fd = os.open(os.path.join(log_dir, f"rank{rank}.err"), os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
    os.dup2(fd, 2)
    os.close(fd)
    if os.environ.get("MMVAE_TRAIN_FAIL_RANK") == str(rank):   # test hook
        print(f"rank {rank}: MMVAE_TRAIN_FAIL_RANK set, exiting 1", file=sys.stderr, flush=True)
        sys.exit(1)
    main(rank, ws, args, port)


if __name__ == "__main__":
    a = parse_args()
    print(json.dumps({"world_size": a.gpus, "share_gpu": a.share_gpu}), flush=True)
    sys.exit(launch(a))
