"""Data parallelism for the cpl-mixVAE step: one process per GPU, parameters replicated, ONE
all-reduce (average) of the flat fp32 gradient buffer per step over RCCL/xGMI.

Replaces the reference's process-group bring-up (mmidas/_dist_utils.py:12-55) and its (dead) FSDP
wrap of the VAE (train.py:140-143): the model is 1.07 M parameters per arm, so sharding is
batch-only.  BatchNorm / inv_var statistics stay rank-local, exactly what
DDP/FSDP without SyncBatchNorm would do.
"""
from __future__ import annotations

import os
import socket
from typing import Optional

import torch
import torch.distributed as dist


def find_port(addr: str = "127.0.0.1") -> int:
    """mmidas/_dist_utils.py:58-67"""
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind((addr, 0))
        return s.getsockname()[1]


def init_dist_env(rank: int, world_size: int, addr: str = "127.0.0.1", port: Optional[int] = None,
                  backend: Optional[str] = None, timeout_s: int = 300):
    """mmidas/_dist_utils.py:12-55: env rendezvous, NCCL(=RCCL on ROCm) on GPUs, gloo on CPU."""
    import datetime

    os.environ.setdefault("MASTER_ADDR", addr)
    if port is not None:
        os.environ.setdefault("MASTER_PORT", str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(rank % max(torch.cuda.device_count(), 1))
    dist.init_process_group(backend, rank=rank, world_size=world_size,
                            timeout=datetime.timedelta(seconds=timeout_s))


def is_dist() -> bool:
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def broadcast_flat(buf: torch.Tensor, src: int = 0):
    """Parameter broadcast at init: the reference never sets sync_module_states (train.py:141-143)
    and relies on identical seeds; we make replicas identical explicitly."""
    if is_dist():
        dist.broadcast(buf, src=src)


def allreduce_mean_(buf: torch.Tensor):
    """One collective per step over the whole flat gradient buffer (8.6 MB for A=2, D=5000)."""
    if not is_dist():
        return buf
    ws = dist.get_world_size()
    if dist.get_backend() == "nccl":
        dist.all_reduce(buf, op=dist.ReduceOp.AVG)
    else:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
        buf.div_(ws)
    return buf


def allreduce_sum_(buf: torch.Tensor):
    """Per-epoch scalar reduction (cpl_mixvae.py:480-483), folded into one small tensor."""
    if is_dist():
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    return buf


def shard_rows(n: int, rank: int, world_size: int):
    """Contiguous shard of n cells for this rank."""
    per = n // world_size
    return rank * per, (rank + 1) * per
