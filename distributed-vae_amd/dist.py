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


class DirectComm:
    """An RCCL communicator of this process for the library's own gradient all-reduce (``mmvae_allreduce_grads``,
    include/mmvae.h): ONE stream-ordered ``ncclAllReduce(avg)`` on the stream the step runs on -- no torch NCCL stream, no
    event round trip.  Rank 0 draws the RCCL unique id and the initialised ``torch.distributed`` group (any backend) carries
    its 128 bytes to the other ranks; every rank must construct it (``ncclCommInitRank`` is collective)."""

    def __init__(self, rank: int, world_size: int, device):
        import ctypes as C
        from . import _native as N
        self.device = torch.device(device)
        ids = [None]
        if rank == 0:
            buf = (C.c_uint8 * 128)()
            N.check(N.lib().mmvae_dp_unique_id(buf), "mmvae_dp_unique_id")
            ids = [bytes(buf)]
        if world_size > 1:
            dist.broadcast_object_list(ids, src=0)
        self._comm = C.c_void_p()
        raw = (C.c_uint8 * 128).from_buffer_copy(ids[0])
        with torch.cuda.device(self.device):
            N.check(N.lib().mmvae_dp_init(raw, int(rank), int(world_size), C.byref(self._comm)), "mmvae_dp_init")

    def allreduce_mean_(self, buf: torch.Tensor):
        from . import _native as N
        assert buf.is_cuda and buf.dtype == torch.float32 and buf.is_contiguous()
        if buf.device.index != (self.device.index or 0):
            raise N.NativeError(f"DirectComm of {self.device} asked to reduce a buffer on {buf.device}")
        N.check(N.lib().mmvae_allreduce_grads(self._comm, buf.data_ptr(), buf.numel(),
                                              torch.cuda.current_stream(buf.device).cuda_stream), "mmvae_allreduce_grads")
        return buf

    def close(self):
        from . import _native as N
        if self._comm:
            N.lib().mmvae_dp_destroy(self._comm)
            self._comm = None


_DIRECT = {}


def direct_comm(device) -> "DirectComm":
    """The process's DirectComm for ``device`` (created on first use: collective, every rank reaches it in its first
    data-parallel step)."""
    key = torch.device(device).index or 0
    if key not in _DIRECT:
        ws = dist.get_world_size() if dist.is_initialized() else 1
        rk = dist.get_rank() if dist.is_initialized() else 0
        _DIRECT[key] = DirectComm(rk, ws, device)
        if len(_DIRECT) == 1:
            import atexit
            atexit.register(close_direct_comms)       # communicators are destroyed before the interpreter tears torch down
    return _DIRECT[key]


def close_direct_comms():
    """Destroy the process's library-owned RCCL communicators (also registered with ``atexit``); call it ahead of
    ``dist.destroy_process_group()``."""
    for comm in list(_DIRECT.values()):
        try:
            comm.close()
        except Exception:   # noqa: BLE001
            pass
    _DIRECT.clear()


def dp_train_step(model, xs, temp, optimizer, rehearse: bool = False, rows=None):
    """One data-parallel step: fused forward + loss + backward, gradient all-reduce (average), Adam.

    Default: ONE all-reduce of the whole flat buffer when backward is done.  MMVAE_DP_OVERLAP=1 (RCCL only) splits it in
    TWO: the fc11.weight / fc11.bias ranges of every arm (47 % of the buffer) are final as soon as their GEMM has finished
    on the side stream, so they are gathered into one staging buffer and all-reduced there, on a communication stream,
    beside the rest of backward; the remaining ranges follow as the second collective.  Off by default: no multi-GPU node
    has been available to tune it on (round 2's form, 2 A collectives, cost 65 us per step on one rank).
    MMVAE_DP_DIRECT=1 (RCCL only): the single all-reduce is issued by the library itself (``mmvae_allreduce_grads``), on the
    step's own stream."""
    import os
    from . import _native as N
    active = is_dist() or (rehearse and dist.is_available() and dist.is_initialized())   # rehearse: world size 1
    overlap = (active and dist.get_backend() == "nccl" and os.environ.get("MMVAE_DP_OVERLAP", "0") == "1")
    B = xs.shape[-2] if rows is None else int(rows[1].numel())
    eng = model._ensure(B)
    eng.enable_early_grad_event(overlap)
    if rows is not None:      # (data, row indices[, bf16 copy of data]): the batch is read through a row map, never materialised
        buf = model.fused_train_step_rows(rows[0], rows[1], temp, optimizer, do_adam=False,
                                          data16=rows[2] if len(rows) > 2 else None)
    else:
        buf = model.fused_train_step(xs, temp, optimizer, do_adam=False)
    flat = model.flat_grad()
    if overlap and eng.early_event is not None and eng.early_recorded():
        # TWO collectives (round 2 issued 2 A): the fc11 ranges of all arms are gathered into one contiguous staging buffer
        # (one strided copy kernel) and all-reduced on the communication stream as soon as their GEMM has finished; the
        # remaining ranges follow the same way when backward is done
        lay = model._layout
        per_arm, o26 = int(lay.per_arm), int(lay.offset[26])
        A = flat.numel() // per_arm
        v = flat.view(A, per_arm)
        st = getattr(model, "_dp_stage", None)
        if st is None or st[0].numel() != A * (per_arm - o26) or st[0].device != flat.device:
            st = (torch.empty(A, per_arm - o26, device=flat.device), torch.empty(A, o26, device=flat.device))
            model._dp_stage = st
        comm = N.shared_stream(flat.device, "comm")
        comm.wait_event(eng.early_event)
        with torch.cuda.stream(comm):
            st[0].copy_(v[:, o26:])
            work = dist.all_reduce(st[0], op=dist.ReduceOp.AVG, async_op=True)
        st[1].copy_(v[:, :o26])
        dist.all_reduce(st[1], op=dist.ReduceOp.AVG)
        v[:, :o26].copy_(st[1])
        with torch.cuda.stream(comm):
            work.wait()                      # the communication stream waits for its collective
            v[:, o26:].copy_(st[0])
        torch.cuda.current_stream(flat.device).wait_stream(comm)
    elif active and os.environ.get("MMVAE_DP_DIRECT", "0") == "1" and dist.get_backend() == "nccl":
        # the library's own RCCL all-reduce on the step's stream (opt-in: no multi-GPU node has run it yet)
        direct_comm(flat.device).allreduce_mean_(flat)
    elif active and not is_dist():
        dist.all_reduce(flat, op=dist.ReduceOp.AVG if dist.get_backend() == "nccl" else dist.ReduceOp.SUM)
    else:
        allreduce_mean_(flat)
    if not hasattr(optimizer, "_bind"):      # a stock torch.optim optimizer: it reads the gradients through p.grad
        model.bind_grads()
    optimizer.step()
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
