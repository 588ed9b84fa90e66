"""Per-step time of the fused step on fixed batches against row-indexed batches (mmvae_train_step_rows): random rows of the
resident matrix, and the identity map on the same rows -- what the row map itself costs, without any loader work."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import distributed_vae_amd  # noqa
from distributed_vae_amd.cpl_mixvae import FusedAdam
from distributed_vae_amd.nn_model import mixVAE_model
dev = torch.device("cuda", 0)
A, B, D = 2, 5000, 5000
data = bench.synthetic_rows(50000, D, 546, dev)
torch.manual_seed(546)
m = mixVAE_model(input_dim=D, fc_dim=100, n_categories=92, state_dim=2, lowD_dim=10, x_drop=0.5, s_drop=0.0, n_arm=A, lam=1, lam_pc=1,
                 tau=0.005, beta=1.0, hard=False, variational=True, device=dev, eps=1e-8, momentum=0.01, ref_prior=False, loss_mode="MSE").to(dev)
m.train()
m.gemm_dtype = os.environ.get("GEMM", "fp32")
opt = FusedAdam(m, lr=1e-3)
perm = torch.randperm(50000, device=dev)
batches = [data[i * B:(i + 1) * B] for i in range(10)]
ident = [torch.arange(i * B, (i + 1) * B, device=dev) for i in range(10)]
rnd = [perm[i * B:(i + 1) * B].contiguous() for i in range(10)]


def timed(fn, n=60):
    for i in range(10):
        fn(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i)
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n

from distributed_vae_amd import _native as N
data16 = N.to_bf16(data) if m.gemm_dtype == "bf16" else None
mode = os.environ.get("MODE")      # one variant only (under rocprofv3 --kernel-trace --stats: per-kernel durations of that variant)
if mode:
    fn = {"fixed": lambda i: m.fused_train_step(batches[i % 10].expand(A, -1, -1), 1.0, opt, True),
          "identity": lambda i: m.fused_train_step_rows(data, ident[i % 10], 1.0, opt, True),
          "random": lambda i: m.fused_train_step_rows(data, rnd[i % 10], 1.0, opt, True),
          "storage16": lambda i: m.fused_train_step_rows(data, rnd[i % 10], 1.0, opt, True, data16=data16),
          "sorted": lambda i: m.fused_train_step_rows(data, rnd[i % 10].sort().values, 1.0, opt, True)}[mode]
    print("%s %.4f ms" % (mode, timed(fn)))
    sys.exit(0)
print("fixed batches          %.4f ms" % timed(lambda i: m.fused_train_step(batches[i % 10].expand(A, -1, -1), 1.0, opt, True)))
print("row map, identity      %.4f ms" % timed(lambda i: m.fused_train_step_rows(data, ident[i % 10], 1.0, opt, True)))
print("row map, random rows   %.4f ms" % timed(lambda i: m.fused_train_step_rows(data, rnd[i % 10], 1.0, opt, True)))
print("fixed batches          %.4f ms" % timed(lambda i: m.fused_train_step(batches[i % 10].expand(A, -1, -1), 1.0, opt, True)))
if data16 is not None:
    print("random rows, bf16 storage %.4f ms" % timed(lambda i: m.fused_train_step_rows(data, rnd[i % 10], 1.0, opt, True, data16=data16)))
    print("row map, random rows   %.4f ms" % timed(lambda i: m.fused_train_step_rows(data, rnd[i % 10], 1.0, opt, True)))
