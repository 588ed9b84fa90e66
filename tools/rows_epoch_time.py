"""Shuffled epochs through the trainer (row-indexed steps) against the same steps on fixed batches; with TRACE=1 only the
shuffled epochs run (for rocprofv3 --kernel-trace: tools/kstats.py lists what is launched besides the step's kernels)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import distributed_vae_amd  # noqa
from distributed_vae_amd.cpl_mixvae import cpl_mixVAE
from distributed_vae_amd.utils.dataloader import DeviceLoader
dev = torch.device("cuda", 0)
A, B, D = 2, 5000, 5000
data = bench.synthetic_rows(50000, D, 546, dev)
tr = cpl_mixVAE(saving_folder="", device=dev, save_flag=False)
tr.init_model(n_categories=92, state_dim=2, input_dim=D, fc_dim=100, lowD_dim=10, x_drop=0.5, s_drop=0.0, n_arm=A)
tr.model.train()
ld = DeviceLoader(data, torch.arange(50000), B, True, True, seed=546)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def epochs(n_ep, fn):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record()
    n = 0
    for _ in range(n_ep):
        n += fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n, (time.perf_counter() - t0) * 1e3 / n


def shuffled():
    k = 0
    for _ in tr.epoch_steps(ld):
        k += 1
    return k


fixed = [data[i * B:(i + 1) * B] for i in range(10)]


def fixed_epoch():
    for b in fixed:
        tr.train_step(b)
    return 10


if os.environ.get("TRACE"):
    print("shuffled epoch %.4f ms per step (host %.4f)" % epochs(3, shuffled))
else:
    for _ in range(2):
        print("fixed batches  %.4f ms per step (host %.4f)" % epochs(6, fixed_epoch))
        print("shuffled epoch %.4f ms per step (host %.4f)" % epochs(6, shuffled))
