"""GPU diagnostic: host time per fused step (enqueue only) and per loader batch, against the GPU time of the step."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import distributed_vae_amd  # noqa
from distributed_vae_amd.cpl_mixvae import cpl_mixVAE
from distributed_vae_amd.utils.dataloader import DeviceLoader
A, B, D = 2, 5000, 5000
dev = torch.device("cuda", 0)
data = torch.rand(50000, D, device=dev)
tr = cpl_mixVAE(saving_folder="", device=dev, save_flag=False)
tr.init_model(n_categories=92, state_dim=2, input_dim=D, fc_dim=100, lowD_dim=10, x_drop=0.5, s_drop=0.0, n_arm=A)
x = data[:B].contiguous()
xs = x.expand(A, -1, -1)
for _ in range(5):
    tr._step(xs)
torch.cuda.synchronize()
n = 200
t0 = time.perf_counter()
for _ in range(n):
    tr._step(xs)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"fused step: host enqueue {(t1 - t0) / n * 1e6:.0f} us per step; GPU-limited total {(t2 - t0) / n * 1e6:.0f} us per step")
ld = DeviceLoader(data, torch.arange(data.shape[0]), B, True, True, seed=546)
it = iter(ld)
next(it)
torch.cuda.synchronize()
t0 = time.perf_counter()
k = 0
for b in it:
    k += 1
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"loader: host {(t1 - t0) / max(k, 1) * 1e6:.0f} us per batch ({k} batches)")
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    for _b in tr.epoch_steps(ld):
        pass
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
