import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import distributed_vae_amd  # noqa
from distributed_vae_amd.cpl_mixvae import cpl_mixVAE
from distributed_vae_amd.utils.dataloader import DeviceLoader
dev = torch.device("cuda", 0)
B, D, A = 5000, 5000, 2
import bench
data = bench.synthetic_rows(50000, D, 546, dev) if os.environ.get('BENCHDATA') else (torch.rand(50000, D, device=dev) < 0.2).float()
tr = cpl_mixVAE(saving_folder="", device=dev, save_flag=False)
tr.init_model(n_categories=92, state_dim=2, input_dim=D, fc_dim=100, lowD_dim=10, x_drop=0.5, s_drop=0.0, n_arm=A)
tr.pipeline = os.environ.get('PIPE','1') == '1'
ld = DeviceLoader(data, torch.arange(50000), B, True, True, seed=546)
for _ in tr.epoch_steps(ld): pass
torch.cuda.synchronize()
for ep in range(8):
    s0 = torch.cuda.memory_stats(); t0 = time.time(); marks = []
    for buf in tr.epoch_steps(ld):
        marks.append(time.time())
    t1 = time.time(); torch.cuda.synchronize(); t2 = time.time(); s1 = torch.cuda.memory_stats()
    print(f"epoch {ep}: host {1e3*(t1-t0):.2f} ms, total {1e3*(t2-t0):.2f} ms, mallocs +{s1['num_device_alloc']-s0['num_device_alloc']}, per-iter host ms:",
          [round(1e3 * (b - a), 2) for a, b in zip([t0] + marks[:-1], marks)])
