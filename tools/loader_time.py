import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import distributed_vae_amd  # noqa
from distributed_vae_amd.cpl_mixvae import cpl_mixVAE
from distributed_vae_amd.utils.dataloader import DeviceLoader
dev = torch.device("cuda", 0)
B, D, A = 5000, 5000, 2
data = (torch.rand(50000, D, device=dev) < 0.2).float()
tr = cpl_mixVAE(saving_folder="", device=dev, save_flag=False)
tr.init_model(n_categories=92, state_dim=2, input_dim=D, fc_dim=100, lowD_dim=10, x_drop=0.5, s_drop=0.0, n_arm=A)
ld = DeviceLoader(data, torch.arange(50000), B, True, True, seed=546)
def t(fn, n=3):
    fn(); torch.cuda.synchronize(); s0 = torch.cuda.memory_stats(); t0 = time.time(); per = []
    for _ in range(n):
        ta = time.time(); fn(); torch.cuda.synchronize(); per.append(round((time.time() - ta) * 1e3, 1))
    s1 = torch.cuda.memory_stats()
    print("   per epoch", per, "mallocs +%d frees +%d retries +%d reserved %.1f GB" % (
        s1["num_device_alloc"] - s0["num_device_alloc"], s1["num_device_free"] - s0["num_device_free"],
        s1["num_alloc_retries"] - s0["num_alloc_retries"], s1["reserved_bytes.all.current"] / 1e9))
    return (time.time() - t0) / n * 1e3
print("loader only  ms/epoch", t(lambda: [None for _ in ld]))
xs = [data[i * B:(i + 1) * B] for i in range(10)]
tr.pipeline = False
print("steps on fixed batches ms/epoch", t(lambda: [None for _ in tr.epoch_steps(xs)]))
print("steps from loader ms/epoch", t(lambda: [None for _ in tr.epoch_steps(ld)]))
tr.pipeline = True
print("pipelined from loader ms/epoch", t(lambda: [None for _ in tr.epoch_steps(ld)]))
print("pipelined fixed ms/epoch", t(lambda: [None for _ in tr.epoch_steps(xs)]))
