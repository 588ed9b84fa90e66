"""GPU tuning aid: a shuffled augmented epoch from the device-resident loader through the trainer (pipelined), benchmark shape;
MMVAE_ROWS=1 (the augmenter reads the rows out of the loader's planes) / 0 (gathered batches).  argv[1] = fp32 | bf16."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import distributed_vae_amd  # noqa
from distributed_vae_amd.augmentation import Augmenter_smartseq
from distributed_vae_amd.cpl_mixvae import cpl_mixVAE
from distributed_vae_amd.utils.dataloader import DeviceLoader
A, B, D, NR = 2, 5000, 5000, 50000
mode = sys.argv[1] if len(sys.argv) > 1 else "fp32"
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(546)
data = (torch.rand(NR, D, generator=g, device=dev) < 0.2).float() * torch.randn(NR, D, generator=g, device=dev).abs() * 3.0
torch.manual_seed(546)
tr = cpl_mixVAE(saving_folder="", device=dev, save_flag=False)
tr.init_model(n_categories=92, state_dim=2, input_dim=D, fc_dim=100, lowD_dim=10, x_drop=0.5, s_drop=0.0, n_arm=A, gemm_dtype=mode)
tr.set_augmenter(Augmenter_smartseq(50, 10, D, 500).to(dev).eval())
ld = DeviceLoader(data, torch.arange(NR), B, True, True, seed=546)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in tr.epoch_steps(ld):
    pass
e0.record(); n = 0
for _ in range(3):
    for _b in tr.epoch_steps(ld):
        n += 1
e1.record(); e1.synchronize()
print(f"{mode} MMVAE_ROWS={os.environ.get('MMVAE_ROWS', '1')} rows used: {tr.used_aug_rows}: {e0.elapsed_time(e1) / n:.4f} ms per augmented step", flush=True)
