"""GPU tuning aid: the augmented train step (augmenter of batch i + 1 on the produce stream beside train step i) at the benchmark
shape; argv[1] = fp32 | bf16.  Prints isolated augmenter, isolated step, back-to-back and pipelined times."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import distributed_vae_amd  # noqa
from distributed_vae_amd.augmentation import Augmenter_smartseq
from distributed_vae_amd.cpl_mixvae import cpl_mixVAE
A, B, D = 2, 5000, 5000
mode = sys.argv[1] if len(sys.argv) > 1 else "fp32"
dev = torch.device("cuda", 0)
torch.manual_seed(546)
g = torch.Generator(device=dev).manual_seed(546)
data = (torch.rand(10 * B, D, generator=g, device=dev) < 0.2).float() * torch.randn(10 * B, D, generator=g, device=dev).abs() * 3.0
batches = [data[i * B:(i + 1) * B] for i in range(10)]
net = Augmenter_smartseq(50, 10, D, 500).to(dev).eval()
net.gemm_dtype = mode
tr = cpl_mixVAE(saving_folder="", device=dev, save_flag=False)
tr.init_model(n_categories=92, state_dim=2, input_dim=D, fc_dim=100, lowD_dim=10, x_drop=0.5, s_drop=0.0, n_arm=A, gemm_dtype=mode)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
res = {}
for key, aug, pipe in (("step alone", False, True), ("back to back", True, False), ("pipelined", True, True)):
    tr.set_augmenter(net if aug else None)
    tr.pipeline = pipe
    for _ in tr.epoch_steps(batches[:3]):
        pass
    e0.record(); n = 0
    for _ in range(3):
        for _b in tr.epoch_steps(batches):
            n += 1
    e1.record(); e1.synchronize()
    res[key] = e0.elapsed_time(e1) / n
for _ in range(3):
    net(batches[0].expand(A, -1, -1), True, 0.1)
e0.record()
for b in batches:
    net(b.expand(A, -1, -1), True, 0.1)
e1.record(); e1.synchronize()
res["augmenter alone"] = e0.elapsed_time(e1) / len(batches)
print(mode, " ".join(f"{k}: {v:.3f} ms" for k, v in res.items()), flush=True)
