"""GPU tuning aid: augmenter forward at the benchmark shape (HIP events); argv[1] = fp32 | bf16."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import distributed_vae_amd  # noqa
from distributed_vae_amd.augmentation import Augmenter_smartseq
A, B, D = 2, 5000, 5000
dev = torch.device("cuda", 0)
torch.manual_seed(546)
net = Augmenter_smartseq(50, 10, D, 500).to(dev).eval()
net.gemm_dtype = sys.argv[1] if len(sys.argv) > 1 else "fp32"
x = torch.rand(B, D, device=dev)
for _ in range(3):
    net(x.expand(A, -1, -1), True, 0.1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    net(x.expand(A, -1, -1), True, 0.1)
e1.record(); e1.synchronize()
ms = e0.elapsed_time(e1) / 10
n1, n5 = D // 5, 100
trunk = D * n1 + n1 * n1 + n1 * 500 + 500 * 500 + 500 * n5
tail = n5 * 500 + 500 * 500 + 500 * n1 + n1 * n1 + n1 * D
print(f"{net.gemm_dtype}: {ms:.3f} ms per batch, {2.0 * B * (trunk + A * tail) / ms / 1e9:.1f} TFLOP/s executed", flush=True)
