"""GPU tuning aid: time the augmenter forward at the benchmark shape (HIP events), per call and per GEMM layer."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import distributed_vae_amd  # noqa
from distributed_vae_amd.augmentation import Augmenter_smartseq
A, B, D, ND, NZ, Z = int(os.environ.get("AUG_A", 2)), 5000, int(os.environ.get("AUG_D", 5000)), 500, 50, 10
dev = torch.device("cuda", 0)
torch.manual_seed(1)
m = Augmenter_smartseq(NZ, Z, D, ND).to(dev).eval()
x = (torch.rand(B, D, device=dev) < 0.2).float() * torch.randn(B, D, device=dev).abs() * 3
xs = x.expand(A, -1, -1)
for _ in range(3): m(xs, True, 0.1)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 10
e0.record()
for _ in range(n): m(xs, True, 0.1)
e1.record(); e1.synchronize()
ms = e0.elapsed_time(e1) / n
n1, n5 = D // 5, ND // 5
trunk = D * n1 + n1 * n1 + n1 * ND + ND * ND + ND * n5
tail = n5 * ND + ND * ND + ND * n1 + n1 * n1 + n1 * D
fl = 2.0 * B * (trunk + A * tail)
fl_ref = 2.0 * A * B * (trunk + tail)
print(f"A={A} D={D}: {ms*1e3:.1f} us per call; executed {fl/1e9:.1f} GFLOP -> {fl/ms/1e9:.1f} TF; "
      f"reference call pattern {fl_ref/1e9:.1f} GFLOP -> {fl_ref/ms/1e9:.1f} TF-equivalent")
