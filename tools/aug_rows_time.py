"""GPU tuning aid: Augmenter_smartseq.forward on gathered batches against forward_rows on the resident matrix' planes, benchmark shape."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import distributed_vae_amd  # noqa
from distributed_vae_amd import _native as N
from distributed_vae_amd.augmentation import Augmenter_smartseq
A, B, D, NR = 2, 5000, 5000, 50000
mode = sys.argv[1] if len(sys.argv) > 1 else "fp32"
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(546)
data = (torch.rand(NR, D, generator=g, device=dev) < 0.2).float() * torch.randn(NR, D, generator=g, device=dev).abs() * 3.0
net = Augmenter_smartseq(50, 10, D, 500).to(dev).eval()
net.gemm_dtype = mode
planes = N.tp_planes(data, net.planes_needed())
perm = torch.randperm(NR, device=dev)
rows = [perm[i * B:(i + 1) * B].contiguous() for i in range(10)]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
buf = torch.empty(B, D, device=dev)
def gathered(r):
    N.gather_rows(data, r, buf)
    net(buf.expand(A, -1, -1), True, 0.1)
def mapped(r):
    net.forward_rows(planes, NR, r, A, 0.1)
for name, fn in (("gather + forward", gathered), ("forward_rows", mapped)):
    for r in rows[:3]:
        fn(r)
    e0.record()
    for _ in range(3):
        for r in rows:
            fn(r)
    e1.record(); e1.synchronize()
    print(f"{mode} {name}: {e0.elapsed_time(e1) / 30:.4f} ms per batch", flush=True)
