"""Calibration: what the vendor fp32 GEMM (torch.matmul -> rocBLAS/hipBLASLt) reaches on the five D x H shapes."""
import torch, time
dev = "cuda:0"
B, D, H = 5000, 5000, 100
x = torch.randn(B, D, device=dev); w1 = torch.randn(H, D, device=dev); d10 = torch.randn(B, H, device=dev)
w11 = torch.randn(D, H, device=dev); dz = torch.randn(B, D, device=dev); dz1 = torch.randn(B, H, device=dev)
torch.backends.cuda.matmul.allow_tf32 = False
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
fl = 2.0 * B * D * H
for name, f in [("fc1  x @ W1^T      [5000x5000]x[5000x100]", lambda: x @ w1.t()),
                ("fc11 d10 @ W11^T   [5000x100]x[100x5000]", lambda: d10 @ w11.t()),
                ("gd10 dz @ W11      [5000x5000]x[5000x100]", lambda: dz @ w11),
                ("dW11 dz^T @ d10    [5000x5000]^T x[5000x100]", lambda: dz.t() @ d10),
                ("dW1  dz1^T @ x     [100x5000]x[5000x5000]", lambda: dz1.t() @ x)]:
    us = t(f)
    print(f"{name:48s} {us:8.1f} us  {fl / us / 1e6:6.1f} TFLOP/s")
