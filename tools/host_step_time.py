"""Host-side cost of enqueueing one fused train step (no synchronisation inside the timed loop; the queue is empty at its start
and 20 steps of 0.7 ms fit any queue depth), against the device time of the same steps."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import distributed_vae_amd  # noqa
from distributed_vae_amd.cpl_mixvae import FusedAdam
from distributed_vae_amd.nn_model import mixVAE_model
dev = torch.device("cuda", 0)
A, B, D = int(os.environ.get("ARMS", "2")), 5000, 5000
data = bench.synthetic_rows(50000, D, 546, dev)
torch.manual_seed(546)
m = mixVAE_model(input_dim=D, fc_dim=100, n_categories=92, state_dim=2, lowD_dim=10, x_drop=0.5, s_drop=0.0, n_arm=A, lam=1, lam_pc=1,
                 tau=0.005, beta=1.0, hard=False, variational=True, device=dev, eps=1e-8, momentum=0.01, ref_prior=False, loss_mode="MSE").to(dev)
m.train()
m.gemm_dtype = os.environ.get("GEMM", "fp32")
opt = FusedAdam(m, lr=1e-3)
batches = [data[i * B:(i + 1) * B].expand(A, -1, -1) for i in range(10)]
for i in range(10):
    m.fused_train_step(batches[i % 10], 1.0, opt, True)
for n in (5, 20, 20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        m.fused_train_step(batches[i % 10], 1.0, opt, True)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("n = %2d: host enqueue %.1f us per step; with the final synchronize %.1f us per step" % (n, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
