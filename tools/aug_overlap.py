"""Experiment: augmenter forward of batch i+1 on a second stream beside the train step of batch i (both frozen-weight
independent), against running them back to back.  Prints ms per (augment + step) pair."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import distributed_vae_amd  # noqa
from distributed_vae_amd.augmentation import Augmenter_smartseq
from distributed_vae_amd.cpl_mixvae import FusedAdam
from distributed_vae_amd.nn_model import mixVAE_model
A, B, D, H, L, C, S = 2, 5000, 5000, 100, 10, 92, 2
dev = torch.device("cuda", 0)
torch.manual_seed(546)
m = mixVAE_model(input_dim=D, fc_dim=H, n_categories=C, state_dim=S, lowD_dim=L, x_drop=0.5, s_drop=0.0, n_arm=A, lam=1, lam_pc=1, tau=0.005, beta=1.0, hard=False, variational=True, device=dev, eps=1e-8, momentum=0.01, ref_prior=False, loss_mode="MSE").to(dev)
m.train(); opt = FusedAdam(m, lr=1e-3)
net = Augmenter_smartseq(50, 10, D, 500).to(dev).eval()
xb = [(torch.rand(B, D, device=dev) < 0.2).float() * torch.randn(B, D, device=dev).abs() * 3 for _ in range(4)]
main = torch.cuda.current_stream()
side = torch.cuda.Stream(priority=int(os.environ.get("AUG_PRIO", "0")))
def serial(n):
    for i in range(n):
        xs = net(xb[i % 4].expand(A, -1, -1), True, 0.1)[1]
        m.fused_train_step(xs, 1.0, opt, do_adam=True)
def piped(n):
    with torch.cuda.stream(side):
        cur = net(xb[0].expand(A, -1, -1), True, 0.1)[1]
        ev = torch.cuda.Event(); ev.record(side)
    for i in range(n):
        with torch.cuda.stream(side):
            nxt = net(xb[(i + 1) % 4].expand(A, -1, -1), True, 0.1)[1]
            ev2 = torch.cuda.Event(); ev2.record(side)
        main.wait_event(ev)
        m.fused_train_step(cur, 1.0, opt, do_adam=True)
        cur.record_stream(main)
        cur, ev = nxt, ev2
    main.wait_event(ev)
for fn in (serial, piped, serial, piped):
    fn(3); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(20); e1.record(); torch.cuda.synchronize()
    print(fn.__name__, round(e0.elapsed_time(e1) / 20, 3), "ms per augment + step")
