"""TEST INFRASTRUCTURE ONLY -- emits tests/golden/*.npz from the REAL reference model.

Run in the build container only (needs /root/reference):

    python -m oracle.gen_golden

Every array below is produced by the reference's own ``mixVAE_model.forward`` /
``.loss`` (mmidas/nn_model.py:297, :495), ``loss.backward()`` and
``torch.optim.Adam`` (mmidas/cpl_mixvae.py:274, :434-463) on CPU, fp32, with the
noise it drew recorded (``oracle/ref_loader.explicit_noise``).  The fixtures hold
data only: inputs, recorded noise, parameters and expected outputs.
"""
from __future__ import annotations

import hashlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_loader as RL  # noqa: E402
from oracle import restatement as R  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")

CASES = {
    # name: (A, B, D, H, L, C, S, hard, s_drop)
    "tiny_a2": (2, 32, 64, 16, 5, 7, 2, False, 0.0),
    "tiny_a3_sdrop": (3, 48, 96, 16, 5, 11, 2, False, 0.2),
    "tiny_a5_hard": (5, 40, 96, 24, 6, 13, 2, True, 0.0),
    "ragged_a2": (2, 37, 52, 12, 3, 9, 1, False, 0.0),
}
MID = ("mid_a2", (2, 512, 1024, 100, 10, 92, 2, False, 0.0))
SEED = 546
N_TRAJ = 20
LR = 1e-3


def mk_ref(ref, cfg):
    A, B, D, H, L, C, S, hard, sdrop = cfg
    torch.manual_seed(SEED)
    m = ref.mixVAE_model(input_dim=D, fc_dim=H, n_categories=C, state_dim=S, lowD_dim=L,
                         x_drop=0.5, s_drop=sdrop, n_arm=A, lam=1, lam_pc=1, tau=0.005,
                         beta=1.0, hard=hard, variational=True, device="cpu", eps=1e-8,
                         momentum=0.01, ref_prior=False, loss_mode="MSE")
    h = R.Hyper(input_dim=D, fc_dim=H, n_categories=C, state_dim=S, lowD_dim=L, x_drop=0.5,
                s_drop=sdrop, n_arm=A, hard=hard)
    return m, h


def npy(t):
    return t.detach().cpu().numpy().copy()  # copy: state_dict() hands out live buffers


def pack_noise(d, prefix, noise):
    for k, lst in noise.items():
        if lst:
            d[f"{prefix}{k}"] = np.stack([npy(t) for t in lst])


def noise_digest(noise) -> str:
    hsh = hashlib.sha256()
    for k in sorted(noise):
        for t in noise[k]:
            hsh.update(npy(t).tobytes())
    return hsh.hexdigest()[:16]


def one_case(ref, name, cfg, full=True):
    A, B, D = cfg[0], cfg[1], cfg[2]
    m, h = mk_ref(ref, cfg)
    d = {"cfg": np.array(cfg[:7], dtype=np.int64), "hard": np.array(cfg[7]), "s_drop": np.array(cfg[8]),
         "seed": np.array(SEED)}
    x = R.synthetic_batch(B, D)
    xs = x.expand(A, -1, -1)
    if full:
        d["x"] = npy(x)
        for k, v in m.state_dict().items():
            d[f"sd0/{k}"] = npy(v)

    # ---- step 1: forward + loss + backward, the reference draws (and we record) its noise
    m.train()
    torch.manual_seed(SEED + 1)
    out, lo, noise1 = RL.reference_step(m, xs, 1.0)
    m.zero_grad()
    lo[0].backward()
    names = ["x_rec", None, None, "x_low", "c", "s_smp", "c_smp", "s_mean", "s_logvar", "c_prob"]
    if full:
        pack_noise(d, "noise1/", noise1)
        for i, nm in enumerate(names):
            if nm:
                d[f"fwd/{nm}"] = np.stack([npy(t) for t in out[i]])
        for k, p in m.named_parameters():
            d[f"grad/{k}"] = npy(p.grad)
        for k, v in m.state_dict().items():
            if "running" in k or "num_batches" in k:
                d[f"bn1/{k}"] = npy(v)
    else:
        d["noise1_seed"] = np.array(7)
    d["loss/total"] = npy(lo[0]); d["loss/rec"] = npy(lo[1]); d["loss/joint"] = npy(lo[2])
    d["loss/c_ent"] = npy(lo[3]); d["loss/c_dist"] = npy(lo[4]); d["loss/c_l2"] = npy(lo[5])
    d["loss/kl"] = np.array([float(v) for v in lo[6]], dtype=np.float32)
    d["loss/ll"] = np.array([float(v) for v in lo[8]], dtype=np.float32)
    return d, m, h, x, xs


def trajectory(ref, cfg, d, full_adam=True):
    """N_TRAJ Adam steps driven exactly as cpl_mixvae.py:434-463, seeded explicit noise."""
    A, B, D = cfg[0], cfg[1], cfg[2]
    m, h = mk_ref(ref, cfg)
    opt = torch.optim.Adam(m.parameters(), lr=LR)
    m.train()
    traj = []
    digests = []
    for step in range(N_TRAJ):
        x = R.synthetic_batch(B, D, seed=SEED + 100 + step)
        xs = x.expand(A, -1, -1)
        noise = R.draw_noise(h, B, seed=1000 + step)
        digests.append(noise_digest(noise))
        opt.zero_grad()
        out, lo, _ = RL.reference_step(m, xs, 1.0, noise)
        lo[0].backward()
        opt.step()
        traj.append([float(lo[0]), float(lo[2]), float(lo[3]), float(lo[4]), float(lo[5])]
                    + [float(v) for v in lo[1]])
        if step == 2:
            for k, p in m.named_parameters():
                st = opt.state[p]
                d[f"adam3/p/{k}"] = npy(p)
                if full_adam:
                    d[f"adam3/m/{k}"] = npy(st["exp_avg"])
                    d[f"adam3/v/{k}"] = npy(st["exp_avg_sq"])
    d["traj"] = np.array(traj, dtype=np.float64)
    d["traj_noise_digest"] = np.array(digests)
    for k, v in m.state_dict().items():
        if "running" in k:
            d[f"bnT/{k}"] = npy(v)


def eval_case(ref, cfg, d):
    """model.eval() + forward(eval=True): running-stat BN, no dropout, no Gumbel noise, hard."""
    A, B, D = cfg[0], cfg[1], cfg[2]
    m, h = mk_ref(ref, cfg)
    x = R.synthetic_batch(B, D)
    xs = x.expand(A, -1, -1)
    # a couple of training forwards so the running statistics are non-trivial
    m.train()
    with torch.no_grad():
        for s in range(2):
            RL.reference_step(m, xs, 1.0, R.draw_noise(h, B, seed=50 + s))
    for k, v in m.state_dict().items():
        if "running" in k:
            d[f"eval/sd/{k}"] = npy(v)
    m.eval()
    torch.manual_seed(SEED + 2)
    with torch.no_grad():
        out, lo, noise = RL.reference_step(m, xs, 1.0, None, eval_flag=True)
    pack_noise(d, "eval/noise/", noise)
    names = ["x_rec", None, None, "x_low", "c", "s_smp", "c_smp", "s_mean", "s_logvar", "c_prob"]
    for i, nm in enumerate(names):
        if nm:
            d[f"eval/fwd/{nm}"] = np.stack([npy(t) for t in out[i]])
    d["eval/loss_total"] = npy(lo[0])


def dp_case(ref, cfg, d, ws=2):
    """Virtual-rank data parallel oracle: same weights, ws disjoint
    batches through the reference separately (rank-local BN / variance statistics), gradients
    averaged, one Adam step."""
    A, B, D = cfg[0], cfg[1], cfg[2]
    grads = []
    for r in range(ws):
        m, h = mk_ref(ref, cfg)
        x = R.synthetic_batch(B, D, seed=SEED + 200 + r)
        noise = R.draw_noise(h, B, seed=2000 + r)
        m.train(); m.zero_grad()
        out, lo, _ = RL.reference_step(m, x.expand(A, -1, -1), 1.0, noise)
        lo[0].backward()
        grads.append({k: p.grad.clone() for k, p in m.named_parameters()})
        d[f"dp{ws}/loss_rank{r}"] = npy(lo[0])
    m, h = mk_ref(ref, cfg)
    opt = torch.optim.Adam(m.parameters(), lr=LR)
    for k, p in m.named_parameters():
        p.grad = sum(g[k] for g in grads) / ws
        d[f"dp{ws}/grad/{k}"] = npy(p.grad)
    opt.step()
    for k, p in m.named_parameters():
        d[f"dp{ws}/p/{k}"] = npy(p)


def main():
    torch.set_num_threads(4)
    ref = RL.load_reference_nn_model()
    os.makedirs(GOLDEN, exist_ok=True)
    for name, cfg in CASES.items():
        d, *_ = one_case(ref, name, cfg)
        trajectory(ref, cfg, d, full_adam=(name == "tiny_a2"))
        eval_case(ref, cfg, d)
        if name == "tiny_a2":
            dp_case(ref, cfg, d, ws=2)
        path = os.path.join(GOLDEN, name + ".npz")
        np.savez_compressed(path, **d)
        print(name, os.path.getsize(path) // 1024, "KiB", "loss", float(d["loss/total"]))
    # mid case: seeds + scalar results + per-parameter gradient norms only
    name, cfg = MID
    A, B, D = cfg[0], cfg[1], cfg[2]
    m, h = mk_ref(ref, cfg)
    x = R.synthetic_batch(B, D)
    noise = R.draw_noise(h, B, seed=7)
    m.train(); m.zero_grad()
    out, lo, _ = RL.reference_step(m, x.expand(A, -1, -1), 1.0, noise)
    lo[0].backward()
    d = {"cfg": np.array(cfg[:7], dtype=np.int64), "hard": np.array(cfg[7]), "s_drop": np.array(cfg[8]),
         "seed": np.array(SEED), "noise_seed": np.array(7), "noise_digest": np.array(noise_digest(noise))}
    d["loss/total"] = npy(lo[0]); d["loss/rec"] = npy(lo[1]); d["loss/joint"] = npy(lo[2])
    d["loss/c_ent"] = npy(lo[3]); d["loss/c_dist"] = npy(lo[4]); d["loss/c_l2"] = npy(lo[5])
    d["loss/kl"] = np.array([float(v) for v in lo[6]], dtype=np.float32)
    for k, p in m.named_parameters():
        g = p.grad.double()
        d[f"gnorm/{k}"] = np.array([float(g.norm()), float(g.sum()), float(g.abs().max())])
        if p.numel() <= 256:
            d[f"grad/{k}"] = npy(p.grad)
    for i, nm in [(4, "c"), (6, "c_smp"), (7, "s_mean"), (8, "s_logvar"), (3, "x_low")]:
        d[f"fwd/{nm}"] = np.stack([npy(t) for t in out[i]])[:, :16]
    d["fwd/x_rec_sum"] = np.array([float(t.double().sum()) for t in out[0]])
    path = os.path.join(GOLDEN, name + ".npz")
    np.savez_compressed(path, **d)
    print(name, os.path.getsize(path) // 1024, "KiB", "loss", float(d["loss/total"]))


if __name__ == "__main__":
    main()
