"""TEST INFRASTRUCTURE ONLY -- emits tests/golden/mask_a2.npz from the REAL reference model: the pruning-time forward
``mixVAE_model.forward(..., mask=kept_categories)`` (mmidas/nn_model.py:332-335: the second softmax over
``c_prob[:, mask]``, ``c`` zero elsewhere), in train mode (forward, loss, ``backward()``) and in eval mode (what
``cpl_mixVAE.eval_model`` runs on a pruned checkpoint, cpl_mixvae.py:1476-1478, :1524).

Run in the build container only (needs /root/reference):    python -m oracle.gen_golden_mask
The fixture holds data only: inputs, recorded noise, parameters, the mask and expected outputs.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import gen_golden as GG  # noqa: E402
from oracle import ref_loader as RL  # noqa: E402
from oracle import restatement as R  # noqa: E402

CFG = (2, 40, 72, 16, 5, 11, 2, False, 0.0)      # A, B, D, H, L, C, S, hard, s_drop
MASK = [0, 2, 3, 5, 8, 10]                        # kept categories
NAMES = ["x_rec", None, None, "x_low", "c", "s_smp", "c_smp", "s_mean", "s_logvar", "c_prob"]


def run(ref):
    A, B, D = CFG[:3]
    m, h = GG.mk_ref(ref, CFG)
    d = {"cfg": np.array(CFG[:7], dtype=np.int64), "hard": np.array(CFG[7]), "s_drop": np.array(CFG[8]),
         "mask": np.array(MASK, dtype=np.int64)}
    x = R.synthetic_batch(B, D, seed=77)
    xs = x.expand(A, -1, -1)
    d["x"] = GG.npy(x)
    for k, v in m.state_dict().items():
        d[f"sd0/{k}"] = GG.npy(v)
    mask = np.array(MASK)
    # train mode: forward(mask) + loss + backward, noise recorded
    m.train()
    torch.manual_seed(5)
    with RL.explicit_noise(m, None) as rec:
        out = m(xs, 1.0, 0.0, eval=False, mask=mask)
    lo = m.loss(out[0], [], [], xs, out[7], out[8], out[4], out[6], 0.0)
    m.zero_grad()
    lo[0].backward()
    GG.pack_noise(d, "noise1/", rec)
    for i, nm in enumerate(NAMES):
        if nm:
            d[f"fwd/{nm}"] = np.stack([GG.npy(t) for t in out[i]])
    for k, p in m.named_parameters():
        d[f"grad/{k}"] = GG.npy(p.grad)
    d["loss/total"] = GG.npy(lo[0]); d["loss/rec"] = GG.npy(lo[1]); d["loss/joint"] = GG.npy(lo[2])
    d["loss/c_ent"] = GG.npy(lo[3]); d["loss/c_dist"] = GG.npy(lo[4]); d["loss/c_l2"] = GG.npy(lo[5])
    for k, v in m.state_dict().items():
        if "running" in k or "num_batches" in k:
            d[f"sd1/{k}"] = GG.npy(v)
    # eval mode on the state the train step left (running statistics after one update): only u_state is drawn
    m.eval()
    with torch.no_grad(), RL.explicit_noise(m, None) as rec_e:
        out_e = m(xs, 1.0, 0.0, eval=True, mask=mask)
    GG.pack_noise(d, "noise_eval/", rec_e)
    for i, nm in enumerate(NAMES):
        if nm:
            d[f"eval/{nm}"] = np.stack([GG.npy(t) for t in out_e[i]])
    return d


if __name__ == "__main__":
    ref = RL.load_reference_nn_model()
    d = run(ref)
    path = os.path.join(GG.GOLDEN, "mask_a2.npz")
    np.savez_compressed(path, **d)
    print(path, os.path.getsize(path), "bytes;", "c row 0 arm 0:", d["fwd/c"][0, 0])
