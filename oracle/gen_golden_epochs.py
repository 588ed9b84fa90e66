"""TEST INFRASTRUCTURE ONLY -- emits tests/golden/epochs_a2.npz from the REAL reference TRAINER.

Run in the build container only (needs /root/reference):

    python -m oracle.gen_golden_epochs

The reference's own ``cpl_mixVAE.init_model`` / ``train`` / ``eval_model`` (mmidas/cpl_mixvae.py:193-286, :323-1448,
:1450-1619; class compiled in memory by ``oracle/ref_loader.load_reference_trainer``) run on CPU, fp32, for
N_EPOCH epochs of three 32-cell batches, with ``torch.optim.Adam``; every random draw of every forward is recorded
(``oracle/ref_loader.explicit_noise``) in call order.  What the reference computes per epoch is captured from the
dictionaries it hands to its logger (``run.log``: cpl_mixvae.py:536-553, :655-661, :768-775) -- the epoch means of
:485-492, the training / validation consensus and the validation losses -- and the dictionary ``eval_model`` returns.
The fixture holds data only: inputs, initial parameters, recorded noise, logged values, the returned dictionary, the
final parameters.
"""
from __future__ import annotations

import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_loader as RL  # noqa: E402
from oracle import restatement as R  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
A, B, D, H, L, C, S = 2, 32, 64, 16, 5, 7, 2
N_TRAIN, N_TEST, N_EPOCH, SEED, LR = 96, 32, 3, 546, 1e-3   # 128 rows: eval_model needs whole batches (np.array of ragged lists fails upstream)


class _Log:
    def __init__(self):
        self.rows = []

    def log(self, d):
        self.rows.append(dict(d))


def npy(t):
    return t.detach().cpu().numpy().copy()


def main():
    torch.set_num_threads(2)
    T = RL.load_reference_trainer()
    from torch.utils.data import DataLoader, TensorDataset
    x_all = R.synthetic_batch(N_TRAIN + N_TEST, D, seed=SEED + 7)
    x_tr, x_te = x_all[:N_TRAIN], x_all[N_TRAIN:]
    tr = DataLoader(TensorDataset(x_tr, torch.arange(N_TRAIN, dtype=torch.float32)), batch_size=B, shuffle=False, drop_last=True)
    te = DataLoader(TensorDataset(x_te, torch.arange(N_TEST, dtype=torch.float32)), batch_size=1, shuffle=False)
    al = DataLoader(TensorDataset(x_all, torch.arange(N_TRAIN + N_TEST, dtype=torch.float32)), batch_size=B, shuffle=False)
    d = {"cfg": np.array([A, B, D, H, L, C, S], dtype=np.int64), "n_epoch": np.array(N_EPOCH), "lr": np.array(LR),
         "x_train": npy(x_tr), "x_test": npy(x_te)}
    with tempfile.TemporaryDirectory() as folder:
        os.makedirs(os.path.join(folder, "model"))
        torch.manual_seed(SEED)
        t = T.cpl_mixVAE(saving_folder=folder, aug_file="", device="cpu", save_flag=True)
        t.init_model(n_categories=C, state_dim=S, input_dim=D, fc_dim=H, lowD_dim=L, x_drop=0.5, s_drop=0.0, lr=LR,
                     n_arm=A, temp=1.0, tau=0.005)
        for k, v in t.model.state_dict().items():
            d[f"sd0/{k}"] = npy(v)
        # record the noise of every forward, in call order
        calls = []
        fwd = t.model.forward

        def recording_forward(*a, **k):
            with RL.explicit_noise(t.model, None) as rec:
                out = fwd(*a, **k)
            calls.append((bool(t.model.training), rec))
            return out
        t.model.forward = recording_forward
        log = _Log()
        torch.manual_seed(SEED + 1)
        t.train(tr, te, n_epoch=N_EPOCH, n_epoch_p=0, rank="cpu", run=log, good_enuf_consensus=2.0)
        for k, v in t.model.state_dict().items():
            d[f"sdT/{k}"] = npy(v)
        saved = sorted(os.listdir(os.path.join(folder, "model")))
        d["saved_kinds"] = np.array(sorted({f.split("_A")[0] if "before_pruning" in f else f for f in saved if f.endswith(".pth")}))
        n_train_calls = len(calls)
        out = t.eval_model(al)
    # ---- what the reference logged per epoch (three dictionaries per epoch: train, train consensus, validation)
    keys = ["train/total-loss", "train/joint-loss", "train/negative-joint-entropy", "train/simplex-distance",
            "train/l2-distance", "train/consensus_aug", "train/rec-loss0", "train/rec-loss1", "train/consensus",
            "val/total-loss", "val/rec-loss", "val/consensus"]
    per_epoch = {k: [] for k in keys}
    for row in log.rows:
        for k in keys:
            if k in row:
                per_epoch[k].append(float(row[k]))
    for k in keys:
        assert len(per_epoch[k]) == N_EPOCH, (k, len(per_epoch[k]))
        d["epoch/" + k] = np.array(per_epoch[k], dtype=np.float64)
    # ---- recorded noise, call by call: training forwards carry x_mask / u_gumbel / u_state, eval forwards u_state only
    d["call_training"] = np.array([int(tr_) for tr_, _ in calls], dtype=np.int64)
    d["n_train_calls"] = np.array(n_train_calls)
    for i, (_, rec) in enumerate(calls):
        for k, lst in rec.items():
            if lst:
                d[f"noise/{i}/{k}"] = np.stack([npy(v) for v in lst])
    # ---- eval_model's dictionary (cpl_mixvae.py:1599-1619)
    for k, v in out.items():
        d["eval_model/" + k] = np.asarray(v)
    path = os.path.join(GOLDEN, "epochs_a2.npz")
    np.savez_compressed(path, **d)
    print("epochs_a2", os.path.getsize(path) // 1024, "KiB;", len(calls), "forwards recorded;",
          {k: np.round(v, 4).tolist() for k, v in per_epoch.items() if k in ("train/total-loss", "train/consensus", "val/rec-loss")})


if __name__ == "__main__":
    main()
