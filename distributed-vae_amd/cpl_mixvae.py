"""Drop-in for the hot path of the reference trainer ``mmidas/cpl_mixvae.py::cpl_mixVAE``.

Mirrors ``__init__`` (:153-186), ``init_model`` (:193-286), ``load_model`` (:317-321) and the
training loop of ``train`` (:323-492: per-batch driver :415-478, epoch reductions :480-492,
checkpoints :777-788) on top of the fused HIP train step.  Out of the hot path and therefore not
here: pruning phase, wandb/matplotlib reporting.  The augmenter in front of the step
(:182-186, :422-423) is ``distributed_vae_amd.augmentation``.
"""
from __future__ import annotations

import os
import time
from typing import Optional

import numpy as np
import torch
from torch import nn

from . import _native as N
from . import dist as D
from .nn_model import VAEConfig, mixVAE_model  # noqa: F401


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam / AdamW semantics (cpl_mixvae.py:274, train.py:144-147) as ONE HIP kernel over
    the model's flat parameter buffer.  ``state_dict()`` is torch.optim.Adam-compatible so reference
    checkpoints (``optimizer_state_dict``, cpl_mixvae.py:783-786) load and save unchanged."""

    def __init__(self, model: mixVAE_model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
                 decoupled=False):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False, maximize=False,
                        foreach=None, capturable=False, differentiable=False, fused=None)
        super().__init__(list(model.parameters()), defaults)
        self.model = model
        self.decoupled = bool(decoupled)
        self.step_count = 0
        self.exp_avg = self.exp_avg_sq = None

    def _bind(self, model=None):
        flat = self.model.flat_parameters()
        if self.exp_avg is None or self.exp_avg.device != flat.device or self.exp_avg.numel() != flat.numel():
            old = (self.exp_avg, self.exp_avg_sq)
            self.exp_avg = torch.zeros_like(flat)
            self.exp_avg_sq = torch.zeros_like(flat)
            if old[0] is not None and old[0].numel() == flat.numel():
                self.exp_avg.copy_(old[0])
                self.exp_avg_sq.copy_(old[1])

    @torch.no_grad()
    def step(self, closure=None):
        self._bind()
        self.step_count += 1
        g = self.param_groups[0]
        N.adam_step(self.model.flat_parameters(), self.model.flat_grad(), self.exp_avg, self.exp_avg_sq,
                    self.step_count, g["lr"], g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"],
                    self.decoupled)

    def zero_grad(self, set_to_none: bool = False):
        self.model.flat_grad().zero_()

    def _views(self, buf):
        lay, A = self.model._layout, self.model.n_arm
        out = {}
        for a in range(A):
            for t in range(N.N_PARAM_TENSORS):
                p = self.model._param_of(t, a)
                o = a * int(lay.per_arm) + int(lay.offset[t])
                out[id(p)] = buf[o: o + p.numel()].view(p.shape)
        return [out[id(p)] for p in self.model.parameters()]

    def state_dict(self):
        self._bind()
        m, v = self._views(self.exp_avg), self._views(self.exp_avg_sq)
        n = len(m)
        state = {}
        if self.step_count > 0:
            for i in range(n):
                state[i] = {"step": torch.tensor(float(self.step_count)), "exp_avg": m[i].clone(),
                            "exp_avg_sq": v[i].clone()}
        groups = []
        for g in self.param_groups:
            gg = {k: v_ for k, v_ in g.items() if k != "params"}
            gg["params"] = list(range(n))
            groups.append(gg)
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        self._bind()
        m, v = self._views(self.exp_avg), self._views(self.exp_avg_sq)
        for i, st in sd.get("state", {}).items():
            i = int(i)
            m[i].copy_(st["exp_avg"])
            v[i].copy_(st["exp_avg_sq"])
            self.step_count = int(float(st["step"]))
        for g, src in zip(self.param_groups, sd.get("param_groups", [])):
            for k in ("lr", "betas", "eps", "weight_decay"):
                if k in src:
                    g[k] = tuple(src[k]) if k == "betas" else src[k]


def get_device(device=None) -> torch.device:
    """cpl_mixvae.py:110-125."""
    if device in ("cpu", "mps"):
        return torch.device(device)
    if device == "cuda":
        return torch.device("cuda")
    if isinstance(device, int):
        torch.cuda.set_device(device)
        return torch.device("cuda", device)
    if isinstance(device, torch.device):
        return device
    if device is None:
        return torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")
    return torch.device(device)


class cpl_mixVAE:
    def __init__(self, saving_folder="", aug_file="", device=None, eps=1e-8, save_flag=True, load_weights=True):
        self.eps = eps
        self.save = save_flag
        self.folder = saving_folder
        self.aug_file = aug_file
        self.models = []
        self.device = get_device(device)
        self.aug_model, self.aug_param, self.netA = None, None, None
        self.pipeline = os.environ.get("MMVAE_PIPELINE", "1") != "0"   # see epoch_steps
        # bf16 configuration with a device-resident loader: the step reads x through a bf16 COPY of the matrix and keeps dZ11 as
        # bf16 (the reconstruction loss then sees bf16-rounded x; +50 % resident memory).  An attribute of the trainer (default
        # from MMVAE_BF16_STORAGE, on), recorded per run in ``used_bf16_storage`` and in train()'s history.
        self.bf16_storage = os.environ.get("MMVAE_BF16_STORAGE", "1") != "0"
        self.used_bf16_storage = False
        if aug_file:                                            # cpl_mixvae.py:182-186
            from .augmentation import mk_augmenter
            self.aug_model, self.aug_param, netA = mk_augmenter(aug_file, load_weights)
            self.set_augmenter(netA)

    def set_augmenter(self, netA):
        """Install an ``Augmenter_smartseq`` (distributed_vae_amd.augmentation) in eval mode on the trainer's device,
        as cpl_mixvae.py:184; ``None`` returns to raw ``x.expand``."""
        self.netA = None if netA is None else netA.to(self.device).eval()
        if self.netA is not None and getattr(self, "model", None) is not None:
            self.netA.gemm_dtype = self.model.gemm_dtype     # the bf16 configuration covers the augmenter's GEMMs too

    def init_model(self, n_categories, state_dim, input_dim, fc_dim=100, lowD_dim=10, x_drop=0.5, s_drop=0.2,
                   lr=0.001, lam=1, lam_pc=1, n_arm=2, temp=1.0, tau=0.005, beta=1.0, hard=False, variational=True,
                   ref_prior=False, trained_model="", n_pr=0, momentum=0.01, mode="MSE", gemm_dtype="fp32"):
        """cpl_mixvae.py:193-286.  ``gemm_dtype`` (not in the reference): "fp32", or "bf16" for BASELINE.json's bf16
        configuration (bf16 operands in the five D x H GEMMs, fp32 everywhere else)."""
        self.lowD_dim = lowD_dim
        self.n_categories = n_categories
        self.state_dim = state_dim
        self.input_dim = input_dim
        self.temp = temp
        self.n_arm = n_arm
        self.fc_dim = fc_dim
        self.ref_prior = ref_prior
        self.model = mixVAE_model(input_dim=input_dim, fc_dim=fc_dim, n_categories=n_categories,
                                  state_dim=state_dim, lowD_dim=lowD_dim, x_drop=x_drop, s_drop=s_drop, n_arm=n_arm,
                                  lam=lam, lam_pc=lam_pc, tau=tau, beta=beta, hard=hard, variational=variational,
                                  device=self.device, eps=self.eps, ref_prior=ref_prior, momentum=momentum,
                                  loss_mode=mode)
        self.model = self.model.to(self.device)
        self.model.gemm_dtype = gemm_dtype
        if self.netA is not None:
            self.netA.gemm_dtype = gemm_dtype
        self.optimizer = FusedAdam(self.model, lr=lr)
        if len(trained_model) > 0:
            loaded = torch.load(trained_model, map_location="cpu", weights_only=True)
            self.model.load_state_dict(loaded["model_state_dict"])
            self.optimizer.load_state_dict(loaded["optimizer_state_dict"])
            self.init = False
            self.n_pr = n_pr
        else:
            self.init = True
            self.n_pr = 0

    def load_model(self, trained_model):
        """cpl_mixvae.py:317-321."""
        loaded = torch.load(trained_model, map_location="cpu", weights_only=True)
        self.model.load_state_dict(loaded["model_state_dict"])
        self.current_time = time.strftime("%Y-%m-%d-%H-%M-%S")

    def save_checkpoint(self, path):
        """Checkpoint dict of cpl_mixvae.py:783-786."""
        torch.save({"model_state_dict": self.model.state_dict(),
                    "optimizer_state_dict": self.optimizer.state_dict()}, path)

    # ---------------------------------------------------------------------------------------
    def _step(self, xs: torch.Tensor):
        """zero_grad / forward / loss / backward / optimizer.step of cpl_mixvae.py:434-463 for one batch.  With the
        trainer's own ``FusedAdam`` the update rides on the gradient reduction of the fused step; any other
        ``torch.optim`` optimizer assigned to ``self.optimizer`` (the reference pattern ``cplMixVAE.optimizer =
        optim.Adam(model.parameters())``, train.py:144-147) gets the gradients of the fused step through each
        parameter's ``.grad`` and steps itself."""
        if D.is_dist():
            return D.dp_train_step(self.model, xs, self.temp, self.optimizer)
        if isinstance(self.optimizer, FusedAdam) and self.optimizer.model is self.model:
            return self.model.fused_train_step(xs, self.temp, self.optimizer, do_adam=True)
        buf = self.model.fused_train_step(xs, self.temp, None, do_adam=False)
        self.model.bind_grads()
        self.optimizer.step()
        return buf

    def _step_rows(self, data: torch.Tensor, rows: torch.Tensor, data16=None):
        """``_step`` on the batch ``data[rows]`` read in place (``mixVAE_model.fused_train_step_rows``); ``data16``: the
        matrix's bf16 copy for the bf16 configuration (bf16 storage)."""
        if D.is_dist():
            return D.dp_train_step(self.model, None, self.temp, self.optimizer, rows=(data, rows, data16))
        if isinstance(self.optimizer, FusedAdam) and self.optimizer.model is self.model:
            return self.model.fused_train_step_rows(data, rows, self.temp, self.optimizer, do_adam=True, data16=data16)
        buf = self.model.fused_train_step_rows(data, rows, self.temp, None, do_adam=False, data16=data16)
        self.model.bind_grads()
        self.optimizer.step()
        return buf

    def train_step(self, x: torch.Tensor):
        """One batch of cpl_mixvae.py:416-463 (x -> device, x.expand over arms, [augmenter,] zero_grad, forward,
        loss, backward, optimizer step).  Returns the device loss vector; no host synchronisation."""
        x = x.to(self.device, non_blocking=True)
        xs = x.expand(self.n_arm, -1, -1)
        if self.netA is not None:
            xs = self.netA(xs, True, 0.1)[1]                    # cpl_mixvae.py:422-423
        return self._step(xs)

    def epoch_steps(self, loader):
        """Generator over the loss vectors of one pass over ``loader`` (the inner loop of cpl_mixvae.py:415-478).

        The loop is software-pipelined over two HIP streams: batch i+1 is *produced* -- fetched from the loader (row
        gather of the device-resident loaders, or the H2D copy of a host batch) and, with an augmenter, passed through
        the frozen eval-mode augmenter -- on a high-priority side stream beside the train step of batch i.  The
        augmenter's GEMMs fill the matrix pipe while the step sits in its latency-bound chain kernels (measured at the
        benchmark shape: 2.93 ms per augmented batch against 3.21 ms back to back).  Same arithmetic and the same order
        of random draws as the unpipelined loop (``pipeline = False`` / MMVAE_PIPELINE=0)."""
        A = self.n_arm

        def first(b):
            return b[0] if isinstance(b, (tuple, list)) else b

        # A device-resident loader without an augmenter in front of the step: the batch is never assembled -- the step reads the
        # resident matrix through the epoch's row indices (mmvae_train_step_rows: fc1, the fused fc11 kernel and dW1 take a
        # row map), so a shuffled epoch costs what fixed batches do (the 100 MB row gather per step, and the hook that hid it
        # beside the encoder chain, are gone from this path).  Where the library does not offer it (other GEMM engines,
        # matrices beyond 4 GB) the first step says so and the epoch falls back to gathered batches.
        if (self.netA is None and self.device.type == "cuda" and hasattr(loader, "iter_rows")
                and getattr(loader, "data", None) is not None and os.environ.get("MMVAE_ROWS", "1") != "0"
                and getattr(self, "_rows_ok", True)):
            it = loader.iter_rows()
            first_rows = next(it, None)
            if first_rows is None:
                return
            # the bf16 configuration on bf16 storage: the loader keeps a bf16 copy of its matrix (made
            # once) and the GEMMs read that; MMVAE_BF16_STORAGE=0 keeps them on the fp32 matrix
            data16 = None
            if getattr(self.model, "gemm_dtype", "fp32") == "bf16" and hasattr(loader, "data_bf16") and self.bf16_storage:
                data16 = loader.data_bf16()
            self.used_bf16_storage = data16 is not None
            try:
                buf = self._step_rows(loader.data, first_rows, data16)
            except NotImplementedError:
                self._rows_ok = False
                if hasattr(loader, "unread_epoch"):
                    loader.unread_epoch()            # the abandoned iterator had taken this epoch's permutation
            else:
                yield buf
                for rows in it:
                    yield self._step_rows(loader.data, rows, data16)
                return
        if self.device.type != "cuda" or not self.pipeline:
            for b in loader:
                yield self.train_step(first(b))
            return
        # ring buffers (loader batches, augmenter outputs) are reused every few batches: batch k may only be produced once
        # the step that read its slot last (step k - lag) is over; a one-slot ring cannot be pipelined at all
        aug_ring = 3
        lring = int(getattr(loader, "ring", 0))
        lag = min(lring if lring > 0 else aug_ring, aug_ring)
        if lag < 2:
            for b in loader:
                yield self.train_step(first(b))
            return
        main = torch.cuda.current_stream(self.device)
        side = N.shared_stream(self.device, "produce")
        # With an augmenter in front of the step and a device-resident loader the batch is not assembled either: the loader keeps
        # its matrix as the augmenter engine's slice planes (made once) and the augmenter's first layer reads the epoch's rows in
        # place (Augmenter_smartseq.forward_rows: no row gather, no per-batch conversion; same results bit for bit).
        # MMVAE_ROWS=0 keeps gathered batches.
        planes, np_aug = None, 0
        if (self.netA is not None and hasattr(loader, "iter_rows") and hasattr(loader, "data_planes")
                and getattr(loader, "data", None) is not None and os.environ.get("MMVAE_ROWS", "1") != "0"
                and hasattr(self.netA, "planes_needed")):
            np_aug = self.netA.planes_needed()
            if np_aug and loader.data.shape[1] == self.netA._dims[0]:
                planes = loader.data_planes(np_aug)
        self.used_aug_rows = planes is not None
        it = loader.iter_rows() if planes is not None else iter(loader)
        aug_out = {}
        done = {}                                               # step index -> main-stream event at its end
        count = [0]

        def produce():
            k = count[0]
            if k - lag in done:
                side.wait_event(done.pop(k - lag))
            b = next(it, None)
            if b is None:
                return None
            if planes is not None:
                key = (k % aug_ring, b.shape[0])
                if key not in aug_out:
                    aug_out[key] = (torch.empty(A, b.shape[0], self.netA._dims[4], device=self.device),
                                    torch.empty(A, b.shape[0], loader.data.shape[1], device=self.device))
                xs = self.netA.forward_rows(planes, loader.data.shape[0], b, A, 0.1, out=aug_out[key])[1]   # cpl_mixvae.py:422-423
                count[0] += 1
                ev = torch.cuda.Event()
                ev.record(side)
                return xs, b, ev
            x = first(b).to(self.device, non_blocking=True)
            xs = x.expand(A, -1, -1)
            if self.netA is not None:
                key = (k % aug_ring, x.shape[0])
                if key not in aug_out:                          # persistent outputs: no allocator traffic per batch
                    aug_out[key] = (torch.empty(A, x.shape[0], self.netA._dims[4], device=self.device),
                                    torch.empty(A, x.shape[0], x.shape[1], device=self.device))
                xs = self.netA(xs, True, 0.1, out=aug_out[key])[1]   # cpl_mixvae.py:422-423
            count[0] += 1
            ev = torch.cuda.Event()
            ev.record(side)
            return xs, x, ev

        side.wait_stream(main)                                  # parameters / loader state written on the main stream
        with torch.cuda.stream(side):
            cur = produce()
        # Without an augmenter the production of a batch is a copy (row gather of a loader the row-indexed step above does
        # not cover, or the H2D copy of a host batch): it is issued BEHIND the step's call, so that it reaches the device
        # some 300 us into the step instead of beside fc1 (measured at the benchmark shape: 0.75 against 0.80 ms per
        # shuffled step).  With an augmenter the production is a 1.7 ms chain of GEMMs: it starts in front of the step.
        late = self.netA is None and not D.is_dist()
        k = 0
        while cur is not None:
            if not late:
                with torch.cuda.stream(side):
                    nxt = produce()
            xs, x, ev = cur
            main.wait_event(ev)
            buf = self._step(xs)
            if late:
                with torch.cuda.stream(side):
                    nxt = produce()
            x.record_stream(main)                               # produced on the side stream, read on the main one
            fin = torch.cuda.Event()
            fin.record(main)
            done[k] = fin
            done.pop(k - 2 * lag, None)
            k += 1
            yield buf
            cur = nxt

    def train(self, train_loader, test_loader, n_epoch, n_epoch_p=0, c_p=0, c_onehot=0, min_con=0.5,
              max_prun_it=0, rank=None, run=None, ws=1, good_enuf_consensus=0.75):
        """Training loop of cpl_mixvae.py:397-492, the per-epoch consensus on the training set (:563-657), the
        validation block (:665-775), the 10-epoch checkpoint (:777-788) and the stop at ``good_enuf_consensus``
        (:851-927: checkpoint ``cns_cpl_mixVAE_model_before_pruning_A{A}_...pth``, then ``break``).  Pruning
        (:996-1444) is disabled upstream and not offered.  Returns the per-epoch history (the reference returns None and
        hands the same numbers to its logger ``run``)."""
        A, Dm = self.n_arm, self.input_dim
        dev = self.device
        if not self.init:
            return {}
        if self.n_arm == 1:
            raise ZeroDivisionError("division by zero")   # nn_model.py:592-594
        if D.is_dist():
            D.broadcast_flat(self.model.flat_parameters())
        hist = {"losses": [], "loss_joints": [], "loss_recs": [[] for _ in range(A)], "c_ents": [], "c_l2_dists": [],
                "c_dists": [], "validation_loss": [], "validation_rec_loss": [], "consensus_train": [], "consensus_aug": [],
                "consensus_val": [], "epoch_times": [], "stopped_at": None}
        self.current_time = time.strftime("%Y-%m-%d-%H-%M-%S")
        # the reference classifies the training set batch by batch only when the TEST loader's batch size exceeds one,
        # otherwise the whole ``train_loader.dataset.tensors`` (:567 / :617): every row, no shuffle consumed
        whole_train = test_loader is not None and not ((getattr(test_loader, "batch_size", None) or 0) > 1)
        from ._utils import confmat_counts, consensus_from_counts
        E = n_epoch
        for e in range(E):
            t0 = time.time()
            self.model.train()
            acc = torch.zeros(5 + 3 * A, dtype=torch.float32, device=dev)   # sums of the loss vector
            counts_aug = confmat_counts(A, self.n_categories, dev)            # labels of the TRAINING forwards (:510-525)
            nb = 0
            for buf in self.epoch_steps(train_loader):
                acc += buf                                                   # :469-475 without .item()
                nb += 1
                eng = self.model._engine
                N.confmat_accumulate(N.classify(eng.ws_view("c", self.n_categories)), self.n_categories, counts_aug)
            red = torch.cat([acc, torch.tensor([float(nb)], device=dev)])
            both = torch.stack([D.allreduce_sum_(red.clone()), red]).cpu().numpy()   # :480-483 folded into one
            red, loc = both[0], both[1]
            nsteps = red[-1]
            Bs = max(nb, 1)                                                  # len(train_loader) of this rank
            hist["losses"].append(red[N.LOSS_TOTAL] / nsteps)                # :485 (all-reduced sum / all-reduced count)
            hist["loss_joints"].append(loc[N.LOSS_JOINT] / Bs)               # :486-488: rank-local sums / len(loader)
            hist["c_ents"].append(loc[N.LOSS_CENT] / Bs)
            hist["c_l2_dists"].append(loc[N.LOSS_CL2] / Bs)
            hist["c_dists"].append(red[N.LOSS_CDIST] / nsteps)               # :489
            for a in range(A):
                hist["loss_recs"][a].append(red[N.LOSS_REC0 + a] / Dm / nsteps)   # :475, :491
            hist["consensus_aug"].append(float(np.mean(consensus_from_counts(counts_aug).cpu().numpy())) if nb else float("nan"))
            # consensus between the arms on the training set (cpl_mixvae.py:563-657), on the device
            cons = self.consensus(train_loader, whole_set=whole_train)
            hist["consensus_train"].append(cons)
            # validation loss (cpl_mixvae.py:665-775): eval mode, no Gumbel noise, hard sample
            if test_loader is not None:
                val_tot, val, val_cons = self.validate(test_loader, full=True)
            else:
                val_tot = val = val_cons = float("nan")
            hist["validation_loss"].append(val_tot)                # :764
            hist["validation_rec_loss"].append(val)                # :763
            hist["consensus_val"].append(val_cons)                 # :762
            dt = time.time() - t0
            hist["epoch_times"].append(dt)
            if rank in (None, 0, dev) or not D.is_dist():
                print(f"epoch {e} | loss: {hist['losses'][-1]:.2f} | rec: {hist['loss_recs'][0][-1]:.2f} | "
                      f"distance: {hist['c_dists'][-1]:.2f} | l2 distance: {hist['c_l2_dists'][-1]:.2f} | "
                      f"aug-cns: {hist['consensus_aug'][-1]:.2f} | train-cns: {cons:.2f} | val: {val:.2f} | "
                      f"time: {dt:.2f}", flush=True)
            if run:
                run.log({"train/total-loss": hist["losses"][-1], "train/joint-loss": hist["loss_joints"][-1],
                         "train/negative-joint-entropy": hist["c_ents"][-1],
                         "train/simplex-distance": hist["c_dists"][-1], "train/l2-distance": hist["c_l2_dists"][-1],
                         "train/time": dt, "train/consensus_aug": hist["consensus_aug"][-1],
                         **{f"train/rec-loss{a}": hist["loss_recs"][a][-1] for a in range(A)}})
                run.log({"train/consensus": cons})
                run.log({"val/total-loss": val_tot, "val/rec-loss": val, "val/consensus": val_cons})
            if self.save and self.folder and (e > 0) and (e % 10 == 0):    # :777-788
                os.makedirs(os.path.join(self.folder, "model"), exist_ok=True)
                self.save_checkpoint(os.path.join(self.folder, "model", f"cpl_mixVAE_model_epoch_{e}.pth"))
            if cons >= good_enuf_consensus or e == E - 1:                   # :851-927 (a NaN consensus never stops)
                if self.folder:
                    os.makedirs(os.path.join(self.folder, "model"), exist_ok=True)
                    self.save_checkpoint(os.path.join(
                        self.folder, "model", f"cns_cpl_mixVAE_model_before_pruning_A{A}_{self.current_time}.pth"))
                hist["stopped_at"] = e
                break
        if self.save and self.folder and n_epoch > 0:                       # :958-972
            os.makedirs(os.path.join(self.folder, "model"), exist_ok=True)
            self.save_checkpoint(os.path.join(self.folder, "model",
                                              f"cpl_mixVAE_model_before_pruning_A{A}_{self.current_time}.pth"))
        # Pruning phase (:996-1444): upstream forces ``stop_prune = True`` at :1007 whatever ``n_epoch_p`` is, so its
        # ``while not stop_prune`` body never runs; what remains of the phase are these two lines of output.
        if rank in (None, 0, dev) or not D.is_dist():
            print("warning: stopping pruning")                              # :1008
            print("Training is done!")                                      # :1446
        hist["bf16_storage"] = bool(self.used_bf16_storage)    # (not in the reference: which copy of x the run's losses were taken against)
        return hist

    @torch.no_grad()
    def consensus(self, loader, whole_set: Optional[bool] = None, chunk: Optional[int] = None) -> float:
        """Mean over arm pairs of ``confmat_mean(confmat_normalize(compute_confmat(labels_a, labels_b, C)))`` with
        ``labels = classify(c)`` of the eval-mode forward (cpl_mixvae.py:563-657).  ``whole_set``: classify every row
        of ``loader.dataset.tensors`` in its base order (what the reference does when the test loader's batch size is
        one, :617) instead of walking the loader (:567; a shuffled drop_last loader would skip its tail rows and spend a
        permutation); None decides from the loader's own batch size, as the validation block does.  Eval mode uses the
        running statistics, so cells are independent and the whole set is processed in chunks of ``chunk`` rows
        (default: the loader's batch size) with identical labels.  Labels, counts and the normalisation stay on the
        device: one host read of A(A-1)/2 doubles per epoch.  Under data parallelism every rank counts its own shard and
        the integer counts are summed (the reference, never run distributed, would report rank-local values)."""
        from ._utils import confmat_counts, consensus_from_counts
        was_training = self.model.training
        self.model.eval()
        counts = confmat_counts(self.n_arm, self.n_categories, self.device)
        seen = 0
        for x in self._eval_batches(loader, whole_set, chunk):
            x = x.to(self.device)
            if x.shape[0] < 1:
                continue
            self.model.eval_labels(x.expand(self.n_arm, -1, -1), self.temp, counts)
            seen += x.shape[0]
        if D.is_dist():
            D.allreduce_sum_(counts)
        self.model.train(was_training)
        if seen == 0:
            return float("nan")
        return float(np.mean(consensus_from_counts(counts).cpu().numpy()))   # np.mean(np.array(consensus)), :654

    @staticmethod
    def _eval_batches(loader, whole_set: Optional[bool] = None, chunk: Optional[int] = None):
        """The reference walks a loader batch by batch when its ``batch_size > 1`` and otherwise takes the whole set
        as ONE batch from ``loader.dataset.tensors`` (cpl_mixvae.py:567-640, :670-760; the default test loader has
        ``batch_size=1``).  Yields x tensors accordingly.  ``chunk``: rows per yielded piece of the whole set (only
        for per-cell work such as the evaluation labels; None = one piece, as the loss needs)."""
        has_set = hasattr(loader, "dataset") and hasattr(loader.dataset, "tensors")
        if whole_set is None:
            whole_set = getattr(loader, "batch_size", None) == 1 and has_set
        if whole_set and has_set:
            if chunk is None and getattr(loader, "batch_size", None) == 1:
                yield loader.dataset.tensors[0]
                return
            step = int(chunk or getattr(loader, "batch_size", None) or 1)
            if hasattr(loader, "row_chunks"):          # device-resident loader: gather piece by piece
                yield from loader.row_chunks(step)
                return
            x = loader.dataset.tensors[0]
            for i in range(0, x.shape[0], step):
                yield x[i:i + step]
            return
        for batch in loader:
            yield batch[0] if isinstance(batch, (tuple, list)) else batch

    @torch.no_grad()
    def validate(self, loader, full: bool = False):
        """The validation block of cpl_mixvae.py:665-775 in eval mode: ``validation_rec_loss = sum over batches and arms
        of loss_rec[a] / D, divided by len(loader) and n_arm`` (:742-763), ``validation_loss = sum of the total loss /
        len(loader)`` (:741, :764) and the between-arm consensus of the labels (:752-762, on the device).  Returns the
        rec loss, or the triple ``(validation_loss, validation_rec_loss, consensus_val)`` with ``full=True``.  With
        the reference's default test loader (batch_size 1) the whole set is one batch and ``len(loader)`` its row
        count, as in the reference."""
        from ._utils import confmat_counts, consensus_from_counts
        was_training = self.model.training
        self.model.eval()
        A = self.n_arm
        counts = confmat_counts(A, self.n_categories, self.device)
        tot = torch.zeros((), dtype=torch.float64, device=self.device)
        rec = torch.zeros((), dtype=torch.float64, device=self.device)
        seen = 0
        for x in self._eval_batches(loader):
            x = x.to(self.device)
            if x.shape[0] < 2:     # a one-cell batch has no batch variance: the reference's loss is NaN there (nn_model.py:75)
                continue
            xs = x.expand(A, -1, -1)
            out = self.model(xs, self.temp, 0.0, eval=True)
            lt = self.model.loss(out[0], [], [], xs, out[7], out[8], out[4], out[6], 0.0)
            tot += lt[0].double()                                             # :741 val_loss += loss.data.item()
            rec += lt[1].double().sum() / self.input_dim                      # :742-743 sum_a loss_rec[a] / D
            labels = torch.stack([N.classify(c) for c in out[4]])            # :744 / :752 classify(cs[a])
            N.confmat_accumulate(labels, self.n_categories, counts)
            seen += 1
        self.model.train(was_training)
        nb = max(len(loader), 1) if hasattr(loader, "__len__") else max(seen, 1)
        val_loss, val_rec = float(tot) / nb, float(rec) / nb / A
        if not full:
            return val_rec
        cons = float(np.mean(consensus_from_counts(counts).cpu().numpy())) if seen and A > 1 else float("nan")
        return val_loss, val_rec, cons

    @torch.no_grad()
    def eval_model(self, dl, c_p=0, c_onehot=0):
        """``cpl_mixVAE.eval_model`` (cpl_mixvae.py:1450-1619): eval-mode forward + loss over every batch ``(x, index)`` of
        ``dl`` and the reference's dictionary -- same keys, shapes and dtypes (float64 numpy arrays, as ``np.zeros``
        gives them): ``state_mu`` / ``state_var`` (s_mean, s_logvar) [A,N,S], ``state_cat`` (argmax c + 1) and
        ``prob_cat`` (max c) [A,N], ``total_loss_rec`` / ``total_likelihood`` [A] (means over batches),
        ``total_dist_z`` / ``total_dist_qz`` (mean simplex / l2 distance), ``mean_test_rec`` (zeros), ``predicted_label``
        [A,N], ``data_indx`` [N], ``z_prob`` (c) and ``z_sample`` (c_smp) [A,N,C], ``x_low`` [A,N,L], ``recon_c``
        (x_rec) [A,N,D], ``prune_indx`` (categories whose fcc bias is zero) and ``cnss`` (between-arm consensus of the
        labels).  Forward, loss, labels, confusion counts and consensus run on the device; the per-batch outputs are
        collected in device buffers and copied to the host once."""
        if self.ref_prior:
            raise NotImplementedError("ref_prior is rejected by the reference loss (nn_model.py:578)")
        A, Cc, Dm, L, S = self.n_arm, self.n_categories, self.input_dim, self.lowD_dim, self.state_dim
        n_rows = len(dl.dataset)
        B = dl.batch_size
        if B is None:
            raise ValueError("error: expected non-None value")            # unwrap(dl.batch_size), cpl_mixvae.py:104-107
        dev = self.device
        was_training = self.model.training
        self.model.eval()
        bias = self.model.fcc[0].bias.detach().cpu().numpy()
        pruning_mask = np.where(bias != 0.0)[0]
        prune_indx = np.where(bias == 0.0)[0]
        f32 = dict(dtype=torch.float32, device=dev)
        x_recs, s_means = torch.zeros(A, n_rows, Dm, **f32), torch.zeros(A, n_rows, S, **f32)
        s_logvars, cs = torch.zeros(A, n_rows, S, **f32), torch.zeros(A, n_rows, Cc, **f32)
        c_smps, x_lows = torch.zeros(A, n_rows, Cc, **f32), torch.zeros(A, n_rows, L, **f32)
        data_indx = torch.zeros(n_rows, dtype=torch.float64, device=dev)
        loss_vecs = []
        from ._utils import confmat_counts, consensus_from_counts
        for i, (x, data_idx) in enumerate(dl):
            n_fst, n_lst = i * B, min((i + 1) * B, n_rows)
            x = x.to(dev)
            xs = x.expand(A, -1, -1)                                      # xs = [x for _ in range(A)], :1519
            out = self.model(xs, self.temp, prior_c=0.0, eval=True, mask=pruning_mask)
            self.model.loss(out[0], out[1], out[2], xs, out[7], out[8], out[4], out[6], 0.0)
            loss_vecs.append(self.model._engine.loss_buf.clone())         # the 9-tuple's scalars, still on the device
            for dst, k in ((s_means, 7), (s_logvars, 8), (cs, 4), (c_smps, 6), (x_lows, 3), (x_recs, 0)):
                dst[:, n_fst:n_lst] = torch.stack(list(out[k]))
            data_indx[n_fst:n_lst] = torch.as_tensor(data_idx).to(dev).to(torch.int64).to(torch.float64)
        labels = N.classify(cs)                                            # argmax c, first maximum (np.argmax), [A,N]
        prob = cs.max(dim=-1).values
        counts = N.confmat_accumulate(labels, Cc, confmat_counts(A, Cc, dev))
        cnss = float(np.mean(consensus_from_counts(counts).cpu().numpy())) if A > 1 and n_rows else float("nan")
        lv = torch.stack(loss_vecs).double().cpu().numpy() if loss_vecs else np.zeros((0, 5 + 3 * A))
        self.model.train(was_training)
        to64 = lambda t: t.double().cpu().numpy()
        lab1 = to64(labels) + 1.0
        return {
            "state_mu": to64(s_means),
            "state_var": to64(s_logvars),
            "state_cat": lab1.copy(),
            "prob_cat": to64(prob),
            "total_loss_rec": lv[:, N.LOSS_REC0:N.LOSS_REC0 + A].mean(axis=0),
            "total_likelihood": lv[:, N.LOSS_REC0 + 2 * A:N.LOSS_REC0 + 3 * A].mean(axis=0),
            "total_dist_z": np.mean(lv[:, N.LOSS_CDIST]),
            "total_dist_qz": np.mean(lv[:, N.LOSS_CL2]),
            "mean_test_rec": np.zeros(A),
            "predicted_label": lab1,
            "data_indx": data_indx.cpu().numpy(),
            "z_prob": to64(cs),
            "z_sample": to64(c_smps),
            "x_low": to64(x_lows),
            "recon_c": to64(x_recs),
            "prune_indx": prune_indx,
            "cnss": cnss,
        }
