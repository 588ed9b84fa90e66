"""Drop-in for the reference's ``mmidas/nn_model.py`` hot path, running on the HIP engine.

Mirrors (names, argument meaning, return contract, error behaviour):

* ``mixVAE_model``            mmidas/nn_model.py:89   (``__init__`` :112-261, ``forward`` :297-368,
                              ``loss`` :495-598)
* ``mk_vae``                  mmidas/nn_model.py:679-721
* ``VAEConfig``               mmidas/nn_model.py:14-36

``state_dict()`` has the reference's 46-keys-per-arm layout (``fc1.{a}.weight`` ...
``batch_s.{a}.num_batches_tracked``); every parameter is a view into one flat fp32 buffer that the
kernels read directly (include/mmvae.h).  All arithmetic happens in libmmvae_hip.so; there is no
PyTorch fallback -- without a GPU (or without the built library) ``forward`` raises.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional

import torch
from torch import nn
from torch.nn import ModuleList as mdl

from . import _native as N

# construction order of the reference (nn_model.py:184-203): decides RNG consumption at init
_LINEAR_ORDER = ["fc1", "fc2", "fc3", "fc4", "fc5", "fcc", "fc_mu", "fc_sigma", "fc6", "fc7", "fc8", "fc9", "fc10",
                 "fc11"]
_BN_ORDER = ["batch_l1", "batch_l2", "batch_l3", "batch_l4", "batch_l5", "batch_s"]


@dataclass
class VAEConfig:
    n_categories: int = 92
    state_dim: int = 2
    input_dim: int = 5032
    fc_dim: int = 100
    lowD_dim: int = 10
    x_drop: float = 0.5
    s_drop: float = 0.2
    lr: float = 0.001
    lam: float = 1
    lam_pc: float = 1
    n_arm: int = 2
    temp: float = 1.0
    tau: float = 0.005
    beta: float = 1.0
    hard: bool = False
    variational: bool = True
    ref_prior: bool = False
    trained_model: Optional[str] = None
    n_pr: int = 0
    momentum: float = 0.01
    mode: str = "MSE"


class _LossFn(torch.autograd.Function):
    """``loss.backward()`` (cpl_mixvae.py:462): routes into mmvae_backward and hands each parameter a
    view of the flat gradient buffer."""

    @staticmethod
    def forward(ctx, model, loss_value, *params):
        ctx.model = model
        ctx.step_id = model._step_id
        return loss_value.clone()

    @staticmethod
    def backward(ctx, grad_out):
        model = ctx.model
        if ctx.step_id != model._step_id or model._ctx is None:
            raise RuntimeError("backward() must follow the forward()/loss() pair that produced this loss")
        c = model._ctx
        model._engine.backward(c["hyper"], c["noise"], model._flat, c["x"], c["x_arm_stride"], model._flat_grad,
                               grad_scale=float(grad_out))
        return (None, None) + tuple(model._grad_views)


class mixVAE_model(nn.Module):
    """Multi-arm coupled mixture VAE (cpl-mixVAE); same constructor as nn_model.py:112-134."""

    def __init__(self, input_dim, fc_dim, n_categories, state_dim, lowD_dim, x_drop, s_drop, n_arm, lam, lam_pc,
                 tau, beta, hard, variational, device, eps, momentum, ref_prior, loss_mode, norm="batch"):
        super().__init__()
        self.input_dim = input_dim
        self.fc_dim = fc_dim
        self.lowD_dim = lowD_dim
        self.state_dim = state_dim
        self.n_categories = n_categories
        self.x_dp = nn.Dropout(x_drop)
        self.s_dp = nn.Dropout(s_drop)
        self.hard = hard
        self.n_arm = n_arm
        self.lam = lam
        self.lam_pc = lam_pc
        self.tau = tau
        self.beta = beta
        self.varitional = variational   # (sic) attribute name of the reference, nn_model.py:175
        self.eps = eps
        self.ref_prior = ref_prior
        self.momentum = momentum
        self.device = device
        self.loss_mode = loss_mode

        D, H, L, Cc, S = input_dim, fc_dim, lowD_dim, n_categories, state_dim
        shapes = {"fc1": (D, H), "fc2": (H, H), "fc3": (H, H), "fc4": (H, H), "fc5": (H, L), "fcc": (L, Cc),
                  "fc_mu": (L + Cc, S), "fc_sigma": (L + Cc, S), "fc6": (S + Cc, L), "fc7": (L, H), "fc8": (H, H),
                  "fc9": (H, H), "fc10": (H, H), "fc11": (H, D)}
        for name in _LINEAR_ORDER:
            i, o = shapes[name]
            setattr(self, name, mdl([nn.Linear(i, o) for _ in range(n_arm)]))
        if loss_mode == "ZINB":   # nn_model.py:204-206 builds them; forward rejects ZINB (:315)
            self.fc11_p = mdl([nn.Linear(H, D) for _ in range(n_arm)])
            self.fc11_r = mdl([nn.Linear(H, D) for _ in range(n_arm)])
        bn_dims = {"batch_l1": H, "batch_l2": H, "batch_l3": H, "batch_l4": H, "batch_l5": L, "batch_s": S}
        for name in _BN_ORDER:
            setattr(self, name, mdl([nn.BatchNorm1d(num_features=bn_dims[name], eps=eps, momentum=momentum,
                                                    affine=False) for _ in range(n_arm)]))
        # engine state (created lazily on the parameters' device)
        self._engine: Optional[N.Engine] = None
        self._engines = {}
        self._flat = self._flat_grad = self._bn_flat = self._nbt = None
        self._grad_views: List[torch.Tensor] = []
        self._layout = None
        self._ctx = None
        self._step_id = 0
        self._explicit_noise = None
        self._noise_seed = None
        self._noise_offset = 0
        self._exec: Optional[N.Exec] = None   # None: split factors / experiment switches from the environment
        # operand type of the five D x H GEMMs: "fp32" (the parity configuration) or "bf16" (BASELINE.json's bf16
        # configuration: operands rounded to bf16, fp32 accumulation; everything else and all parameters stay fp32)
        self.gemm_dtype = "fp32"

    # ------------------------------------------------------------------ flat parameter storage
    def _dims(self, B: int) -> N.Dims:
        return N.Dims(self.n_arm, B, self.input_dim, self.fc_dim, self.lowD_dim, self.n_categories, self.state_dim)

    def _param_of(self, t: int, a: int) -> nn.Parameter:
        name, kind = N.PARAM_NAMES[t].split(".")
        return getattr(getattr(self, name)[a], kind)

    def _is_packed(self) -> bool:
        if self._flat is None:
            return False
        p = self.fc1[0].weight
        lay = self._layout
        return (p.device == self._flat.device and p.data_ptr() == self._flat.data_ptr() + 4 * int(lay.offset[0])
                and self.fc11[self.n_arm - 1].bias.data_ptr()
                == self._flat.data_ptr() + 4 * (int(lay.per_arm) * (self.n_arm - 1) + int(lay.offset[27]))
                and self.batch_l1[0].running_mean.data_ptr() == self._bn_flat.data_ptr())

    def _pack(self):
        """(Re)build the flat buffers on the parameters' current device and re-point every parameter /
        BatchNorm buffer at a view of them.  Values are preserved."""
        dev = self.fc1[0].weight.device
        lay = N.param_layout(self._dims(2))
        A = self.n_arm
        flat = torch.zeros(A * int(lay.per_arm), dtype=torch.float32, device=dev)
        grad = torch.zeros_like(flat)
        bn = torch.zeros(A * int(lay.bn_per_arm), dtype=torch.float32, device=dev)
        nbt = torch.zeros(A * N.N_BN, dtype=torch.int64, device=dev)
        views = []
        with torch.no_grad():
            for a in range(A):
                for t in range(N.N_PARAM_TENSORS):
                    p = self._param_of(t, a)
                    o = a * int(lay.per_arm) + int(lay.offset[t])
                    v = flat[o: o + p.numel()].view(p.shape)
                    v.copy_(p.data.to(torch.float32))
                    p.data = v
                    gv = grad[o: o + p.numel()].view(p.shape)
                    if p.grad is not None:
                        gv.copy_(p.grad)
                        p.grad = gv
                    views.append((a, t, gv))
                for i, name in enumerate(_BN_ORDER):
                    m = getattr(self, name)[a]
                    n = int(lay.bn_dim[i])
                    om = a * int(lay.bn_per_arm) + int(lay.bn_mean_offset[i])
                    ov = a * int(lay.bn_per_arm) + int(lay.bn_var_offset[i])
                    bn[om: om + n].copy_(m.running_mean)
                    bn[ov: ov + n].copy_(m.running_var)
                    m.running_mean = bn[om: om + n]
                    m.running_var = bn[ov: ov + n]
                    nbt[a * N.N_BN + i] = m.num_batches_tracked
                    m.num_batches_tracked = nbt[a * N.N_BN + i]
        # gradient views in model.parameters() order (layer-major, then arm; nn_model.py:184-203)
        order = {id(p): k for k, p in enumerate(self.parameters())}
        gv_sorted = [None] * len(order)
        for a, t, gv in views:
            gv_sorted[order[id(self._param_of(t, a))]] = gv
        self._grad_views = gv_sorted
        self._flat, self._flat_grad, self._bn_flat, self._nbt, self._layout = flat, grad, bn, nbt, lay
        self._engine = None
        self._engines = {}

    def _ensure(self, B: int) -> N.Engine:
        if not self._is_packed():
            self._pack()
        mode = N.gemm_mode(self.gemm_dtype) & 0xFF
        if (self._engine is None or self._engine.dims.B != B or self._engine.device != self._flat.device
                or self._engine.gemm_engine != mode):
            # a trainer alternates between a few batch sizes (training batch, evaluation chunks, their ragged tails):
            # keep the last few engines (workspace + events each) instead of reallocating 0.5 GB per switch.  The GEMM
            # engine is part of the key: the workspace's split factors are chosen for its workgroup shapes.
            key = (B, str(self._flat.device), id(self._exec), mode)
            eng = self._engines.pop(key, None)
            if eng is None:
                d = self._dims(B)
                eng = N.Engine(d.A, d.B, d.D, d.H, d.L, d.C, d.S, self._flat.device, self._exec, gemm_engine=mode)
            self._engines[key] = eng
            while len(self._engines) > 3:
                self._engines.pop(next(iter(self._engines)))
            self._engine = eng
        return self._engine

    def flat_parameters(self) -> torch.Tensor:
        """The flat fp32 parameter buffer (arm-major, layout of mmvae_param_layout)."""
        if not self._is_packed():
            self._pack()
        return self._flat

    def flat_grad(self) -> torch.Tensor:
        if not self._is_packed():
            self._pack()
        return self._flat_grad

    def bind_grads(self):
        """Point every parameter's ``.grad`` at its view of the flat gradient buffer (what ``loss.backward()`` does on
        the three-call path), so that a stock ``torch.optim`` optimizer can follow a fused step run with do_adam=False."""
        if not self._is_packed():
            self._pack()
        for p, gv in zip(self.parameters(), self._grad_views):
            p.grad = gv

    # ------------------------------------------------------------------ noise control
    def set_explicit_noise(self, noise):
        """Parity hook: x_mask uint8 [A,B,D], u_gumbel [A,B,C], u_state [A,B,S], s_mask uint8 [A,B,S]
        (device tensors) consumed by the next forward passes; None returns to in-kernel Philox.  A LIST of such
        dicts is a schedule: every forward / fused step takes the next entry (running out raises), which is how a
        recorded run of the reference trainer is replayed step by step."""
        self._explicit_noise = list(noise) if isinstance(noise, (list, tuple)) else noise

    def _hyper(self, temp: float, eval_flag: bool) -> N.Hyper:
        return N.Hyper(self.tau, float(temp), self.beta, self.lam, self.eps, self.momentum, float(self.x_dp.p),
                       float(self.s_dp.p), int(bool(self.hard)), int(self.training), int(bool(eval_flag)),
                       N.gemm_mode(self.gemm_dtype))

    def _next_noise(self) -> N.Noise:
        if isinstance(self._explicit_noise, list):
            if not self._explicit_noise:
                raise RuntimeError("explicit noise schedule exhausted")
            self._noise_keep = self._explicit_noise.pop(0)      # keeps the tensors alive while the kernels read them
            return N.make_noise(self._noise_keep)
        if self._explicit_noise is not None:
            return N.make_noise(self._explicit_noise)
        if self._noise_seed is None:
            self._noise_seed = int(torch.initial_seed()) & 0xFFFFFFFFFFFFFFFF
        self._noise_offset += 1
        return N.make_noise(None, self._noise_seed, self._noise_offset)

    # ------------------------------------------------------------------ reference API
    def _prep_x(self, x):
        """Accepts the reference's inputs: a list of A [B,D] tensors or an [A,B,D] tensor (typically
        ``x.expand(A,-1,-1)``, cpl_mixvae.py:425).  Returns (tensor, arm stride in floats)."""
        if isinstance(x, torch.Tensor):
            assert x.dim() == 3, "x must be [n_arm, batch, input_dim]"
            if x.stride(0) == 0:
                return x[0].contiguous().float(), 0
            xc = x.contiguous().float()
            return xc, xc.shape[1] * xc.shape[2]
        xs = list(x)
        if all(t.data_ptr() == xs[0].data_ptr() and t.shape == xs[0].shape for t in xs):
            return xs[0].contiguous().float(), 0
        xc = torch.stack([t.float() for t in xs]).contiguous()
        return xc, xc.shape[1] * xc.shape[2]

    def forward(self, x, temp, prior_c=[], eval=False, mask=None):
        """Same contract as nn_model.py:297-368: returns
        ``x_recs, [], [], x_lows, cs, s_smps, c_smps, s_means, s_logvars, c_probs`` (lists over arms)."""
        assert not self.loss_mode == "ZINB", "ZINB not implemented"
        assert self.varitional, "Non-variational not implemented"
        assert len(x) == self.n_arm
        mask_words = None
        if mask is not None:
            # nn_model.py:332-335: c = softmax(c_prob[:, mask] / tau) on the kept categories, 0 elsewhere.  eval_model passes the
            # indices of the categories whose fcc bias is non-zero (cpl_mixvae.py:1476-1478, :1524): every category for an
            # unpruned model, a subset for a checkpoint of a pruned one.  (A boolean mask selects like an index list.)
            import numpy as _np
            mk = _np.asarray(mask.detach().cpu() if isinstance(mask, torch.Tensor) else mask)
            mk = _np.flatnonzero(mk) if mk.dtype == _np.bool_ else _np.unique(mk.astype(_np.int64))
            if mk.size == 0 or mk[0] < 0 or mk[-1] >= self.n_categories:
                raise IndexError(f"category mask {mk.tolist()} outside [0, {self.n_categories})")
            if mk.size < self.n_categories:
                mask_words = [0, 0, 0, 0]
                for k in mk.tolist():
                    mask_words[k >> 5] |= 1 << (k & 31)
        if self.ref_prior:
            raise NotImplementedError("ref_prior is rejected by the reference loss (nn_model.py:578)")
        xt, xs = self._prep_x(x)
        if xt.device.type != "cuda":
            raise N.NativeError("mixVAE_model.forward needs GPU tensors: the model runs only on the HIP engine")
        A, B, D = self.n_arm, xt.shape[-2], xt.shape[-1]
        assert D == self.input_dim
        eng = self._ensure(B)
        hyper = self._hyper(temp, eval)
        if mask_words is not None:
            for i in range(4):
                hyper.cat_mask[i] = mask_words[i]
        noise = self._next_noise()
        need_grad = bool(self.training and torch.is_grad_enabled())
        x_rec = torch.empty(A, B, D, dtype=torch.float32, device=xt.device)
        eng.forward(hyper, noise, self._flat, self._bn_flat, self._nbt if self.training else None, xt, xs, x_rec,
                    need_grad)
        self._step_id += 1
        self._ctx = {"hyper": hyper, "noise": noise, "x": xt, "x_arm_stride": xs, "need_grad": need_grad,
                     "keep": self._explicit_noise}
        L, Cc, S = self.lowD_dim, self.n_categories, self.state_dim
        grab = lambda name, w: list(eng.ws_view(name, w).clone().unbind(0))
        out = (list(x_rec.unbind(0)), [], [], grab("x_low", L), grab("c", Cc), grab("s_smp", S), grab("c_smp", Cc),
               grab("s_mean", S), grab("s_logvar", S), grab("c_prob", Cc))
        self._ctx["outs"] = out           # loss() checks that it is handed exactly these, unmodified
        self._ctx["versions"] = {i: [t._version for t in out[i]] for i in (0, 4, 6, 7, 8)}
        self._ctx["x_in"] = x if isinstance(x, torch.Tensor) else list(x)
        return out

    @staticmethod
    def _same(given, kept, versions) -> bool:
        """Is ``given`` (a list of per-arm tensors, or one stacked tensor) the list ``forward`` returned, unmodified?"""
        if isinstance(given, torch.Tensor):
            given = list(given.unbind(0)) if given.dim() == kept[0].dim() + 1 else [given]
        try:
            given = list(given)
        except TypeError:
            return False
        return len(given) == len(kept) and all(
            isinstance(g, torch.Tensor) and g.data_ptr() == k.data_ptr() and g.shape == k.shape and g._version == v
            for g, k, v in zip(given, kept, versions))

    def loss(self, recon_x, p_x, r_x, x, mu, log_sigma, qc, c, prior_c=[]):
        """Same contract as nn_model.py:495-598; returns the reference's 9-tuple.

        The loss is evaluated by the kernels on what the preceding ``forward`` left in the workspace, so ``recon_x``,
        ``mu``, ``log_sigma``, ``qc`` and ``c`` MUST be the tensors that ``forward`` returned (outputs 0, 7, 8, 4, 6,
        unmodified) and ``x`` the input it was given: anything else raises instead of silently returning the loss of
        other tensors (the reference, mmidas/model.py:108-113, computes from whatever it is handed)."""
        assert len(recon_x) == len(c) == self.n_arm
        assert not self.ref_prior
        if self._ctx is None:
            raise RuntimeError("loss() must follow forward()")
        A = self.n_arm
        kept = self._ctx["outs"]
        for name, given, idx in (("recon_x", recon_x, 0), ("mu", mu, 7), ("log_sigma", log_sigma, 8), ("qc", qc, 4),
                                 ("c", c, 6)):
            if not self._same(given, kept[idx], self._ctx["versions"][idx]):
                raise ValueError(f"loss(): `{name}` is not the unmodified output {idx} of the preceding forward(); the HIP "
                                 "engine evaluates the loss on the tensors its forward produced")
        x_in = self._ctx["x_in"]
        same_obj = x is x_in or (not isinstance(x, torch.Tensor) and not isinstance(x_in, torch.Tensor) and x is not None
                                 and len(x) == len(x_in) and all(p is q for p, q in zip(x, x_in)))
        if x is not None and not same_obj:   # (None: callers that only want the scalars of the last forward)
            xt, _ = self._prep_x(x)
            x0 = self._ctx["x"]
            if not (xt.data_ptr() == x0.data_ptr() or (xt.shape == x0.shape and torch.equal(xt, x0))):
                raise ValueError("loss(): `x` is not the input of the preceding forward()")
        eng = self._engine
        buf = eng.loss(self._ctx["hyper"])
        if A == 1:
            # the reference divides by len([]) here (nn_model.py:592-594)
            raise ZeroDivisionError("division by zero")
        vals = buf.clone()
        total = vals[N.LOSS_TOTAL]
        if self._ctx["need_grad"]:
            total = _LossFn.apply(self, total, *self.parameters())
        rec = vals[N.LOSS_REC0: N.LOSS_REC0 + A]
        kl = vals[N.LOSS_REC0 + A: N.LOSS_REC0 + 2 * A]
        ll = vals[N.LOSS_REC0 + 2 * A: N.LOSS_REC0 + 3 * A]
        return (total, rec.clone(), vals[N.LOSS_JOINT], vals[N.LOSS_CENT], vals[N.LOSS_CDIST], vals[N.LOSS_CL2],
                list(kl.unbind(0)), [], list(ll.unbind(0)))

    # ------------------------------------------------------------------ evaluation labels (consensus path)
    @torch.no_grad()
    def eval_labels(self, x, temp=1.0, counts=None) -> torch.Tensor:
        """``classify(cs[a])`` of ``self(x, temp, eval=True)`` for every arm without leaving the device
        (cpl_mixvae.py:596-611): int32 [n_arm, batch].  Runs the encoder and the latent block only (BatchNorm running
        statistics, no Gumbel noise); ``counts`` (int64 [pairs, C, C], see ``_utils.confmat_counts``) also receives
        this batch's between-arm confusion counts.  The module must be in eval mode, as in the reference's loop."""
        if self.training:
            raise RuntimeError("eval_labels() needs model.eval(): the reference classifies in eval mode "
                               "(cpl_mixvae.py:563)")
        xt, xs = self._prep_x(x)
        if xt.device.type != "cuda":
            raise N.NativeError("mixVAE_model.eval_labels needs GPU tensors: the model runs only on the HIP engine")
        eng = self._ensure(xt.shape[-2])
        labels = torch.empty(self.n_arm, xt.shape[-2], dtype=torch.int32, device=xt.device)
        eng.eval_classify(self._hyper(temp, True), self._flat, self._bn_flat, xt, xs, labels, counts)
        self._ctx = None
        return labels

    # ------------------------------------------------------------------ fused step (trainer path)
    def fused_train_step(self, x, temp, opt=None, do_adam=True):
        """forward + loss + backward (+ Adam) in one C-ABI call: cpl_mixvae.py:434-463.
        Returns the device loss vector (see include/mmvae.h MMVAE_LOSS_*); no host sync."""
        xt, xs = self._prep_x(x)
        eng = self._ensure(xt.shape[-2])
        hyper = self._hyper(temp, False)
        noise = self._next_noise()
        self._step_id += 1
        self._ctx = None
        if do_adam:
            opt._bind(self)
            opt.step_count += 1
            g = opt.param_groups[0]
            return eng.train_step(hyper, noise, self._flat, self._bn_flat, self._nbt, xt, xs, self._flat_grad, True,
                                  opt.exp_avg, opt.exp_avg_sq, opt.step_count, g["lr"], g["betas"][0], g["betas"][1],
                                  g["eps"], g["weight_decay"], opt.decoupled)
        return eng.train_step(hyper, noise, self._flat, self._bn_flat, self._nbt, xt, xs, self._flat_grad, False,
                              None, None, 1, 0.0)


    def fused_train_step_rows(self, data, rows, temp, opt=None, do_adam=True, data16=None):
        """``fused_train_step`` on the batch ``data[rows]`` without materialising it (``x.expand`` over the arms): the step
        reads the cells x genes matrix through a row map (mmvae_train_step_rows; bit-identical to gather + step).  Raises
        ``NotImplementedError`` where the library does not offer it -- engines other than fp32x3, no input dropout, a
        matrix beyond 4 GB --: gather the batch and call ``fused_train_step`` then.  ``data16`` (``gemm_dtype == "bf16"``
        only): the matrix's bf16 copy (``_native.to_bf16``, ``DeviceLoader.data_bf16()``): bf16 storage, see DESIGN.md section 13."""
        if data.device.type != "cuda" or data.dtype != torch.float32 or data.shape[1] != self.input_dim:
            raise N.NativeError("fused_train_step_rows needs the float32 cells x genes matrix on the GPU")
        rows = rows.to(device=data.device, dtype=torch.int64).contiguous()
        eng = self._ensure(int(rows.numel()))
        hyper = self._hyper(temp, False)
        noise = self._next_noise()
        try:
            if do_adam:
                opt._bind(self)
                g = opt.param_groups[0]
                buf = eng.train_step_rows(hyper, noise, self._flat, self._bn_flat, self._nbt, data, rows, self._flat_grad, True,
                                          opt.exp_avg, opt.exp_avg_sq, opt.step_count + 1, g["lr"], g["betas"][0],
                                          g["betas"][1], g["eps"], g["weight_decay"], opt.decoupled, data16=data16)
                opt.step_count += 1
            else:
                buf = eng.train_step_rows(hyper, noise, self._flat, self._bn_flat, self._nbt, data, rows, self._flat_grad, False,
                                          None, None, 1, 0.0, data16=data16)
        except NotImplementedError:
            if self._explicit_noise is None:
                self._noise_offset -= 1            # the refused call consumed nothing
            elif isinstance(self._explicit_noise, list):
                self._explicit_noise.insert(0, self._noise_keep)
            raise
        self._step_id += 1
        self._ctx = None
        return buf


def mk_vae(C, state_dim, input_dim, device, eps=1e-8, fc_dim=100, latent_dim=10, x_drop=0.5, s_drop=0.2, lr=0.001,
           lam=1, lam_pc=1, A=2, tau=0.005, beta=1.0, hard=False, variational=True, ref_prior=False, momentum=0.01,
           mode="MSE") -> nn.Module:
    """nn_model.py:679-721."""
    return mixVAE_model(input_dim=input_dim, fc_dim=fc_dim, n_categories=C, state_dim=state_dim, lowD_dim=latent_dim,
                        x_drop=x_drop, s_drop=s_drop, n_arm=A, lam=lam, lam_pc=lam_pc, tau=tau, beta=beta, hard=hard,
                        variational=variational, device=device, eps=eps, ref_prior=ref_prior, momentum=momentum,
                        loss_mode=mode).to(device)
