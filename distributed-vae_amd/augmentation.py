"""Drop-in for the augmenter the reference trainer runs in front of every train step.

``Augmenter_smartseq`` mirrors ``mmidas/augmentation/udagan.py:217-329`` (constructor arguments, sub-module names and
therefore ``state_dict`` keys, ``forward(x, batched, scale)`` returning ``(s, x_aug)``); ``mk_augmenter`` mirrors
``mmidas/cpl_mixvae.py:128-149``.  The forward runs in the HIP library (``csrc/augment.hip``) and exists for eval mode
only -- the one the trainer uses (``self.netA = netA.to(self.device).eval()``, cpl_mixvae.py:184): Dropout is the
identity and every BatchNorm1d normalises with its running statistics.  There is no CPU / PyTorch fallback.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch
from torch import nn

from . import _native as N


class Augmenter_smartseq(nn.Module):
    def __init__(self, noise_dim, latent_dim, input_dim=5000, n_dim=500, p_drop=0.5):
        super().__init__()
        moment = 0.01
        self.noise_dim = noise_dim
        self.dp = nn.Dropout(p_drop)
        self.noise = nn.Linear(noise_dim, noise_dim, bias=False)
        self.bnz = nn.BatchNorm1d(noise_dim)
        n1, n5 = input_dim // 5, n_dim // 5
        bn = lambda n: nn.BatchNorm1d(num_features=n, eps=1e-10, momentum=moment, affine=False)  # noqa: E731
        # (name, in, out) in the reference's construction order (the order fixes the RNG stream of the initialisation)
        enc = [("fc1", input_dim, n1), ("fc2", n1, n1), ("fc3", n1, n_dim), ("fc4", n_dim, n_dim),
               ("fc5", n_dim + noise_dim, n5)]
        dec = [("fc6", latent_dim, n5), ("fc7", n5, n_dim), ("fc8", n_dim, n_dim), ("fc9", n_dim, n1), ("fc10", n1, n1)]
        for name, i, o in enc:
            setattr(self, name, nn.Linear(i, o))
            setattr(self, "batch_" + name, bn(o))
        self.fc_mu = nn.Linear(n5, latent_dim)
        self.fc_sigma = nn.Linear(n5, latent_dim)
        self.batch_fc_mu = bn(latent_dim)
        for name, i, o in dec:
            setattr(self, name, nn.Linear(i, o))
            setattr(self, "batch_" + name, bn(o))
        self.fc11 = nn.Linear(n1, input_dim)
        self._dims = (input_dim, n1, n_dim, n5, latent_dim, noise_dim)
        self.gemm_dtype = "fp32"          # "bf16": bf16 operands in the ten large Linear layers (fp32 accumulation)
        self._packed: Optional[torch.Tensor] = None
        self._ws: Optional[torch.Tensor] = None
        self._explicit = None

    # ------------------------------------------------------------------ packed weights
    def _apply(self, fn, *a, **k):
        self._packed = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._packed = None
        return super().load_state_dict(*a, **k)

    def repack(self):
        """Call after modifying parameters or running statistics in place (the packed copy is otherwise kept)."""
        self._packed = None

    def _exec(self) -> N.Exec:
        if getattr(self, "_ex", None) is None:
            self._ex = N.exec_from_env()      # MMVAE_AUG_TILE (A/B timing); the library itself reads no environment
        return self._ex

    def _aug_dims(self, A, B) -> N.AugDims:
        D, n1, n3, n5, z, nz = self._dims
        return N.AugDims(A, B, D, n1, n3, n5, z, nz)

    def _pack(self, dims: N.AugDims):
        dev = self.fc1.weight.device
        t = N.AugTensors()
        keep = []

        def ptr(x):
            x = x.detach().contiguous().float()
            keep.append(x)
            return x.data_ptr()
        for i in range(11):
            lin = getattr(self, f"fc{i + 1}")
            t.w[i], t.b[i] = ptr(lin.weight), ptr(lin.bias)
        for i in range(10):
            b = getattr(self, f"batch_fc{i + 1}")
            t.bn_mean[i], t.bn_var[i] = ptr(b.running_mean), ptr(b.running_var)
        t.w_mu, t.b_mu = ptr(self.fc_mu.weight), ptr(self.fc_mu.bias)
        t.w_sigma, t.b_sigma = ptr(self.fc_sigma.weight), ptr(self.fc_sigma.bias)
        t.bn_mu_mean, t.bn_mu_var = ptr(self.batch_fc_mu.running_mean), ptr(self.batch_fc_mu.running_var)
        t.noise_w = ptr(self.noise.weight)
        t.bnz_weight, t.bnz_bias = ptr(self.bnz.weight), ptr(self.bnz.bias)
        t.bnz_mean, t.bnz_var = ptr(self.bnz.running_mean), ptr(self.bnz.running_var)
        n = int(N.lib().mmvae_aug_packed_floats(C.byref(dims)))
        if n == 0:
            N.check(-2, "mmvae_aug_packed_floats")
        packed = torch.empty(n, dtype=torch.float32, device=dev)
        N.check(N.lib().mmvae_aug_pack(C.byref(dims), C.byref(t), N._ptr(packed), N._stream(dev)), "mmvae_aug_pack")
        torch.cuda.current_stream(dev).synchronize()   # `keep` may be freed once the pack kernels have run
        self._packed = packed

    # ------------------------------------------------------------------ noise control (parity hook)
    def set_explicit_noise(self, z0: Optional[torch.Tensor], eps: Optional[torch.Tensor]):
        """z0 [A,B,noise_dim] and eps [A,B,latent_dim] standard-normal draws used by the next forward passes instead of
        torch.randn; (None, None) returns to torch.randn on the device."""
        self._explicit = None if z0 is None else (z0, eps)

    # ------------------------------------------------------------------ reference API
    @torch.no_grad()
    def forward(self, x, batched, scale=1.0, out=None):
        """udagan.py:281-329 in eval mode.  batched: x is [A, B, D] (typically ``x.expand(A, -1, -1)``), returns
        ``(s [A,B,latent], x_aug [A,B,D])``; otherwise x is [B, D] and the leading axis is dropped.  ``out``: optional
        preallocated ``(s, x_aug)`` float32 device tensors of those shapes (the trainer's pipeline reuses a ring of them)."""
        if self.training:
            raise NotImplementedError("the HIP augmenter implements eval mode only: the trainer runs netA.eval() "
                                      "(cpl_mixvae.py:184)")
        if x.device.type != "cuda":
            raise N.NativeError("Augmenter_smartseq.forward needs GPU tensors: it runs only on the HIP engine")
        D, n1, n3, n5, Z, NZ = self._dims
        if batched:
            assert x.dim() == 3 and x.shape[-1] == D
            A, B = x.shape[0], x.shape[1]
            if x.stride(0) == 0:
                xt, xs = x[0].contiguous().float(), 0
            else:
                xt, xs = x.contiguous().float(), B * D
        else:
            assert x.dim() == 2 and x.shape[-1] == D
            A, B = 1, x.shape[0]
            xt, xs = x.contiguous().float(), 0
        dims = self._aug_dims(A, B)
        if self._packed is None or self._packed.device != xt.device:
            self._pack(dims)
        if self._explicit is not None:
            z0, eps = (t.to(xt.device).float().contiguous() for t in self._explicit)
            assert z0.shape == (A, B, NZ) and eps.shape == (A, B, Z)
        else:
            z0 = torch.randn(A, B, NZ, device=xt.device)          # udagan.py:283-289
            eps = torch.randn(A, B, Z, device=xt.device)          # reparam_trick, aug_utils.py:64
        need = int(N.lib().mmvae_aug_workspace_bytes(C.byref(dims), int(xs == 0)))
        if self._ws is None or self._ws.numel() * 4 < need or self._ws.device != xt.device:
            self._ws = torch.empty(need // 4, dtype=torch.float32, device=xt.device)
        if out is not None:
            s, out = out
            assert s.shape == (A, B, Z) and out.shape == (A, B, D) and s.is_contiguous() and out.is_contiguous()
            assert s.dtype == out.dtype == torch.float32 and out.device == xt.device
        else:
            s = torch.empty(A, B, Z, dtype=torch.float32, device=xt.device)
            out = torch.empty(A, B, D, dtype=torch.float32, device=xt.device)
        N.check(N.lib().mmvae_augment(C.byref(dims), N._ptr(self._packed), N._ptr(xt), xs, N._ptr(z0), N._ptr(eps),
                                      float(scale), N._ptr(self._ws), self._ws.numel() * 4, N._ptr(s), N._ptr(out),
                                      N.gemm_mode(self.gemm_dtype), C.byref(self._exec()), N._stream(xt.device)),
                "mmvae_augment")
        return (s, out) if batched else (s[0], out[0])


    # ------------------------------------------------------------------ row-indexed forward (no reference counterpart)
    def planes_needed(self) -> int:
        """Slice planes per matrix element the row-indexed forward wants from ``DeviceLoader.data_planes`` for the current
        ``gemm_dtype``: 3 (fp32: the exact fp32x3 slices), 1 (bf16), 0 = not offered (the fp32 matrix-instruction engine)."""
        mode = N.gemm_mode(self.gemm_dtype) & 0xFF
        return 3 if mode == 2 else (1 if mode == 1 else 0)

    @torch.no_grad()
    def forward_rows(self, planes, n_rows, rows, n_arm, scale=1.0, out=None):
        """``forward(data[rows].expand(n_arm, -1, -1), True, scale)`` for a batch that is rows of a resident matrix held as
        the GEMM engine's slice planes (``_native.tp_planes(data, self.planes_needed())``): the first layer reads the rows in
        place -- no gathered batch, no per-batch conversion; same results bit for bit.  rows: int64 [B] on the device."""
        if self.training:
            raise NotImplementedError("the HIP augmenter implements eval mode only")
        D, n1, n3, n5, Z, NZ = self._dims
        A, B = int(n_arm), int(rows.shape[0])
        dev = rows.device
        dims = self._aug_dims(A, B)
        if self._packed is None or self._packed.device != dev:
            self._pack(dims)
        if self._explicit is not None:
            z0, eps = (t.to(dev).float().contiguous() for t in self._explicit)
            assert z0.shape == (A, B, NZ) and eps.shape == (A, B, Z)
        else:
            z0 = torch.randn(A, B, NZ, device=dev)          # udagan.py:283-289
            eps = torch.randn(A, B, Z, device=dev)          # reparam_trick, aug_utils.py:64
        need = int(N.lib().mmvae_aug_workspace_bytes(C.byref(dims), 1))
        if self._ws is None or self._ws.numel() * 4 < need or self._ws.device != dev:
            self._ws = torch.empty(need // 4, dtype=torch.float32, device=dev)
        if out is not None:
            s, out = out
            assert s.shape == (A, B, Z) and out.shape == (A, B, D) and s.is_contiguous() and out.is_contiguous()
        else:
            s = torch.empty(A, B, Z, dtype=torch.float32, device=dev)
            out = torch.empty(A, B, D, dtype=torch.float32, device=dev)
        rows = rows.to(torch.int64).contiguous()
        N.check(N.lib().mmvae_augment_rows(C.byref(dims), N._ptr(self._packed), N._ptr(planes), int(n_rows), self.planes_needed(),
                                           N._ptr(rows), N._ptr(z0), N._ptr(eps), float(scale), N._ptr(self._ws),
                                           self._ws.numel() * 4, N._ptr(s), N._ptr(out), N.gemm_mode(self.gemm_dtype),
                                           C.byref(self._exec()), N._stream(dev)), "mmvae_augment_rows")
        return s, out


def mk_augmenter(pretrained: str, load: bool = True):
    """cpl_mixvae.py:128-149: the checkpoint holds ``parameters`` (num_n, num_z, n_features) and ``netA``.  Loaded with
    ``weights_only=True`` (tensors and plain containers only)."""
    aug_model = torch.load(pretrained, map_location="cpu", weights_only=True)
    aug_param = aug_model["parameters"]
    netA = Augmenter_smartseq(noise_dim=aug_param["num_n"], latent_dim=aug_param["num_z"],
                              input_dim=aug_param["n_features"])
    if load:
        netA.load_state_dict(aug_model["netA"])
    return aug_model, aug_param, netA
