"""ctypes binding of libmmvae_hip.so (C ABI: include/mmvae.h).

This is the only door between the Python host code and the HIP kernels.  There is no CPU or
PyTorch fallback: if the shared library is missing or a call fails, this module raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MMVAE_LIB") or os.path.join(HERE, "libmmvae_hip.so")   # env override: A/B timing of builds

ABI_VERSION = 4
N_PARAM_TENSORS = 28
N_BN = 6
MAX_ARMS = 8

# tensor order of mmvae_param_layout_t (include/mmvae.h)
PARAM_NAMES = [
    "fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias", "fc3.weight", "fc3.bias", "fc4.weight", "fc4.bias",
    "fc5.weight", "fc5.bias", "fcc.weight", "fcc.bias", "fc_mu.weight", "fc_sigma.weight", "fc_mu.bias",
    "fc_sigma.bias", "fc6.weight", "fc6.bias", "fc7.weight", "fc7.bias", "fc8.weight", "fc8.bias",
    "fc9.weight", "fc9.bias", "fc10.weight", "fc10.bias", "fc11.weight", "fc11.bias",
]
BN_NAMES = ["batch_l1", "batch_l2", "batch_l3", "batch_l4", "batch_l5", "batch_s"]

WS_IDS = {name: i for i, name in enumerate([
    "x_low", "c_prob", "c", "c_smp", "s_mean", "s_logvar", "s_smp", "y_soft",
    "r1", "r2", "r3", "r4", "r5", "d6", "d7", "d8", "d9", "d10", "zin", "dz11", "dz1", "gzin", "gzc", "g5",
    "bn_mean1", "gd10_slab", "g1", "g2", "g3", "g4", "dz2", "dz3", "dz4", "dz5",
])}

LOSS_TOTAL, LOSS_JOINT, LOSS_CENT, LOSS_CDIST, LOSS_CL2, LOSS_REC0 = 0, 1, 2, 3, 4, 5


class Dims(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("A", "B", "D", "H", "L", "C", "S")]


class Hyper(C.Structure):
    _fields_ = [("tau", C.c_float), ("temp", C.c_float), ("beta", C.c_float), ("lam", C.c_float),
                ("eps", C.c_float), ("bn_momentum", C.c_float), ("x_drop", C.c_float), ("s_drop", C.c_float),
                ("hard", C.c_int32), ("training", C.c_int32), ("eval_flag", C.c_int32), ("gemm_bf16", C.c_int32),
                ("cat_mask", C.c_uint32 * 4)]


class Noise(C.Structure):
    _fields_ = [("mode", C.c_int32), ("_pad", C.c_int32), ("x_mask", C.c_void_p), ("u_gumbel", C.c_void_p),
                ("u_state", C.c_void_p), ("s_mask", C.c_void_p), ("seed", C.c_uint64), ("offset", C.c_uint64)]


class ParamLayout(C.Structure):
    _fields_ = [("per_arm", C.c_int64), ("offset", C.c_int64 * N_PARAM_TENSORS),
                ("rows", C.c_int64 * N_PARAM_TENSORS), ("cols", C.c_int64 * N_PARAM_TENSORS),
                ("bn_per_arm", C.c_int64), ("bn_mean_offset", C.c_int64 * N_BN),
                ("bn_var_offset", C.c_int64 * N_BN), ("bn_dim", C.c_int64 * N_BN)]


class AugDims(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("A", "B", "D", "N1", "N3", "N5", "Z", "NZ")]


class AugTensors(C.Structure):
    _fields_ = [("w", C.c_void_p * 11), ("b", C.c_void_p * 11), ("bn_mean", C.c_void_p * 10), ("bn_var", C.c_void_p * 10),
                ("w_mu", C.c_void_p), ("b_mu", C.c_void_p), ("w_sigma", C.c_void_p), ("b_sigma", C.c_void_p),
                ("bn_mu_mean", C.c_void_p), ("bn_mu_var", C.c_void_p), ("noise_w", C.c_void_p),
                ("bnz_weight", C.c_void_p), ("bnz_bias", C.c_void_p), ("bnz_mean", C.c_void_p), ("bnz_var", C.c_void_p)]


N_EVENTS = 8
N_TUNE = 24
# mmvae_exec.tune indices (private: csrc/tune.h; public: MMVAE_TUNE_ENGINE in include/mmvae.h) and the
# environment switch that sets each one: the LIBRARY reads
# no environment variables, this module translates them (experiments and A/B timing only; none is needed in production)
TUNE_ENV = {
    "MMVAE_EVAL_CHAIN": (0, lambda v: int(int(v) == 0)), "MMVAE_AUG_TILE": (3, int), "MMVAE_ABLATE_C": (4, int), "MMVAE_ABLATE": (5, int),
    "MMVAE_FC11_ZG": (8, lambda v: int(int(v) == 0)), "MMVAE_COUPLE_SIDE": (13, int), "MMVAE_ABLATE_L": (14, int), "MMVAE_ABLATE_B": (16, int),
    "MMVAE_BF16_NARROW_FP32": (18, int), "MMVAE_BN_PARTIALS": (19, int), "MMVAE_PRESPLIT_ALL": (20, int), "MMVAE_CHAIN_FP32": (21, int),
}
TUNE_ENGINE = 17    # MMVAE_TUNE_ENGINE: the GEMM engine the caller runs (the layout's split factors are chosen for it)


class Exec(C.Structure):
    """mmvae_exec: the caller-owned execution context of one engine (side stream, fork / join events, split factors,
    experiment switches)."""
    _fields_ = [("side_stream", C.c_void_p), ("ev", C.c_void_p * N_EVENTS), ("early_grad_event", C.c_void_p),
                ("early_recorded", C.c_int32), ("split", C.c_int32 * 6), ("tune", C.c_int32 * N_TUNE)]


def exec_from_env(engine: int = 0) -> Exec:
    """An Exec with the split factors (MMVAE_SPLIT<i>) and experiment switches the environment asks for; ``engine`` is the
    GEMM engine the caller is going to run (gemm_mode(...) & 0xFF)."""
    ex = Exec()
    ex.tune[TUNE_ENGINE] = engine & 0xFF
    for w in range(6):
        v = os.environ.get(f"MMVAE_SPLIT{w}")
        if v:
            ex.split[w] = int(v)
    for name, (idx, conv) in TUNE_ENV.items():
        v = os.environ.get(name)
        if v:
            ex.tune[idx] = conv(v)
    return ex


class NativeError(RuntimeError):
    pass


_lib = None


def lib():
    """Load libmmvae_hip.so; raises (never falls back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeError(
            f"{LIB_PATH} not found: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()' or python distributed-vae_amd/build.py). "
            "There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp, i64, f32, i32 = C.c_void_p, C.c_int64, C.c_float, C.c_int
    L.mmvae_abi_version.restype = C.c_int
    L.mmvae_last_error_string.restype = C.c_char_p
    L.mmvae_check_dims.argtypes = [C.POINTER(Dims)]
    L.mmvae_param_layout.argtypes = [C.POINTER(Dims), C.POINTER(ParamLayout)]
    ex = C.POINTER(Exec)
    L.mmvae_workspace_bytes.argtypes = [C.POINTER(Dims), ex]
    L.mmvae_workspace_bytes.restype = C.c_size_t
    L.mmvae_ws_offset.argtypes = [C.POINTER(Dims), ex, C.c_int]
    L.mmvae_ws_offset.restype = i64
    L.mmvae_ws_debug_offset.argtypes = [C.POINTER(Dims), ex]
    L.mmvae_ws_debug_offset.restype = i64
    L.mmvae_splits.argtypes = [C.POINTER(Dims), ex, C.POINTER(C.c_int32 * 6)]
    L.mmvae_forward.argtypes = [C.POINTER(Dims), C.POINTER(Hyper), C.POINTER(Noise), vp, vp, vp, vp, i64, vp, i32,
                                vp, C.c_size_t, ex, vp]
    L.mmvae_loss.argtypes = [C.POINTER(Dims), C.POINTER(Hyper), vp, C.c_size_t, vp, ex, vp]
    L.mmvae_backward.argtypes = [C.POINTER(Dims), C.POINTER(Hyper), C.POINTER(Noise), vp, vp, i64, f32, vp,
                                 C.c_size_t, vp, ex, vp]
    L.mmvae_adam_step.argtypes = [i64, vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, i32, vp]
    L.mmvae_train_step.argtypes = [C.POINTER(Dims), C.POINTER(Hyper), C.POINTER(Noise), vp, vp, vp, vp, i64, vp,
                                   C.c_size_t, vp, vp, i32, vp, vp, i64, f32, f32, f32, f32, f32, i32, ex, vp]
    L.mmvae_train_step_rows.argtypes = [C.POINTER(Dims), C.POINTER(Hyper), C.POINTER(Noise), vp, vp, vp, vp, vp, i64, i64, vp, vp,
                                        C.c_size_t, vp, vp, i32, vp, vp, i64, f32, f32, f32, f32, f32, i32, ex, vp]
    L.mmvae_train_step_rows.restype = C.c_int
    L.mmvae_to_bf16.argtypes = [vp, i64, i64, i32, vp, vp]
    L.mmvae_to_bf16.restype = C.c_int
    L.mmvae_debug_stage.argtypes = [C.POINTER(Dims), C.POINTER(Hyper), C.POINTER(Noise), i32, vp, vp, i64, vp,
                                    C.c_size_t, vp, ex, vp]
    L.mmvae_dump_noise.argtypes = [C.POINTER(Dims), C.POINTER(Hyper), C.POINTER(Noise), vp, vp, vp, vp, vp]
    L.mmvae_eval_classify.argtypes = [C.POINTER(Dims), C.POINTER(Hyper), vp, vp, vp, i64, vp, C.c_size_t, vp, vp, ex,
                                      vp]
    L.mmvae_classify.argtypes = [vp, i64, i32, vp, vp]
    L.mmvae_confmat_accumulate.argtypes = [vp, i32, i64, i32, vp, vp]
    L.mmvae_consensus.argtypes = [vp, i32, i32, vp, vp, vp]
    L.mmvae_aug_packed_floats.argtypes = [C.POINTER(AugDims)]
    L.mmvae_aug_packed_floats.restype = C.c_size_t
    L.mmvae_aug_workspace_bytes.argtypes = [C.POINTER(AugDims), i32]
    L.mmvae_aug_workspace_bytes.restype = C.c_size_t
    L.mmvae_aug_pack.argtypes = [C.POINTER(AugDims), C.POINTER(AugTensors), vp, vp]
    L.mmvae_augment.argtypes = [C.POINTER(AugDims), vp, vp, i64, vp, vp, f32, vp, C.c_size_t, vp, vp, i32, ex, vp]
    L.mmvae_gather_rows.argtypes = [vp, i64, i64, vp, i64, i32, vp, vp]
    L.mmvae_tp_planes_bytes.argtypes = [i64, i32, i32]
    L.mmvae_tp_planes_bytes.restype = C.c_size_t
    L.mmvae_tp_planes.argtypes = [vp, i64, i64, i32, i32, vp, vp]
    L.mmvae_augment_rows.argtypes = [C.POINTER(AugDims), vp, vp, i64, i32, vp, vp, vp, f32, vp, C.c_size_t, vp, vp, i32, ex, vp]
    for fn in ("mmvae_check_dims", "mmvae_param_layout", "mmvae_splits", "mmvae_forward", "mmvae_loss",
               "mmvae_backward", "mmvae_adam_step", "mmvae_train_step", "mmvae_dump_noise", "mmvae_debug_stage",
               "mmvae_eval_classify", "mmvae_classify", "mmvae_confmat_accumulate", "mmvae_consensus", "mmvae_aug_pack",
               "mmvae_augment", "mmvae_gather_rows", "mmvae_tp_planes", "mmvae_augment_rows"):
        getattr(L, fn).restype = C.c_int
    L.mmvae_dp_unique_id.argtypes = [vp]
    L.mmvae_dp_init.argtypes = [vp, i32, i32, C.POINTER(C.c_void_p)]
    L.mmvae_allreduce_grads.argtypes = [vp, vp, i64, vp]
    L.mmvae_dp_destroy.argtypes = [vp]
    for fn in ("mmvae_dp_unique_id", "mmvae_dp_init", "mmvae_allreduce_grads", "mmvae_dp_destroy"):
        getattr(L, fn).restype = C.c_int
    if L.mmvae_abi_version() != ABI_VERSION:
        raise NativeError("libmmvae_hip.so ABI version mismatch (rebuild: python distributed-vae_amd/build.py)")
    _lib = L
    return L


def check(rc: int, what: str):
    if rc != 0:
        msg = lib().mmvae_last_error_string().decode()
        if rc == -2:
            raise NotImplementedError(f"{what}: {msg}")
        raise NativeError(f"{what} failed (code {rc}): {msg}")


def param_layout(dims: Dims) -> ParamLayout:
    pl = ParamLayout()
    check(lib().mmvae_param_layout(C.byref(dims), C.byref(pl)), "mmvae_param_layout")
    return pl


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream(device=None):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def make_noise(explicit: Optional[Dict[str, torch.Tensor]] = None, seed: int = 0, offset: int = 0) -> Noise:
    """explicit: dict with x_mask (uint8 [A,B,D]), u_gumbel, u_state (float32), s_mask -> mode 0;
    otherwise Philox mode keyed by (seed, offset)."""
    n = Noise()
    if explicit is not None:
        n.mode = 0
        n.x_mask = explicit["x_mask"].data_ptr() if explicit.get("x_mask") is not None else None
        n.u_gumbel = explicit["u_gumbel"].data_ptr() if explicit.get("u_gumbel") is not None else None
        n.u_state = explicit["u_state"].data_ptr() if explicit.get("u_state") is not None else None
        n.s_mask = explicit["s_mask"].data_ptr() if explicit.get("s_mask") is not None else None
    else:
        n.mode = 1
        n.seed = seed & 0xFFFFFFFFFFFFFFFF
        n.offset = offset & 0xFFFFFFFFFFFFFFFF
    return n


_STREAMS: Dict = {}


def shared_stream(device, name: str) -> "torch.cuda.Stream":
    """One high-priority stream per (device, role) for the whole process.  HIP multiplexes streams onto a few hardware
    queues: every further stream an engine or a trainer created made it more likely that two streams meant to overlap
    share a queue (measured: the third engine of a process ran its steps in 2.0 ms instead of 1.05 ms).  Roles: "step"
    (dW11 / coupling beside the backward chain) and "produce" (next batch: gather, augmenter)."""
    key = (torch.device(device).index or 0, name)
    st = _STREAMS.get(key)
    if st is None:
        # (MMVAE_SIDE_PRIORITY / MMVAE_PRODUCE_PRIORITY: A/B timing; -1 = high)
        prio = os.environ.get("MMVAE_PRODUCE_PRIORITY" if name == "produce" else "MMVAE_SIDE_PRIORITY", "-1")
        st = torch.cuda.Stream(device=device, priority=int(prio))
        _STREAMS[key] = st
    return st


class Engine:
    """Owns the workspace and the execution context (mmvae_exec: side stream, events, split factors, switches) of one
    (dims, device) and issues the C-ABI calls on torch's current stream OF THAT DEVICE.  Nothing is shared between
    engines except the process-wide side stream of a device (see ``shared_stream``): each engine has its own events."""

    def __init__(self, A, B, D, H, L, Cc, S, device, ex: Optional[Exec] = None, gemm_engine: int = 0):
        self.dims = Dims(A, B, D, H, L, Cc, S)
        self.gemm_engine = gemm_engine & 0xFF      # which engine the layout's split factors are chosen for
        check(lib().mmvae_check_dims(C.byref(self.dims)), "mmvae_check_dims")
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise NativeError("the HIP engine needs a GPU device (no CPU fallback)")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        # the engine's OWN execution context: from a caller's Exec only the split factors and the switches are taken --
        # stream, events and the hook are per engine (two engines built from one Exec must not share them)
        self.ex = exec_from_env(self.gemm_engine)
        if ex is not None:
            for i in range(6):
                self.ex.split[i] = ex.split[i]
            for i in range(N_TUNE):
                self.ex.tune[i] = ex.tune[i]
        self.side = None
        self.early_event = None
        self._events = []
        with torch.cuda.device(self.device):
            self.ws_bytes = int(lib().mmvae_workspace_bytes(C.byref(self.dims), C.byref(self.ex)))
            self.ws = torch.empty(self.ws_bytes // 4, dtype=torch.float32, device=self.device)
            assert self.ws.data_ptr() % 256 == 0
            self.loss_buf = torch.zeros(5 + 3 * A, dtype=torch.float32, device=self.device)
            # side stream: the dW11 GEMM overlaps the latency-bound backward chain (MMVAE_SIDE_STREAM=0 disables)
            if os.environ.get("MMVAE_SIDE_STREAM", "1") != "0":
                # High priority: HIP maps streams of one priority onto a small set of hardware queues round-robin; once
                # RCCL has created its streams the side stream can share a queue with the main stream and the overlap
                # is silently lost (measured with an initialised process group: 1.146 ms per step against 1.028 ms with
                # a high-priority side stream; no difference without a process group).
                self.side = shared_stream(self.device, "step")
                self.ex.side_stream = self.side.cuda_stream
                for i in range(N_EVENTS):
                    ev = torch.cuda.Event()
                    ev.record(self.side)               # creates the underlying hipEvent_t on this device
                    self._events.append(ev)
                    self.ex.ev[i] = ev.cuda_event

    def _s(self):
        return _stream(self.device)

    def _x(self):
        return C.byref(self.ex)

    def enable_early_grad_event(self, on: bool = True):
        """Data-parallel overlap: the next train_step(do_adam=False) records ``self.early_event`` on the side stream
        once the fc11 gradients are final (see ``early_recorded``)."""
        if on and self.early_event is None and self.side is not None:
            with torch.cuda.device(self.device):
                self.early_event = torch.cuda.Event()
                self.early_event.record(self.side)     # creates the underlying hipEvent_t
            self.ex.early_grad_event = self.early_event.cuda_event
        elif not on:
            self.early_event = None
            self.ex.early_grad_event = None

    def early_recorded(self) -> bool:
        return bool(self.ex.early_recorded)

    def ws_view(self, name: str, width: int) -> torch.Tensor:
        off = int(lib().mmvae_ws_offset(C.byref(self.dims), self._x(), WS_IDS[name]))
        if off < 0:
            raise NativeError(f"unknown workspace region {name}")
        d = self.dims
        return self.ws[off: off + d.A * d.B * width].view(d.A, d.B, width)

    def splits(self):
        """Split factors of the layout (order of mmvae_exec.split)."""
        out = (C.c_int32 * 6)()
        check(lib().mmvae_splits(C.byref(self.dims), self._x(), C.byref(out)), "mmvae_splits")
        return list(out)

    def ws_raw(self, name: str, numel: int) -> torch.Tensor:
        off = int(lib().mmvae_ws_offset(C.byref(self.dims), self._x(), WS_IDS[name]))
        return self.ws[off: off + numel]

    def ws_debug(self, numel: int = 1024) -> torch.Tensor:
        off = int(lib().mmvae_ws_debug_offset(C.byref(self.dims), self._x()))
        return self.ws[off: off + numel]

    def forward(self, hyper: Hyper, noise: Noise, params, bn_running, nbt, x, x_arm_stride, x_rec, need_grad):
        check(lib().mmvae_forward(C.byref(self.dims), C.byref(hyper), C.byref(noise), _ptr(params), _ptr(bn_running),
                                  _ptr(nbt), _ptr(x), x_arm_stride, _ptr(x_rec), int(need_grad), _ptr(self.ws),
                                  self.ws_bytes, self._x(), self._s()), "mmvae_forward")

    def loss(self, hyper: Hyper) -> torch.Tensor:
        check(lib().mmvae_loss(C.byref(self.dims), C.byref(hyper), _ptr(self.ws), self.ws_bytes, _ptr(self.loss_buf),
                               self._x(), self._s()), "mmvae_loss")
        return self.loss_buf

    def backward(self, hyper: Hyper, noise: Noise, params, x, x_arm_stride, grads, grad_scale=1.0):
        check(lib().mmvae_backward(C.byref(self.dims), C.byref(hyper), C.byref(noise), _ptr(params), _ptr(x),
                                   x_arm_stride, float(grad_scale), _ptr(self.ws), self.ws_bytes, _ptr(grads),
                                   self._x(), self._s()), "mmvae_backward")

    def train_step(self, hyper, noise, params, bn_running, nbt, x, x_arm_stride, grads, do_adam, exp_avg,
                   exp_avg_sq, step, lr, b1=0.9, b2=0.999, adam_eps=1e-8, wd=0.0, decoupled=False):
        check(lib().mmvae_train_step(C.byref(self.dims), C.byref(hyper), C.byref(noise), _ptr(params),
                                     _ptr(bn_running), _ptr(nbt), _ptr(x), x_arm_stride, _ptr(self.ws), self.ws_bytes,
                                     _ptr(grads), _ptr(self.loss_buf), int(do_adam), _ptr(exp_avg), _ptr(exp_avg_sq),
                                     int(step), lr, b1, b2, adam_eps, wd, int(decoupled), self._x(), self._s()),
              "mmvae_train_step")
        return self.loss_buf

    def train_step_rows(self, hyper, noise, params, bn_running, nbt, data, rows, grads, do_adam, exp_avg,
                        exp_avg_sq, step, lr, b1=0.9, b2=0.999, adam_eps=1e-8, wd=0.0, decoupled=False, data16=None):
        """The fused step on a batch that is never materialised: cell b = row rows[b] of the resident matrix ``data``
        (mmvae_train_step_rows).  ``data16``: the matrix's bf16 copy (``to_bf16``; bf16 engine only).  Raises
        NotImplementedError where the library does not offer it (gather then)."""
        assert data.dim() == 2 and data.stride(1) == 1 and rows.dtype == torch.int64 and rows.numel() == self.dims.B
        if data16 is not None:
            assert (data16.dtype == torch.bfloat16 and data16.shape == data.shape and data16.stride() == data.stride()
                    and data16.device == data.device)
        check(lib().mmvae_train_step_rows(C.byref(self.dims), C.byref(hyper), C.byref(noise), _ptr(params),
                                          _ptr(bn_running), _ptr(nbt), _ptr(data), _ptr(data16), int(data.stride(0)), int(data.shape[0]),
                                          _ptr(rows), _ptr(self.ws), self.ws_bytes, _ptr(grads), _ptr(self.loss_buf),
                                          int(do_adam), _ptr(exp_avg), _ptr(exp_avg_sq), int(step), lr, b1, b2, adam_eps, wd,
                                          int(decoupled), self._x(), self._s()), "mmvae_train_step_rows")
        return self.loss_buf

    def eval_classify(self, hyper: Hyper, params, bn_running, x, x_arm_stride, labels, counts=None):
        """Encoder + latent block in eval mode, labels[a, b] = argmax c; counts (int64 [pairs, C, C]) accumulate."""
        check(lib().mmvae_eval_classify(C.byref(self.dims), C.byref(hyper), _ptr(params), _ptr(bn_running), _ptr(x),
                                        x_arm_stride, _ptr(self.ws), self.ws_bytes, _ptr(labels), _ptr(counts),
                                        self._x(), self._s()), "mmvae_eval_classify")
        return labels

    def debug_stage(self, stage: int, hyper: Hyper, noise: Noise, params, x, x_arm_stride, grads=None):
        check(lib().mmvae_debug_stage(C.byref(self.dims), C.byref(hyper), C.byref(noise), int(stage), _ptr(params),
                                      _ptr(x), x_arm_stride, _ptr(self.ws), self.ws_bytes, _ptr(grads), self._x(),
                                      self._s()), "mmvae_debug_stage")

    def dump_noise(self, hyper: Hyper, noise: Noise):
        d = self.dims
        xm = torch.empty(d.A, d.B, d.D, dtype=torch.uint8, device=self.device)
        ug = torch.empty(d.A, d.B, d.C, dtype=torch.float32, device=self.device)
        us = torch.empty(d.A, d.B, d.S, dtype=torch.float32, device=self.device)
        sm = torch.empty(d.A, d.B, d.S, dtype=torch.uint8, device=self.device)
        check(lib().mmvae_dump_noise(C.byref(self.dims), C.byref(hyper), C.byref(noise), _ptr(xm), _ptr(ug), _ptr(us),
                                     _ptr(sm), self._s()), "mmvae_dump_noise")
        return {"x_mask": xm, "u_gumbel": ug, "u_state": us, "s_mask": sm}


def adam_step(params, grads, exp_avg, exp_avg_sq, step, lr, b1=0.9, b2=0.999, eps=1e-8, wd=0.0, decoupled=False):
    check(lib().mmvae_adam_step(params.numel(), _ptr(params), _ptr(grads), _ptr(exp_avg), _ptr(exp_avg_sq), int(step),
                                lr, b1, b2, eps, wd, int(decoupled), _stream(params.device)), "mmvae_adam_step")


def classify(probs: torch.Tensor) -> torch.Tensor:
    """argmax over the last axis of a float32 CUDA tensor -> int32 labels (first maximum on ties)."""
    if probs.device.type != "cuda":
        raise NativeError("classify needs a CUDA tensor (no CPU fallback)")
    p = probs.contiguous().float()
    n, Cc = p.numel() // p.shape[-1], p.shape[-1]
    out = torch.empty(p.shape[:-1], dtype=torch.int32, device=p.device)
    check(lib().mmvae_classify(_ptr(p), n, Cc, _ptr(out), _stream(p.device)), "mmvae_classify")
    return out


def confmat_accumulate(labels: torch.Tensor, Cc: int, counts: Optional[torch.Tensor] = None) -> torch.Tensor:
    """labels int32 [A, n] on the GPU -> counts int64 [A(A-1)/2, C, C] (+= when given)."""
    if labels.device.type != "cuda":
        raise NativeError("confmat_accumulate needs CUDA tensors (no CPU fallback)")
    lab = labels.contiguous().to(torch.int32)
    A, n = lab.shape
    if counts is None:
        counts = torch.zeros(max(A * (A - 1) // 2, 1), Cc, Cc, dtype=torch.int64, device=lab.device)
    check(lib().mmvae_confmat_accumulate(_ptr(lab), A, n, Cc, _ptr(counts), _stream(lab.device)),
          "mmvae_confmat_accumulate")
    return counts


def consensus(counts: torch.Tensor, want_norm: bool = False):
    """counts int64 [pairs, C, C] -> consensus float64 [pairs] (and the normalised matrices when want_norm)."""
    if counts.device.type != "cuda":
        raise NativeError("consensus needs CUDA tensors (no CPU fallback)")
    cnt = counts.contiguous()
    P, Cc, _ = cnt.shape
    out = torch.empty(P, dtype=torch.float64, device=cnt.device)
    norm = torch.empty(P, Cc, Cc, dtype=torch.float64, device=cnt.device) if want_norm else None
    check(lib().mmvae_consensus(_ptr(cnt), P, Cc, _ptr(norm), _ptr(out), _stream(cnt.device)), "mmvae_consensus")
    return (out, norm) if want_norm else out


def to_bf16(data: torch.Tensor) -> torch.Tensor:
    """The bf16 copy of a resident float32 matrix (round to nearest even; same shape and strides in elements) for the bf16
    engine's row-indexed step (mmvae_to_bf16; made once per data set)."""
    if data.device.type != "cuda":
        raise NativeError("to_bf16 needs a CUDA tensor (no CPU fallback)")
    assert data.dim() == 2 and data.dtype == torch.float32 and data.stride(1) == 1
    ld = int(data.stride(0))
    # (a column-offset view of a wider matrix spans (rows - 1) * ld + D elements, not rows * ld: the copy owns exactly that,
    # rounded up to whole 8-byte pieces; the kernel touches only the rows' own columns)
    span = (data.shape[0] - 1) * ld + ((data.shape[1] + 3) // 4) * 4
    out = torch.empty(span, dtype=torch.bfloat16, device=data.device).as_strided(data.shape, data.stride())
    check(lib().mmvae_to_bf16(_ptr(data), ld, data.shape[0], data.shape[1], _ptr(out), _stream(data.device)), "mmvae_to_bf16")
    return out


def tp_planes(data: torch.Tensor, n_planes: int) -> Optional[torch.Tensor]:
    """A resident float32 matrix as the planes x planes GEMM engine's tiled bf16 slice planes (mmvae_tp_planes; made once per
    data set: 3 planes = the exact slices of the fp32x3 engine, 1 = the matrix rounded to bf16) for the augmenter's row-indexed
    forward (mmvae_augment_rows).  Returns an opaque uint16 device tensor, or None where the library does not offer it."""
    if data.device.type != "cuda":
        raise NativeError("tp_planes needs a CUDA tensor (no CPU fallback)")
    assert data.dim() == 2 and data.dtype == torch.float32 and data.stride(1) == 1
    nbytes = int(lib().mmvae_tp_planes_bytes(data.shape[0], data.shape[1], n_planes))
    if nbytes == 0:
        return None
    out = torch.zeros(nbytes // 2, dtype=torch.int16, device=data.device)     # (padding rows of the planes are read, never used)
    check(lib().mmvae_tp_planes(_ptr(data), int(data.stride(0)), data.shape[0], data.shape[1], n_planes, _ptr(out),
                                _stream(data.device)), "mmvae_tp_planes")
    return out


def gather_rows(data: torch.Tensor, idx: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[i] = data[idx[i]] for a 2-D float32 CUDA matrix (rows may be strided) and int64 CUDA indices."""
    if data.device.type != "cuda" or idx.device.type != "cuda":
        raise NativeError("gather_rows needs CUDA tensors (no CPU fallback)")
    assert data.dim() == 2 and data.dtype == torch.float32 and data.stride(1) == 1
    idx = idx.to(torch.int64).contiguous()
    n, Dm = idx.numel(), data.shape[1]
    if out is None:
        out = torch.empty(n, Dm, dtype=torch.float32, device=data.device)
    if n:
        check(lib().mmvae_gather_rows(_ptr(data), data.stride(0), data.shape[0], _ptr(idx), n, Dm, _ptr(out),
                                      _stream(data.device)), "mmvae_gather_rows")
    return out


# ---------------------------------------------------------------------------------------------------------------
# operand type of the large GEMMs (mmvae_hyper.gemm_bf16 / mmvae_augment's gemm_bf16 argument)
#   "fp32"       fp32 results; the library's fastest fp32-grade engine ("fp32x3", see FP32_ENGINE below; shapes the split
#                engine does not take -- fc_dim > 111, D % 4 != 0 -- run the fp32 matrix instruction by themselves)
#   "fp32_mfma"  fp32 operands on the fp32 matrix instruction (v_mfma_f32_32x32x2_f32: an exact fmaf chain)
#   "fp32x3"     fp32 operands split exactly into three bf16 slices, six slice products per product on the bf16 matrix
#                pipe, fp32 accumulation: truncation <= 2^-26 per product, below the fp32 rounding of the accumulation
#   "bf16"       operands rounded to bf16 (BASELINE.json's bf16 configuration)
# MMVAE_FP32_ENGINE=fp32_mfma|fp32x3 picks what "fp32" means (A/B timing, and running the parity suite on either).
# ---------------------------------------------------------------------------------------------------------------
FP32_ENGINE = os.environ.get("MMVAE_FP32_ENGINE", "fp32x3")
_GEMM_MODES = {"fp32_mfma": 0, "bf16": 1, "fp32x3": 2}


def gemm_mode(dtype: str) -> int:
    if dtype == "fp32":
        dtype = FP32_ENGINE
    if dtype not in _GEMM_MODES:
        raise ValueError(f"gemm_dtype must be 'fp32', 'fp32_mfma', 'fp32x3' or 'bf16', got {dtype!r}")
    mode = _GEMM_MODES[dtype]
    if mode == 2:   # diagnostics: MMVAE_X3_OFF=<mask> keeps single products on the fp32 matrix instruction (1 fc1, 2 fc11, 4 dW1, 8 dW11)
        mode |= (int(os.environ.get("MMVAE_X3_OFF", "0")) & 0xF) << 8
    return mode
