"""CPU (no GPU): host logic of the drop-in boundary and the C-ABI library's exported surface."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

import distributed_vae_amd  # noqa: F401
from distributed_vae_amd import _native as N
from distributed_vae_amd.cpl_mixvae import FusedAdam, cpl_mixVAE
from distributed_vae_amd.nn_model import VAEConfig, mixVAE_model, mk_vae
from oracle import restatement as R
from tests import golden_util as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "mmvae.h")).read()
    names = sorted(set(re.findall(r"\b(mmvae_[a-z_]+)\s*\(", hdr)))
    assert len(names) >= 13
    L = N.lib()
    for n in names:
        assert hasattr(L, n), n
    assert L.mmvae_abi_version() == N.ABI_VERSION == 4
    # the library keeps no process-wide or per-thread knobs: no setters, and it reads no environment variables
    assert not [n for n in names if n.startswith("mmvae_set_")]
    blob = open(N.LIB_PATH, "rb").read()
    assert b"getenv" not in blob and b"MMVAE_" not in blob


def test_param_layout_matches_state_dict_shapes():
    d = N.Dims(2, 5000, 5000, 100, 10, 92, 2)
    lay = N.param_layout(d)
    h = R.Hyper()
    shapes = h.linear_shapes()
    total = 0
    for t, nm in enumerate(N.PARAM_NAMES):
        layer, kind = nm.split(".")
        o, i = shapes[layer]
        if kind == "weight":
            assert (lay.rows[t], lay.cols[t]) == (o, i), nm
            total += o * i
        else:
            assert (lay.rows[t], lay.cols[t]) == (o, 1), nm
            total += o
        assert lay.offset[t] % 4 == 0 or t in (13, 15)
    assert total == 1070184                                 # parameters per arm at D=5000
    assert lay.per_arm >= total and lay.per_arm % 64 == 0
    # the state head is one [2S, L+C] matrix for the kernels
    assert lay.offset[13] == lay.offset[12] + 2 * 102 and lay.offset[15] == lay.offset[14] + 2
    assert N.lib().mmvae_workspace_bytes(C.byref(d), None) > 200e6
    # split factors travel in the caller's mmvae_exec and change the layout
    ex = N.Exec()
    base = (C.c_int32 * 6)()
    assert N.lib().mmvae_splits(C.byref(d), C.byref(ex), C.byref(base)) == 0 and all(v >= 1 for v in base)
    ex.split[2] = base[2] + 3
    out = (C.c_int32 * 6)()
    assert N.lib().mmvae_splits(C.byref(d), C.byref(ex), C.byref(out)) == 0 and out[2] == base[2] + 3
    assert N.lib().mmvae_workspace_bytes(C.byref(d), C.byref(ex)) > N.lib().mmvae_workspace_bytes(C.byref(d), None)


def test_constructor_matches_reference_initialisation():
    """Same torch seed -> same parameters as the reference constructor (fixture sd0 comes from the
    reference's mixVAE_model.__init__, nn_model.py:184-203) and the same 46-keys-per-arm layout."""
    for name in ["tiny_a2", "tiny_a5_hard"]:
        g = G.load(name)
        h = G.hyper_of(g)
        torch.manual_seed(int(g["seed"]))
        m = mixVAE_model(input_dim=h.input_dim, fc_dim=h.fc_dim, n_categories=h.n_categories, state_dim=h.state_dim,
                         lowD_dim=h.lowD_dim, x_drop=0.5, s_drop=h.s_drop, n_arm=h.n_arm, lam=1, lam_pc=1, tau=0.005,
                         beta=1.0, hard=h.hard, variational=True, device="cpu", eps=1e-8, momentum=0.01,
                         ref_prior=False, loss_mode="MSE")
        ref = G.state_dict_of(g)
        sd = m.state_dict()
        assert list(sd.keys()) == list(ref.keys())
        assert len(sd) == 46 * h.n_arm
        for k in ref:
            assert torch.equal(sd[k], ref[k]), k
        # packing into the flat buffer keeps values and makes parameters views of it
        flat = m.flat_parameters()
        assert m._is_packed()
        for k, v in m.state_dict().items():
            assert torch.equal(v, ref[k]), k
        assert m.fc11[h.n_arm - 1].bias.untyped_storage().data_ptr() == flat.untyped_storage().data_ptr()
        assert [p.shape for p in m.parameters()] == [ref[k].shape for k in R.param_keys(h)]


def test_mk_vae_and_config_defaults():
    cfg = VAEConfig()
    assert (cfg.n_categories, cfg.state_dim, cfg.input_dim, cfg.fc_dim, cfg.lowD_dim) == (92, 2, 5032, 100, 10)
    assert (cfg.tau, cfg.momentum, cfg.mode) == (0.005, 0.01, "MSE")
    m = mk_vae(10, 2, 784, "cpu", A=5)                      # the notebook call, mmidas/_mnist.ipynb cell 3
    assert m.n_arm == 5 and m.fcc[0].bias.shape == (10,) and m.x_dp.p == 0.5 and m.s_dp.p == 0.2


def test_forward_without_gpu_fails_loudly():
    m = mk_vae(7, 2, 64, "cpu", fc_dim=16, latent_dim=5)
    x = torch.zeros(8, 64)
    with pytest.raises(N.NativeError):
        m(x.expand(2, -1, -1), 1.0)
    with pytest.raises(AssertionError):
        m([x], 1.0)                                          # len(x) == n_arm, nn_model.py:317


def test_unsupported_paths_raise():
    m = mk_vae(7, 2, 64, "cpu", fc_dim=16, latent_dim=5, mode="ZINB")
    with pytest.raises(AssertionError):
        m([torch.zeros(4, 64)] * 2, 1.0)                     # ZINB rejected, nn_model.py:315
    with pytest.raises(FileNotFoundError):
        cpl_mixVAE(aug_file="some.pth", device="cpu")           # the augmenter checkpoint is loaded (cpl_mixvae.py:183)
    with pytest.raises(NotImplementedError):
        N.Engine(2, 32, 64, 300, 5, 7, 2, "cuda:0") if False else N.check(-2, "x")


def test_fused_adam_state_dict_is_torch_compatible():
    m = mk_vae(7, 2, 64, "cpu", fc_dim=16, latent_dim=5)
    opt = FusedAdam(m, lr=2e-3)
    opt._bind()
    opt.step_count = 4
    opt.exp_avg.uniform_(-1, 1)
    opt.exp_avg_sq.uniform_(0, 1)
    sd = opt.state_dict()
    ref = torch.optim.Adam(m.parameters(), lr=1e-3)
    ref.load_state_dict(sd)                                  # torch accepts the format
    ps = list(m.parameters())
    assert ref.param_groups[0]["lr"] == 2e-3
    for i, p in enumerate(ps):
        assert torch.equal(ref.state[p]["exp_avg"], sd["state"][i]["exp_avg"])
        assert ref.state[p]["exp_avg"].shape == p.shape
    # and back: a torch.optim.Adam state loads into FusedAdam
    opt2 = FusedAdam(m, lr=1e-3)
    opt2.load_state_dict(ref.state_dict())
    assert opt2.step_count == 4
    for a, b in zip(opt2._views(opt2.exp_avg) + opt2._views(opt2.exp_avg_sq),
                    opt._views(opt.exp_avg) + opt._views(opt.exp_avg_sq)):
        assert torch.equal(a, b)                             # (alignment gaps of the flat buffer excluded)


def test_philox_host_reference_vector():
    """Philox4x32-10 known-answer vectors (Random123 kat_vectors): the device code uses the same
    round function; this pins the constants/rounds through a pure-Python restatement and the
    keep-threshold convention."""
    M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85

    def philox(c, k):
        c = list(c); k = list(k)
        for _ in range(10):
            p0, p1 = M0 * c[0], M1 * c[2]
            c = [((p1 >> 32) ^ c[1] ^ k[0]) & 0xFFFFFFFF, p1 & 0xFFFFFFFF, ((p0 >> 32) ^ c[3] ^ k[1]) & 0xFFFFFFFF,
                 p0 & 0xFFFFFFFF]
            k = [(k[0] + W0) & 0xFFFFFFFF, (k[1] + W1) & 0xFFFFFFFF]
        return c

    assert philox([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert philox([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_widened_entry_points_validate_arguments_on_the_host():
    """The consensus / augmenter / data-path entry points reject bad arguments before touching the GPU (so this runs
    without one), with the documented codes."""
    L = N.lib()
    # consensus
    assert L.mmvae_consensus(None, 1, 4, None, None, None) == -1
    buf = (C.c_int64 * 4)()
    out = (C.c_double * 1)()
    assert L.mmvae_consensus(buf, 1, 200, None, out, None) == -2          # C > 128
    assert b"128" in L.mmvae_last_error_string()
    assert L.mmvae_confmat_accumulate(None, 2, 10, 4, buf, None) == -1
    assert L.mmvae_classify(None, 10, 4, None, None) == -1
    # augmenter
    ok = N.AugDims(2, 100, 5000, 1000, 500, 100, 10, 50)
    assert L.mmvae_aug_packed_floats(C.byref(ok)) > 5000 * 1000 * 2
    assert L.mmvae_aug_workspace_bytes(C.byref(ok), 1) < L.mmvae_aug_workspace_bytes(C.byref(ok), 0)
    assert L.mmvae_aug_packed_floats(C.byref(N.AugDims(2, 100, 5002, 1000, 500, 100, 10, 50))) == 0      # D % 4
    assert L.mmvae_aug_packed_floats(C.byref(N.AugDims(2, 100, 5000, 1000, 1000, 200, 10, 50))) == 0     # n/5 > 128
    assert L.mmvae_aug_packed_floats(C.byref(N.AugDims(2, 100, 400, 80, 640, 128, 64, 128))) == 0        # LDS
    assert L.mmvae_aug_pack(C.byref(ok), None, None, None) == -1
    assert L.mmvae_augment(C.byref(ok), None, None, 0, None, None, 0.1, None, 0, None, None, 0, None, None) == -1
    # data path
    assert L.mmvae_gather_rows(None, 8, 4, None, 2, 8, None, None) == -1
    # the resident matrix as the augmenter engine's slice planes, and the forward on rows of it
    assert L.mmvae_tp_planes_bytes(50000, 5000, 3) == 3 * 313 * 50176 * 32 and L.mmvae_tp_planes_bytes(50000, 5000, 1) == 313 * 50176 * 32
    assert L.mmvae_tp_planes_bytes(50000, 5000, 2) == 0 and L.mmvae_tp_planes_bytes(50000, 5002, 3) == 0       # planes, K % 4
    assert L.mmvae_tp_planes_bytes(500000, 5000, 3) == 0                                                       # a plane of 4 GB or more
    assert L.mmvae_tp_planes(None, 5000, 100, 5000, 3, None, None) == -1
    assert L.mmvae_augment_rows(C.byref(ok), None, None, 100, 3, None, None, None, 0.1, None, 0, None, None, 2, None, None) == -1
    # eval labels: needs eval mode
    d = N.Dims(2, 32, 64, 16, 4, 6, 2)
    h = N.Hyper(0.005, 1.0, 1.0, 1.0, 1e-8, 0.01, 0.5, 0.0, 0, 1, 0)     # training = 1
    one = (C.c_float * 4)()
    assert L.mmvae_eval_classify(C.byref(d), C.byref(h), one, one, one, 0, one, 16, one, None, None, None) in (-2, -4)


def test_bench_refuses_more_gpus_than_visible():
    """``python bench.py --gpus N`` without a launcher starts its own ranks; with fewer than N devices it must fail
    loudly instead of running the job on fewer GPUs and printing ``n_gpus: 1``."""
    import subprocess
    import sys
    n = torch.cuda.device_count() + 2
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n)], capture_output=True, text=True,
                       timeout=300, env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
    assert r.returncode != 0 and "refusing" in r.stderr and r.stdout.strip() == ""


def test_gemm_engine_names_map_to_the_abi_values(monkeypatch):
    """mmvae_hyper.gemm_bf16: 0 fp32 matrix instruction, 1 bf16 operands, 2 fp32x3 (include/mmvae.h); "fp32" is the
    library's fp32 engine (fp32x3 unless MMVAE_FP32_ENGINE says otherwise); the diagnostic mask rides in bits 8.."""
    from distributed_vae_amd import _native as N
    assert N.gemm_mode("fp32_mfma") == 0 and N.gemm_mode("bf16") == 1 and N.gemm_mode("fp32x3") == 2
    monkeypatch.setattr(N, "FP32_ENGINE", "fp32x3")
    assert N.gemm_mode("fp32") == 2
    monkeypatch.setattr(N, "FP32_ENGINE", "fp32_mfma")
    assert N.gemm_mode("fp32") == 0
    monkeypatch.setenv("MMVAE_X3_OFF", "5")
    assert N.gemm_mode("fp32x3") == 2 | (5 << 8) and N.gemm_mode("bf16") == 1
    with pytest.raises(ValueError):
        N.gemm_mode("fp16")
    ex = N.exec_from_env(2)
    assert ex.tune[N.TUNE_ENGINE] == 2


def test_split_factors_follow_the_engine_hint():
    """mmvae_exec.tune[MMVAE_TUNE_ENGINE] = 2: the layout's split factors are chosen for the fp32x3 engine's one-workgroup-
    per-CU kernels (A = 2, B = D = 5000: fc1 6, small layers 20, fc11 gene split 3, dW1 6; dW11 4 -- about 140 workgroups
    beside the backward chain when the chain kernels run their own GEMMs on the split engine too, 2 -- about 96 -- when
    they stay on the fp32 matrix instruction)."""
    from distributed_vae_amd import _native as N
    d = N.Dims(2, 5000, 5000, 100, 10, 92, 2)
    got = {}
    for eng, chain_fp32 in ((0, 0), (2, 0), (2, 1)):
        ex = N.Exec()
        ex.tune[N.TUNE_ENGINE] = eng
        ex.tune[21] = chain_fp32                    # MMVAE_TUNE_CHAIN_FP32
        sp = (C.c_int32 * 6)()
        N.check(N.lib().mmvae_splits(C.byref(d), C.byref(ex), C.byref(sp)), "mmvae_splits")
        got[eng, chain_fp32] = list(sp)
    assert got[0, 0] == [6, 6, 12, 32, 6, 5], got
    assert got[2, 0] == [6, 6, 6, 20, 3, 4], got
    assert got[2, 1] == [6, 6, 6, 20, 3, 2], got
    assert N.TUNE_ENV["MMVAE_CHAIN_FP32"][0] == 21


def test_batch_size_cap_of_the_exact_accumulators():
    """Training batches above 4096 x 8 = 32768 cells per rank are refused (MMVAE_E_UNSUPPORTED -> NotImplementedError in the
    Python layer): the fixed-point batch-sum accumulators hold 4096 addends per column and the producer with the fewest cells
    per workgroup adds 8 at a time (csrc/common.hpp ACC_MAX_ADDENDS / ACC_MIN_PRODUCER_ROWS).  The reference has no such cap
    (README, "differences").  No GPU needed: the check precedes every launch; an accepted size gets as far as the
    workspace-size check (a zero-byte workspace here)."""
    L = N.lib()
    hyper = N.Hyper()
    hyper.training = 1
    hyper.x_drop = 0.5
    buf = (C.c_char * 1024)()
    ws = (C.addressof(buf) + 255) & ~255
    loss = (C.c_float * 64)()
    E_UNSUPPORTED, E_WORKSPACE = -2, -4          # include/mmvae.h
    for B, want in ((32768, E_WORKSPACE), (32769, E_UNSUPPORTED)):
        d = N.Dims(2, B, 256, 32, 6, 12, 2)
        rc = L.mmvae_loss(C.byref(d), C.byref(hyper), C.c_void_p(ws), C.c_size_t(0), loss, None, None)
        assert rc == want, (B, rc, L.mmvae_last_error_string())
    assert b"32768" in L.mmvae_last_error_string()
    hyper.training = 0                      # eval mode has no batch sums: no cap
    d = N.Dims(2, 40000, 256, 32, 6, 12, 2)
    assert L.mmvae_loss(C.byref(d), C.byref(hyper), C.c_void_p(ws), C.c_size_t(0), loss, None, None) == E_WORKSPACE


def test_to_bf16_view_span_is_what_the_copy_allocates():
    """A column-offset view big[:, off:] spans (rows - 1) * ld + D elements, not rows * ld: _native.to_bf16 allocates exactly that
    (rounded up to whole 8-byte pieces) and the kernel touches only each row's own columns (csrc/datapath.hip k_to_bf16)."""
    import inspect
    src = inspect.getsource(N.to_bf16)
    assert "(data.shape[0] - 1) * ld" in src
    big = torch.zeros(6, 32)
    v = big[:, 8:24]
    span = (v.shape[0] - 1) * v.stride(0) + ((v.shape[1] + 3) // 4) * 4
    assert span == 5 * 32 + 16 and v.storage_offset() + span <= big.numel()
