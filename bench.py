#!/usr/bin/env python3
"""bench.py -- cells/sec through one full cpl-mixVAE train step (forward + loss + backward + Adam).

Workload (BASELINE.json configs[1], SURVEY.md section 8d "synthetic-10x-v1"): A=2 arms, per-GPU batch
B=5000 cells x D=5000 genes, H=100, L=10, C=92, S=2, fp32, x_drop=0.5, in-kernel Philox noise, the batch
already resident in HBM.  A "step" = one batch through mmvae_train_step (cpl_mixvae.py:434-463).

  python bench.py --gpus N --steps K --warmup W

N > 1 either runs under a launcher (torch.distributed.run sets RANK / LOCAL_RANK / WORLD_SIZE) or -- when those are
absent -- starts its own N rank processes (one per GPU, before this process has touched any GPU; the reference does the
same with mp.spawn, train.py:286) and exits with their status.  Fewer than N visible devices: non-zero exit.

N>1 is plain data parallelism: each rank trains on its own shard of cells, ONE RCCL all-reduce
(average) of the flat gradient buffer per step, Adam on every rank ("weak" scaling: per-GPU batch fixed).

Prints ONE JSON line (rank 0).  Besides the contract fields it carries
  "roofline":     dominant kernel (fc11 forward + loss + dZ11 + d(d10) in one launch: k_x3_fc11g under the default fp32x3
                  engine, k_fc11_zg under --gemm-dtype fp32_mfma): ALGORITHMIC FLOPs (A 4 B D H) and bytes (A 8 B D) per
                  launch / its average duration measured with HIP events on the launch stream.  `bound` / `achieved` /
                  `peak` / `frac` are the BINDING roof for the engine in use -- the one that needs more time for the
                  algorithmic work: HBM (8 TB/s) against the matrix pipe at the engine's ceiling (fp32 matrix instruction
                  157.3 TFLOP/s; fp32x3: 2.5 PFLOP/s bf16 dense / 6 slice products = 416.7 TFLOP/s fp32-equivalent);
                  "hbm" and "mfma" carry both fractions ("mfma.frac_of_fp32_mfma_peak" = rounds 1-2's figure), "executed"
                  the bf16 MFMAs actually issued (six slice products on padded tiles), "step" the whole step against
                  both roofs by SURVEY.md 8(d)'s per-cell arithmetic;
  "cpu_baseline": the oracle (oracle/restatement.py, kind "port": the reference's arithmetic restated,
                  pinned to the reference by tests/) timed on this box's host cores on the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")   # kernel arguments in device memory (before HIP init)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters
PEAK_HBM_GBS = 8000.0
PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense (same guide)


def flops_per_cell_arm(D, H, L, C, S):
    """MAC_fwd = 2DH + 6H^2 + 2HL + LC + 2(L+C)S + (S+C)L;
    FLOP/cell-arm = 3*2*MAC_fwd - 2DH (backward = 2x forward minus the unused dX of fc1)."""
    mac = 2 * D * H + 6 * H * H + 2 * H * L + L * C + 2 * (L + C) * S + (S + C) * L
    return 6 * mac - 2 * D * H


def bytes_per_cell_arm(D, H, P, B):
    """20 D + 80 H + 36 P / B  (fp32)."""
    return 20 * D + 80 * H + 36 * P / B


def synthetic_rows(n, d, seed, device):
    """synthetic-10x-v1, generated on the device in chunks."""
    g = torch.Generator(device=device).manual_seed(seed)
    x = (torch.rand(n, d, generator=g, device=device) < 0.2).float()
    x *= torch.randn(n, d, generator=g, device=device).abs() * 3.0
    return x


def cpu_baseline(args, D, H, L, C, S, A, B):
    """The oracle timed on the host cores (rank 0, N=1 only), bounded to ~10-30 s, in the two variants BASELINE.md section 3
    asks for: "faithful" -- every step draws its dropout keep-masks with ``bernoulli_`` (what ``nn.Dropout`` does inside
    the reference's forward, nn_model.py:264; 62 % of the reference's CPU step) and its uniforms with ``rand`` -- and
    "masks precomputed" -- the same step with the noise drawn outside the timed region."""
    from oracle import restatement as R

    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    # a one-GPU box exposes every host core but grants a 16-core CPU share; more threads than that
    # only oversubscribe (measured: 256 threads -> 74 s/step)
    cores = min(cores, args.cpu_threads)
    torch.set_num_threads(cores)
    h = R.Hyper(input_dim=D, fc_dim=H, n_categories=C, state_dim=S, lowD_dim=L, n_arm=A)
    x = R.synthetic_batch(B, D)

    def draw():
        # the reference's consumption order per arm: bernoulli_[B,D] -> rand[B,C] -> rand_like[B,S]
        nz = {"x_mask": [], "u_gumbel": [], "u_state": [], "s_mask": []}
        for _ in range(A):
            nz["x_mask"].append(torch.empty(B, D).bernoulli_(1.0 - h.x_drop).to(torch.uint8))
            nz["u_gumbel"].append(torch.rand(B, C))
            nz["u_state"].append(torch.rand(B, S))
        return nz

    def run(faithful, budget_s):
        sd = R.init_state_dict(h, 546)
        st = None
        times = []
        fixed = draw()
        for s in range(args.cpu_steps + 1):
            t0 = time.time()
            noise = draw() if faithful else fixed
            _, st = R.train_steps(sd, [x], h, [noise], lr=1e-3, opt_state=st)
            dt = time.time() - t0
            if s > 0:
                times.append(dt)
            if sum(times) > budget_s:
                break
        times.sort()
        return times[len(times) // 2], len(times)

    med, n = run(True, 14.0)
    med_pre, n_pre = run(False, 10.0)
    return {"value": B / med, "unit": "cells/s", "cores": cores, "kind": "port",
            "sample": f"{n} full steps of the same workload (A={A}, B={B}, D={D}) after 1 warm-up, median; faithful variant: "
                      f"the dropout masks are drawn with bernoulli_ inside every step, as the reference's nn.Dropout does",
            "ms_per_step": med * 1e3,
            "masks_precomputed": {"value": B / med_pre, "unit": "cells/s", "ms_per_step": med_pre * 1e3,
                                  "sample": f"{n_pre} steps, median; noise drawn once outside the timed region"}}


def P_count(N, A, B, D, H, L, C, S):
    lay = N.param_layout(N.Dims(A, B, D, H, L, C, S))
    return sum(int(lay.rows[t]) * int(lay.cols[t]) for t in range(N.N_PARAM_TENSORS))


def bf16_config(args, model_args, data, batches, nb, A, B, D, H, L, C, S, P, world, rank, dev, timed, DD, FusedAdam, mixVAE_model):
    """BASELINE.json configs[2] (bf16 GEMM operands, data parallel over the node's GPUs): the same K timed steps with
    ``gemm_dtype = "bf16"`` on a fresh model.  With the matrix pipe 16 x faster the step is priced against HBM: SURVEY.md
    section 8(d)'s algorithmic bytes per cell with bf16 operands, A (10 D + 80 H + 36 P / B), times cells/s, over 8 TB/s.
    The configuration runs on bf16 STORAGE: the resident matrix has a bf16 copy (made once, outside
    the timed region, as the matrix's upload is) and the step reads the same cells through it (mmvae_train_step_rows with
    data_bf16; dZ11 travels as bf16); the same step on the fp32 matrix is timed beside it (``fp32_storage``)."""
    from distributed_vae_amd import _native as N
    torch.manual_seed(546)
    m = mixVAE_model(**model_args).to(dev)
    m.train()
    m.gemm_dtype = "bf16"
    opt = FusedAdam(m, lr=1e-3)
    if world > 1:
        DD.broadcast_flat(m.flat_parameters())
    storage = D % 8 == 0 and data.stride(0) % 8 == 0
    data16 = N.to_bf16(data) if storage else None
    rows = [torch.arange(i * B, (i + 1) * B, device=dev) for i in range(nb)]

    def step16(i):
        if world > 1 or args.rehearse_dp:
            return DD.dp_train_step(m, None, 1.0, opt, rehearse=args.rehearse_dp, rows=(data, rows[i % nb], data16))
        return m.fused_train_step_rows(data, rows[i % nb], 1.0, opt, do_adam=True, data16=data16)

    def step32(i):
        xs = batches[i % nb].expand(A, -1, -1)
        if world > 1 or args.rehearse_dp:
            return DD.dp_train_step(m, xs, 1.0, opt, rehearse=args.rehearse_dp)
        return m.fused_train_step(xs, 1.0, opt, do_adam=True)

    dt32, ev32, loss32 = timed(step32)
    dt, ev, loss = timed(step16) if storage else (dt32, ev32, loss32)
    # the same K steps through a FIXED RANDOM permutation of the resident rows (drawn outside the timed region): what a shuffled
    # epoch asks of the 16-byte bf16 row reads; `value` above uses contiguous row sets
    shuffled = None
    if storage:
        gperm = torch.Generator(device=dev).manual_seed(977 + rank)
        perm = torch.randperm(nb * B, generator=gperm, device=dev)
        prows = [perm[i * B:(i + 1) * B].contiguous() for i in range(nb)]

        def step16p(i):
            if world > 1 or args.rehearse_dp:
                return DD.dp_train_step(m, None, 1.0, opt, rehearse=args.rehearse_dp, rows=(data, prows[i % nb], data16))
            return m.fused_train_step_rows(data, prows[i % nb], 1.0, opt, do_adam=True, data16=data16)
        dtp, evp, _ = timed(step16p)
        shuffled = {"ms_per_step": dtp / args.steps * 1e3, "value": world * B * args.steps / dtp, "ms_per_step_hip_events": evp,
                    "rows": "a fixed random permutation of the resident rows (torch.randperm, seeded), B per step"}
    cells = world * B * args.steps / dt
    by_cell = A * (10 * D + 80 * H + 36 * P / B)
    out = {"value": cells, "unit": "cells/s", "ms_per_step": dt / args.steps * 1e3, "ms_per_step_hip_events": ev,
           "dtype": "bf16 operands in the five D x H GEMMs, f32 accumulation and everything else", "n_gpus": world,
           "storage": "bf16 copy of the resident matrix, dZ11 as bf16 (mmvae_train_step_rows with data_bf16)" if storage
                      else "fp32 (gene count or row pitch not a multiple of 8)",
           "last_loss": loss, "bytes_per_cell_algorithmic": by_cell, "shuffled_rows": shuffled,
           "roofline": {"bound": "hbm", "achieved": cells / world * by_cell / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": cells / world * by_cell / 1e9 / PEAK_HBM_GBS,
                        "note": "whole step, per GPU: algorithmic bytes per cell (SURVEY.md 8d, bf16 operands) x cells/s"},
           "fp32_storage": {"value": world * B * args.steps / dt32, "ms_per_step": dt32 / args.steps * 1e3,
                            "ms_per_step_hip_events": ev32, "last_loss": loss32,
                            "note": "the same configuration reading x and dZ11 as fp32 and rounding on load (rounds 1-2)"}}
    if rank == 0 and not args.no_roofline and (A, B, D, H) == (2, 5000, 5000, 100):
        out["fp32_storage"]["stages"] = measure_bf16_stages(m, batches[0], A, B, D, H)
    return out


def other_configs(args, dev, data, B, mixVAE_model, FusedAdam, N):
    """BASELINE.json's other fp32 workloads at per-GPU full size, driver-timed beside the headline: configs[3] (A = 5 arms,
    D = 5000; train-scripts/*.sh fix the arm counts) and configs[4] (A = 3 at the real SmartSeq gene count D = 5032,
    nn_model.py:18, on the synthetic stand-in of SURVEY.md 8d: 22 365 cells).  20 timed steps each after 5 warm-ups, wall
    clock between synchronisations as the headline; `roofline_step` is SURVEY.md 8(d)'s arithmetic for the whole step."""
    H, L, C, S = 100, 10, 92, 2
    out = {}
    steps, warm = 20, 5
    for key, A, D, cells in (("cfg4_A5_D5000", 5, 5000, None), ("cfg5_A3_D5032", 3, 5032, 22365)):
        if D == data.shape[1]:
            d = data
        else:
            d = synthetic_rows((cells // B) * B, D, 546, dev)
        nb = d.shape[0] // B
        torch.manual_seed(546)
        m = mixVAE_model(input_dim=D, fc_dim=H, n_categories=C, state_dim=S, lowD_dim=L, x_drop=0.5, s_drop=0.0, n_arm=A, lam=1,
                         lam_pc=1, tau=0.005, beta=1.0, hard=False, variational=True, device=dev, eps=1e-8, momentum=0.01,
                         ref_prior=False, loss_mode="MSE").to(dev)
        m.train()
        m.gemm_dtype = args.gemm_dtype
        opt = FusedAdam(m, lr=1e-3)
        bs = [d[i * B:(i + 1) * B] for i in range(nb)]
        for i in range(warm):
            m.fused_train_step(bs[i % nb].expand(A, -1, -1), 1.0, opt, do_adam=True)
        torch.cuda.synchronize()
        stream = torch.cuda.current_stream(dev)
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        t0 = time.perf_counter()
        evs[0].record(stream)
        for i in range(steps):
            buf = m.fused_train_step(bs[(warm + i) % nb].expand(A, -1, -1), 1.0, opt, do_adam=True)
            evs[i + 1].record(stream)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        per = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(steps))
        P = P_count(N, A, B, D, H, L, C, S)
        fl_cell, by_cell = A * flops_per_cell_arm(D, H, L, C, S), A * bytes_per_cell_arm(D, H, P, B)
        cps = B * steps / dt
        x3 = (N.gemm_mode(args.gemm_dtype) & 0xFF) == 2
        eng_peak = PEAK_BF16_MFMA_TFLOPS / 6.0 if x3 else PEAK_FP32_MFMA_TFLOPS
        t_m, t_h = fl_cell / (eng_peak * 1e12), by_cell / (PEAK_HBM_GBS * 1e9)
        out[key] = {"arms": A, "genes": D, "batch": B, "cells_resident": int(d.shape[0]), "steps": steps, "warmup": warm,
                    "ms_per_step": dt / steps * 1e3, "ms_per_step_hip_events_median": per[len(per) // 2],
                    "value": cps, "unit": "cells/s", "last_loss": float(buf[0]),
                    "flop_per_cell": fl_cell, "bytes_per_cell": by_cell,
                    "roofline_step": {"bound": "hbm" if t_h >= t_m else "mfma", "frac": cps * max(t_h, t_m),
                                      "bytes_frac_of_hbm_peak": cps * t_h, "flops_frac_of_engine_peak": cps * t_m}}
        del m, opt, bs
        if d is not data:
            del d
        torch.cuda.empty_cache()
    return out


def measure_bf16_stages(model, x, A, B, D, H):
    """Per-launch duration of the five bf16 GEMM kernels (HIP events on the launch stream, mmvae_debug_stage replays) and
    the HBM traffic they are priced by: fp32 bytes actually streamed per launch (x, dZ11, slabs) / duration."""
    from distributed_vae_amd import _native as N
    eng = model._engine
    hyper = model._hyper(1.0, False)
    noise = N.make_noise(None, 99, 1)
    eng.forward(hyper, noise, model._flat, model._bn_flat, None, x, 0, None, True)
    eng.loss(hyper)
    eng.backward(hyper, noise, model._flat, x, 0, model._flat_grad)
    stream = torch.cuda.current_stream()
    sp = eng.splits()
    # stage id: (kernel, bytes streamed per launch: inputs read once + outputs written once, x shared by the arms)
    stages = {
        14: ("k_bf16_gemm<fc1>", 4.0 * B * D + A * 4.0 * H * D + sp[0] * A * 4.0 * B * 128),
        1: ("k_bf16_fc11g (fc11 + loss + dZ11 + d(d10))", 4.0 * B * D + A * 4.0 * B * D + A * 4.0 * D * H + sp[4] * A * 4.0 * B * H),
        12: ("k_bf16_gemm<dW1>", 4.0 * B * D + A * 4.0 * B * H + sp[2] * A * 4.0 * H * D),
        13: ("k_bf16_gemm<dW11>", A * 4.0 * B * D + A * 4.0 * B * H + sp[5] * A * 4.0 * D * 132),
    }
    res = {}
    for sid, (name, by) in stages.items():
        for _ in range(3):
            eng.debug_stage(sid, hyper, noise, model._flat, x, 0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(20):
            eng.debug_stage(sid, hyper, noise, model._flat, x, 0)
        e1.record(stream)
        e1.synchronize()
        ms = e0.elapsed_time(e1) / 20
        res[name] = {"avg_launch_ms": ms, "streamed_GBs": by / ms / 1e6, "frac_of_hbm_peak": by / ms / 1e6 / PEAK_HBM_GBS,
                     "tflops": A * (4.0 if sid == 1 else 2.0) * B * D * H / ms / 1e9}
    return res


def _launch_mod():
    """distributed-vae_amd/launch.py by path: no torch, no HIP -- the parent of the ranks must not have initialised the GPU."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("mmvae_launch", os.path.join(ROOT, "distributed-vae_amd", "launch.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def spawn_ranks(n: int, share_gpu: bool = False, log_dir: str = "") -> int:
    """--gpus N without a launcher: start N rank processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their
    environment) and supervise them (launch.run_ranks: the first rank that exits non-zero stops the others and its stderr
    tail is printed -- a rank that cannot open its device must not leave the rest waiting in their first collective).
    This process has not initialised the GPU (counting devices does not), and it never replaces itself: the ranks are
    children.  Returns the exit status to leave with."""
    import socket
    have = torch.cuda.device_count()
    if have < n and not (share_gpu and have >= 1):
        print(f"bench.py: --gpus {n} but only {have} GPU(s) visible: refusing to run the {n}-GPU job on fewer devices",
              file=sys.stderr, flush=True)
        return 2
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    return _launch_mod().run_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], n, port, log_dir or None)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--arms", type=int, default=2)
    ap.add_argument("--batch", type=int, default=5000)
    ap.add_argument("--genes", type=int, default=5000)
    ap.add_argument("--cells", type=int, default=50000, help="cells resident per GPU (batches cycle through them)")
    ap.add_argument("--cpu-steps", type=int, default=5)
    ap.add_argument("--cpu-threads", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-bf16", action="store_true", help="skip the bf16-operand configuration measured beside the headline")
    ap.add_argument("--bf16-fp32-storage", action="store_true",
                    help="--gemm-dtype bf16 only: read x / dZ11 as fp32 and round on load instead of the bf16 copy")
    ap.add_argument("--no-eval", action="store_true", help="skip the evaluation-label / consensus measurement")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the A = 5 / D = 5000 and A = 3 / D = 5032 lines (BASELINE.json configs[3], configs[4]) measured beside the headline")
    ap.add_argument("--gemm-dtype", choices=["fp32", "fp32x3", "fp32_mfma", "bf16"], default="fp32",
                    help="operand type of the five D x H GEMMs: fp32 (the headline / parity configuration; the library's "
                         "fp32 engine, i.e. fp32x3: every fp32 operand as three exact bf16 slices, six slice products per "
                         "product on the bf16 matrix pipe, fp32 accumulation), fp32_mfma (the fp32 matrix instruction), or "
                         "bf16 (BASELINE.json configs[2]: operands ROUNDED to bf16, fp32 accumulation, everything else fp32)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal of the N > 1 path on a box with fewer GPUs: every rank uses cuda:0 and the collectives "
                         "go through gloo (RCCL refuses two ranks on one device); the line is marked as a rehearsal")
    ap.add_argument("--rehearse-dp", action="store_true",
                    help="one rank, but through the data-parallel step (RCCL init, all-reduce(AVG) of the flat gradients, "
                         "separate Adam launch): exercises the N > 1 code path on a one-GPU box")
    ap.add_argument("--log-dir", default="", help="N > 1 without a launcher: directory for the ranks' rank<r>.err files")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus, args.share_gpu, args.log_dir))   # no launcher: be one (before any GPU call in this process)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if os.environ.get("MMVAE_BENCH_FAIL_RANK") == str(rank):   # test hook: this rank dies right after start
        print(f"rank {rank}: MMVAE_BENCH_FAIL_RANK set, exiting 1", file=sys.stderr, flush=True)
        sys.exit(1)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.share_gpu:
        local_rank = 0
    elif local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: LOCAL_RANK={local_rank} but only {torch.cuda.device_count()} GPU(s) visible")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    rccl_ranks = None
    backend = None
    if world > 1 or args.rehearse_dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # RCCL prints a version banner on stdout when the communicator is created: keep stdout for the one JSON line
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            backend = "gloo" if args.share_gpu else "nccl"
            import datetime
            # a rank that never arrives must not hold the others for the library default of ten minutes
            tmo = datetime.timedelta(seconds=int(os.environ.get("MMVAE_INIT_TIMEOUT_S", "120")))
            if args.share_gpu:
                dist.init_process_group("gloo", rank=rank, world_size=world, timeout=tmo)
            else:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=tmo)
            warm = torch.ones(8, device=dev)
            dist.all_reduce(warm)                      # a real collective: every rank contributes 1
            torch.cuda.synchronize()
            rccl_ranks = int(round(float(warm[0])))
            if rccl_ranks != dist.get_world_size() or rccl_ranks != world:
                raise SystemExit(f"{backend} all-reduce saw {rccl_ranks} ranks, expected {world}")
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)

    import distributed_vae_amd  # noqa: F401
    from distributed_vae_amd import _native as N
    from distributed_vae_amd import dist as DD
    from distributed_vae_amd.cpl_mixvae import FusedAdam
    from distributed_vae_amd.nn_model import mixVAE_model

    A, B, D = args.arms, args.batch, args.genes
    H, L, C, S = 100, 10, 92, 2
    torch.manual_seed(546)
    model_args = dict(input_dim=D, fc_dim=H, n_categories=C, state_dim=S, lowD_dim=L, x_drop=0.5, s_drop=0.0,
                      n_arm=A, lam=1, lam_pc=1, tau=0.005, beta=1.0, hard=False, variational=True, device=dev,
                      eps=1e-8, momentum=0.01, ref_prior=False, loss_mode="MSE")
    model = mixVAE_model(**model_args).to(dev)
    model.train()
    model.gemm_dtype = args.gemm_dtype
    opt = FusedAdam(model, lr=1e-3)
    if world > 1:
        DD.broadcast_flat(model.flat_parameters())
    # this rank's shard of cells, resident in HBM
    nb = max(1, args.cells // B)
    data = synthetic_rows(nb * B, D, 546 + rank, dev)
    batches = [data[i * B:(i + 1) * B] for i in range(nb)]

    # --gemm-dtype bf16: the configuration runs on bf16 storage (see bf16_config) unless --bf16-fp32-storage
    data16 = None
    if args.gemm_dtype == "bf16" and not args.bf16_fp32_storage and D % 8 == 0 and data.stride(0) % 8 == 0:
        data16 = N.to_bf16(data)
        row_sets = [torch.arange(i * B, (i + 1) * B, device=dev) for i in range(nb)]

    def step(i):
        if data16 is not None:
            if world > 1 or args.rehearse_dp:
                return DD.dp_train_step(model, None, 1.0, opt, rehearse=args.rehearse_dp, rows=(data, row_sets[i % nb], data16))
            return model.fused_train_step_rows(data, row_sets[i % nb], 1.0, opt, do_adam=True, data16=data16)
        xs = batches[i % nb].expand(A, -1, -1)
        if world > 1 or args.rehearse_dp:
            return DD.dp_train_step(model, xs, 1.0, opt, rehearse=args.rehearse_dp)
        return model.fused_train_step(xs, 1.0, opt, do_adam=True)

    def timed(step_fn):
        """W untimed steps, then exactly K steps between barrier + synchronize on both sides; max over ranks."""
        for i in range(args.warmup):
            step_fn(i)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        # per-step HIP events on the stream the step is launched on (torch's current stream): the median step period
        # beside the wall-clock mean that `value` is computed from
        stream = torch.cuda.current_stream(dev)
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
        t0 = time.perf_counter()
        evs[0].record(stream)
        for i in range(args.steps):
            buf_ = step_fn(args.warmup + i)
            evs[i + 1].record(stream)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt_ = time.perf_counter() - t0
        tmax = torch.tensor([dt_], device=dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        per = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(args.steps))
        stats = {"median": per[len(per) // 2], "min": per[0], "max": per[-1], "p10": per[len(per) // 10],
                 "p90": per[(9 * len(per)) // 10], "n": len(per)}
        return float(tmax), stats, float(buf_[0])

    dt, ev_stats, loss_last = timed(step)

    P = 0
    lay = N.param_layout(N.Dims(A, B, D, H, L, C, S))
    for t in range(N.N_PARAM_TENSORS):
        P += int(lay.rows[t]) * int(lay.cols[t])
    fl_cell = A * flops_per_cell_arm(D, H, L, C, S)
    by_cell = A * bytes_per_cell_arm(D, H, P, B)
    cells_per_s = world * B * args.steps / dt

    roof = None
    if rank == 0 and not args.no_roofline:
        try:
            roof = measure_stages(model, batches[0], A, B, D, H)
        except Exception as e:   # noqa: BLE001
            roof = {"error": f"{type(e).__name__}: {e}"}
        # the whole step by SURVEY.md section 8(d)'s arithmetic, with the engine in use: ns per cell each roof allows
        eng_peak = roof.get("mfma", {}).get("engine_peak", PEAK_FP32_MFMA_TFLOPS)
        t_m, t_h = fl_cell / (eng_peak * 1e12), by_cell / (PEAK_HBM_GBS * 1e9)
        per_gpu = cells_per_s / world
        roof["step"] = {"bound": "hbm" if t_h >= t_m else "mfma", "ns_per_cell_hbm_roof": t_h * 1e9,
                        "ns_per_cell_mfma_roof": t_m * 1e9, "ns_per_cell_measured": 1e9 / per_gpu,
                        "frac": per_gpu * max(t_h, t_m),
                        "bytes_frac_of_hbm_peak": per_gpu * t_h, "flops_frac_of_engine_peak": per_gpu * t_m}
        roof["step_flops_frac_of_fp32_mfma_peak"] = per_gpu * fl_cell / (PEAK_FP32_MFMA_TFLOPS * 1e12)
        roof["step_bytes_frac_of_hbm_peak"] = per_gpu * by_cell / (PEAK_HBM_GBS * 1e9)

    out = {
        "metric": "cells/sec per train step (fwd+loss+bwd+opt)",
        "value": cells_per_s,
        "unit": "cells/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "ms_per_step_hip_events": ev_stats,
        "value_at_median_step": world * B / ev_stats["median"] * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        **({"rehearsal": "ranks share one GPU, gloo collectives: NOT a multi-GPU measurement"} if args.share_gpu else {}),
        "vs_baseline": None,
        "dtype": "f32" if args.gemm_dtype != "bf16" else "bf16 operands in the five D x H GEMMs, f32 accumulation and everything else",
        "data": "synthetic",
        "config": {"workload": f"cpl_mixVAE A={A} arms, synthetic-10x-v1 {args.cells} cells x {D} genes per GPU, "
                               f"batch {B}/GPU, fp32, H=100 L=10 C=92 S=2, x_drop=0.5, Adam lr=1e-3",
                   "global_batch": world * B, "parallelism": f"dp{world}", "noise": "in-kernel Philox4x32-10",
                   "gemm_engine": engine_name(N, args.gemm_dtype),
                   "flop_per_cell": fl_cell, "bytes_per_cell": by_cell, "last_loss": loss_last},
    }
    if rccl_ranks is not None:
        # ranks that contributed to a real all-reduce, and through which backend ("nccl" is RCCL on ROCm; "gloo" only in
        # the --share-gpu rehearsal, which is not an RCCL observation)
        out["collective_ranks"] = rccl_ranks
        out["collective_backend"] = "rccl (torch.distributed nccl)" if backend == "nccl" else backend
    if args.gemm_dtype != "bf16" and not args.no_bf16:
        # BASELINE.json configs[2] beside the headline: the same step with bf16 operands in the five D x H GEMMs
        try:
            out["bf16_config"] = bf16_config(args, model_args, data, batches, nb, A, B, D, H, L, C, S, P_count(N, A, B, D, H, L, C, S),
                                             world, rank, dev, timed, DD, FusedAdam, mixVAE_model)
        except Exception as e:   # noqa: BLE001
            out["bf16_config"] = {"error": f"{type(e).__name__}: {e}"}
    if (rank == 0 and world == 1 and not args.no_other_configs and args.gemm_dtype != "bf16"
            and (A, B, D) == (2, 5000, 5000)):
        try:
            out["other_configs"] = other_configs(args, dev, data, B, mixVAE_model, FusedAdam, N)
        except Exception as e:   # noqa: BLE001
            out["other_configs"] = {"error": f"{type(e).__name__}: {e}"}
    if roof is not None:
        out["roofline"] = roof
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(args, D, H, L, C, S, A, B)
        except Exception as e:   # noqa: BLE001
            out["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0 and world == 1 and not args.no_eval:
        # the widened rows beside the headline metric; a failure here must not cost the JSON line
        for key, fn in (("eval_consensus", lambda: eval_consensus(args, model, batches, A, B, D, H, L, C, S, not args.no_cpu_baseline)),
                        ("augmenter", lambda: augmenter_forward(args, batches, A, B, D, not args.no_cpu_baseline)),
                        ("data_path", lambda: data_path(data, A, B, D))):
            try:
                out[key] = fn()
            except Exception as e:   # noqa: BLE001
                out[key] = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1 or args.rehearse_dp:
        # rank 0 measured its stages after the timed region: the others wait for it, so that every rank tears the
        # communicator down together
        dist.barrier()
        torch.cuda.synchronize()
        dist.destroy_process_group()


def eval_consensus(args, model, batches, A, B, D, H, L, C, S, with_cpu):
    """Scope row (f)-1 beside the headline metric: one pass of the per-epoch consensus loop (cpl_mixvae.py:563-657)
    over the resident cells -- eval-mode encoder + latent block, argmax labels, per-pair confusion counts -- then the
    normalisation and mean; HIP events on torch's current stream (the one the C ABI is called on)."""
    from distributed_vae_amd._utils import confmat_counts, consensus_from_counts
    model.eval()
    counts = confmat_counts(A, C, batches[0].device)
    for b in batches[:2]:
        model.eval_labels(b.expand(A, -1, -1), 1.0, counts)
    counts.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    e0.record()
    for _ in range(reps):
        counts.zero_()
        for b in batches:
            model.eval_labels(b.expand(A, -1, -1), 1.0, counts)
        cons = consensus_from_counts(counts)
    e1.record()
    e1.synchronize()
    ms = e0.elapsed_time(e1) / reps
    n = len(batches) * B
    model.train()
    out = {"value": n / ms * 1e3, "unit": "cells/s", "ms_per_pass": ms, "cells": n, "batch": B,
           "consensus": float(cons.mean()),
           "workload": "eval forward (encoder + latent block) + argmax labels + confusion counts per batch, "
                       "normalise + mean once per pass",
           "flop_per_cell": A * 2.0 * (D * H + 3 * H * H + H * L + L * C)}
    out["frac_of_fp32_mfma_peak"] = out["value"] * out["flop_per_cell"] / (PEAK_FP32_MFMA_TFLOPS * 1e12)
    if with_cpu:
        # the oracle's eval forward + numpy consensus on ONE batch (bounded), same host threads as cpu_baseline
        from oracle import consensus as OC
        from oracle import restatement as R
        h = R.Hyper(input_dim=D, fc_dim=H, n_categories=C, state_dim=S, lowD_dim=L, n_arm=A)
        sd = R.init_state_dict(h, 546)
        x = R.synthetic_batch(B, D)
        noise = R.draw_noise(h, B, seed=1)
        ts = []
        for _ in range(3):
            t0 = time.time()
            with torch.no_grad():
                o = R.forward(sd, [x] * A, h, noise, training=False, eval_flag=True, update_running=False)
            lab = np.stack([OC.classify(c.numpy()) for c in o[4]]).astype(np.int64)
            OC.epoch_consensus(lab, C)
            ts.append(time.time() - t0)
        ts.sort()
        out["cpu_baseline"] = {"value": B / ts[1], "unit": "cells/s", "cores": torch.get_num_threads(), "kind": "port",
                               "sample": "3 batches of the same shape through the oracle's full eval forward (as the "
                                         "reference does) + numpy consensus, median"}
    return out


def data_path(data, A, B, D):
    """Scope row (f)-3: assembling a shuffled batch from the HBM-resident matrix (mmvae_gather_rows) -- what replaces
    DataLoader workers + pinned memory + the H2D copy of the reference (utils/dataloader.py:114-132)."""
    from distributed_vae_amd import _native as N
    idx = torch.randperm(data.shape[0], device=data.device)[:B]
    out = torch.empty(B, D, device=data.device)
    for _ in range(3):
        N.gather_rows(data, idx, out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        N.gather_rows(data, idx, out)
    e1.record()
    e1.synchronize()
    ms = e0.elapsed_time(e1) / reps
    by = 2.0 * B * D * 4
    res = {"kernel": "k_gather_rows", "us_per_batch": ms * 1e3, "algorithmic_GBs": by / ms / 1e6,
           "frac_of_hbm_peak": by / ms / 1e6 / PEAK_HBM_GBS, "bytes_per_batch": by,
           "note": "a host-resident batch would cost B*D*4 bytes over PCIe (~1.6 ms at 63 GB/s) per step instead"}
    # a shuffled epoch from the device-resident loader through the trainer: row-indexed steps (mmvae_train_step_rows: the
    # batch is never assembled; the default), and for comparison gathered batches (MMVAE_ROWS=0): the gather of batch i+1 on
    # a side stream beside step i, and gather-then-step
    from distributed_vae_amd.cpl_mixvae import cpl_mixVAE
    from distributed_vae_amd.utils.dataloader import DeviceLoader
    tr = cpl_mixVAE(saving_folder="", device=data.device, save_flag=False)
    tr.init_model(n_categories=92, state_dim=2, input_dim=D, fc_dim=100, lowD_dim=10, x_drop=0.5, s_drop=0.0, n_arm=A)
    ld = DeviceLoader(data, torch.arange(data.shape[0]), B, True, True, seed=546)
    prev = os.environ.get("MMVAE_ROWS")
    try:
        for key, rows, pipe in (("shuffled_epoch_ms_per_step", "1", True),
                                ("shuffled_epoch_ms_per_step_gathered_pipelined", "0", True),
                                ("shuffled_epoch_ms_per_step_gathered_back_to_back", "0", False)):
            os.environ["MMVAE_ROWS"] = rows
            tr.pipeline = pipe
            for _ in tr.epoch_steps(ld):
                pass
            e0.record()
            n = 0
            for _ in range(3):
                for _b in tr.epoch_steps(ld):
                    n += 1
            e1.record()
            e1.synchronize()
            res[key] = e0.elapsed_time(e1) / n
        res["row_indexed_steps"] = bool(getattr(tr, "_rows_ok", True))
        # the same shuffled epoch on the production path (augmenter in front of every step, pipelined): the augmenter's first
        # layer reads the epoch's rows out of the loader's slice planes (made once per data set; Augmenter_smartseq.forward_rows),
        # against gathered batches converted per batch (MMVAE_ROWS=0)
        from distributed_vae_amd.augmentation import Augmenter_smartseq
        torch.manual_seed(546)
        tr.set_augmenter(Augmenter_smartseq(50, 10, D, 500).to(data.device).eval())
        tr.pipeline = True
        for key, rows in (("augmented_shuffled_epoch_ms_per_step", "1"), ("augmented_shuffled_epoch_ms_per_step_gathered", "0")):
            os.environ["MMVAE_ROWS"] = rows
            for _ in tr.epoch_steps(ld):
                pass
            if rows == "1":
                res["augmenter_reads_rows"] = bool(getattr(tr, "used_aug_rows", False))
            e0.record()
            n = 0
            for _ in range(3):
                for _b in tr.epoch_steps(ld):
                    n += 1
            e1.record()
            e1.synchronize()
            res[key] = e0.elapsed_time(e1) / n
        tr.set_augmenter(None)
    finally:
        if prev is None:
            os.environ.pop("MMVAE_ROWS", None)
        else:
            os.environ["MMVAE_ROWS"] = prev
    return res


def augmenter_forward(args, batches, A, B, D, with_cpu):
    """Scope row (f)-2 beside the headline metric: the eval-mode augmenter forward the reference trainer runs in front
    of every step (cpl_mixvae.py:422-423), random-init weights of the Augmenter_smartseq architecture (n_dim 500,
    noise 50, latent 10; the pretrained file is not shipped), x shared by the arms; HIP events on the call stream."""
    from distributed_vae_amd.augmentation import Augmenter_smartseq
    ND, NZ, Z = 500, 50, 10
    torch.manual_seed(546)
    net = Augmenter_smartseq(NZ, Z, D, ND).to(batches[0].device).eval()
    for b in batches[:2]:
        net(b.expand(A, -1, -1), True, 0.1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 2
    e0.record()
    for _ in range(reps):
        for b in batches:
            net(b.expand(A, -1, -1), True, 0.1)
    e1.record()
    e1.synchronize()
    ms = e0.elapsed_time(e1) / (reps * len(batches))
    n1, n5 = D // 5, ND // 5
    trunk = D * n1 + n1 * n1 + n1 * ND + ND * ND + ND * n5
    tail = n5 * ND + ND * ND + ND * n1 + n1 * n1 + n1 * D
    fl_exec = 2.0 * B * (trunk + A * tail)          # the trunk once per cell (the arms share x)
    fl_ref = 2.0 * A * B * (trunk + tail)           # the reference runs it once per cell-arm
    out = {"value": B / ms * 1e3, "unit": "cells/s", "ms_per_batch": ms, "batch": B, "arms": A,
           "workload": f"Augmenter_smartseq eval forward, D={D}, n_dim={ND}, noise {NZ}, latent {Z}, x shared by {A} arms",
           "gflop_executed": fl_exec / 1e9, "gflop_reference_pattern": fl_ref / 1e9,
           "tflops": fl_exec / ms / 1e9, "frac_of_fp32_mfma_peak": fl_exec / ms / 1e9 / PEAK_FP32_MFMA_TFLOPS,
           # the fp32x3 engine's ceiling: 2.5 PFLOP/s of bf16 MFMAs / 6 slice products per product
           "engine": "fp32x3: planes x planes GEMM (csrc/gemm_pp.hip), every layer's epilogue writes the next layer's slice planes",
           "frac_of_engine_peak": fl_exec / ms / 1e9 / (PEAK_BF16_MFMA_TFLOPS / 6.0)}
    # the production loop with augmentation: augmenter of batch i+1 on a side stream beside train step i
    from distributed_vae_amd.cpl_mixvae import cpl_mixVAE
    tr = cpl_mixVAE(saving_folder="", device=batches[0].device, save_flag=False)
    tr.init_model(n_categories=92, state_dim=2, input_dim=D, fc_dim=100, lowD_dim=10, x_drop=0.5, s_drop=0.0, n_arm=A)
    tr.set_augmenter(net)
    for pipe in (False, True):
        tr.pipeline = pipe
        for _ in tr.epoch_steps(batches[:3]):
            pass
        e0.record()
        n = 0
        for _ in range(2):
            for _b in tr.epoch_steps(batches):
                n += 1
        e1.record()
        e1.synchronize()
        out["augmented_step_ms_pipelined" if pipe else "augmented_step_ms_back_to_back"] = e0.elapsed_time(e1) / n
    out["augmented_cells_per_s"] = B / out["augmented_step_ms_pipelined"] * 1e3
    # the bf16 configuration (BASELINE.json configs[2]) covers the augmenter's ten large Linear layers too
    try:
        net.gemm_dtype = "bf16"
        for b in batches[:2]:
            net(b.expand(A, -1, -1), True, 0.1)
        e0.record()
        for _ in range(reps):
            for b in batches:
                net(b.expand(A, -1, -1), True, 0.1)
        e1.record()
        e1.synchronize()
        ms16 = e0.elapsed_time(e1) / (reps * len(batches))
        tr.init_model(n_categories=92, state_dim=2, input_dim=D, fc_dim=100, lowD_dim=10, x_drop=0.5, s_drop=0.0, n_arm=A,
                      gemm_dtype="bf16")
        tr.set_augmenter(net)
        tr.pipeline = True
        for _ in tr.epoch_steps(batches[:3]):
            pass
        e0.record()
        n = 0
        for _ in range(2):
            for _b in tr.epoch_steps(batches):
                n += 1
        e1.record()
        e1.synchronize()
        out["bf16_config"] = {"ms_per_batch": ms16, "tflops": fl_exec / ms16 / 1e9,
                              "frac_of_engine_peak": fl_exec / ms16 / 1e9 / PEAK_BF16_MFMA_TFLOPS,
                              "augmented_step_ms_pipelined": e0.elapsed_time(e1) / n,
                              "augmented_cells_per_s": B / (e0.elapsed_time(e1) / n) * 1e3}
    except Exception as e:   # noqa: BLE001
        out["bf16_config"] = {"error": f"{type(e).__name__}: {e}"}
    finally:
        net.gemm_dtype = "fp32"
    if with_cpu:
        from oracle import augmenter as OA
        sd = OA.random_state_dict(NZ, Z, D, ND, seed=1)
        nb = 500                                       # bounded sample: 500 cells x A arms
        x = torch.rand(nb, D)
        z0, eps = torch.randn(A, nb, NZ), torch.randn(A, nb, Z)
        ts = []
        for _ in range(3):
            t0 = time.time()
            with torch.no_grad():
                OA.forward_eval(sd, x.expand(A, -1, -1), z0, eps, 0.1)
            ts.append(time.time() - t0)
        ts.sort()
        out["cpu_baseline"] = {"value": nb / ts[1], "unit": "cells/s", "cores": torch.get_num_threads(), "kind": "port",
                               "sample": f"{nb} cells x {A} arms through the oracle (torch CPU, reference call pattern), "
                                         f"median of 3"}
    return out


def engine_name(N, gemm_dtype):
    """The GEMM engine behind a gemm_dtype, with what it does to an fp32 product (for the JSON line)."""
    mode = N.gemm_mode(gemm_dtype) & 0xFF
    return {0: "fp32_mfma: v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain)",
            1: "bf16: operands rounded to bf16, v_mfma_f32_32x32x16_bf16, fp32 accumulation",
            2: "fp32x3: each fp32 operand split exactly into three bf16 slices, six slice products per product on "
               "v_mfma_f32_32x32x16_bf16, fp32 accumulation (dropped terms <= 2^-26 per product)"}[mode]


# kernel names of the fp32x3 engine for the same debug stages (gemm_bf16.hip)
STAGES_X3 = {14: "k_x3_gemm<fc1>", 1: "k_x3_fc11g", 12: "k_x3_gemm<dW1>", 13: "k_x3_gemm<dW11>"}

STAGES = {
    # debug-stage id: (kernel, algorithmic FLOPs per launch, algorithmic HBM bytes per launch, on the critical path?)
    # as functions of (A,B,D,H).  dW11 runs on a side stream beside the latency-bound backward chain (with fewer,
    # longer workgroups on purpose), so it is timed but never chosen as the roofline kernel.
    14: ("k_fc1_fwd_v3|k_fc1_fwd_v2", lambda A, B, D, H: A * 2.0 * B * D * H, lambda A, B, D, H: A * 4.0 * B * D, True),
    # fc11 forward + loss + dZ11 + d(d10): one fused kernel at fc_dim 100 (2 GEMMs' worth of FLOPs), two launches otherwise
    1: ("k_fc11_zg|k_fc11_zt+k_gd10_v2", lambda A, B, D, H: A * 4.0 * B * D * H, lambda A, B, D, H: A * 2 * 4.0 * B * D, True),
    12: ("k_tn_v3m<dW1>|k_tn_v2<dW1>", lambda A, B, D, H: A * 2.0 * B * D * H, lambda A, B, D, H: A * 4.0 * B * D, True),
    13: ("k_tn_v3n<dW11>|k_tn_v2<dW11>", lambda A, B, D, H: A * 2.0 * B * D * H, lambda A, B, D, H: A * 4.0 * B * D, False),
}


def measure_stages(model, x, A, B, D, H):
    """Per-launch duration of the five MFMA-bound kernels (the five D x H GEMMs of SURVEY.md 8d), measured
    live with HIP events on the stream the kernels are launched on (torch's current stream), each kernel
    replayed alone on a workspace that a full forward+loss+backward has prepared (mmvae_debug_stage).
    The one with the longest launch on the step's critical path is reported as the roofline object; FLOPs are algorithmic
    (2*B*D*H per arm and GEMM; padding H -> 104/128 is not counted)."""
    from distributed_vae_amd import _native as N

    eng = model._engine
    hyper = model._hyper(1.0, False)
    x3 = (hyper.gemm_bf16 & 0xFF) == 2 and H + 1 <= 112
    noise = N.make_noise(None, 99, 1)
    eng.forward(hyper, noise, model._flat, model._bn_flat, None, x, 0, None, True)
    eng.loss(hyper)
    eng.backward(hyper, noise, model._flat, x, 0, model._flat_grad)
    stream = torch.cuda.current_stream()
    res, sid_of = {}, {}
    for sid, (names, fl, by, crit) in STAGES.items():
        # fc_dim 100 runs the 96 + 4 column kernels (v3), any other width the 128-wide tiles (v2)
        name = names.split("|")[0 if (H == 100 or "|" not in names) else 1]
        if x3:
            name = STAGES_X3[sid]
        sid_of[name] = sid
        for _ in range(3):
            eng.debug_stage(sid, hyper, noise, model._flat, x, 0)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        ev0.record(stream)
        for _ in range(reps):
            eng.debug_stage(sid, hyper, noise, model._flat, x, 0)
        ev1.record(stream)
        ev1.synchronize()
        ms = ev0.elapsed_time(ev1) / reps
        res[name] = {"avg_launch_ms": ms, "tflops": fl(A, B, D, H) / (ms * 1e-3) / 1e12,
                     "algorithmic_GBs": by(A, B, D, H) / (ms * 1e-3) / 1e9, "critical_path": crit}
    dom = max((k for k in res if res[k]["critical_path"]), key=lambda k: res[k]["avg_launch_ms"])
    dom_sid = sid_of[dom]
    t_s = res[dom]["avg_launch_ms"] * 1e-3
    fl_launch, by_launch = STAGES[dom_sid][1](A, B, D, H), STAGES[dom_sid][2](A, B, D, H)
    bf16_engine = (hyper.gemm_bf16 & 0xFF) == 1
    # The ceiling of the matrix pipe FOR THE ENGINE IN USE: the fp32 matrix instruction runs at 157.3 TFLOP/s; the fp32x3
    # engine forms an fp32 product from six bf16 slice products on the bf16 pipe (2.5 PFLOP/s dense / 6 = 416.7 TFLOP/s of
    # fp32-equivalent work); bf16 operands run at the bf16 peak.  The kernel's binding roof is the one that takes longer
    # for its ALGORITHMIC work (per launch A 4 B D H FLOP and A 8 B D bytes for the fused fc11 kernel).
    engine_peak = PEAK_BF16_MFMA_TFLOPS if bf16_engine else (PEAK_BF16_MFMA_TFLOPS / 6.0 if x3 else PEAK_FP32_MFMA_TFLOPS)
    t_mfma, t_hbm = fl_launch / (engine_peak * 1e12), by_launch / (PEAK_HBM_GBS * 1e9)
    ach_tf, ach_gbs = fl_launch / t_s / 1e12, by_launch / t_s / 1e9
    bound = "hbm" if t_hbm >= t_mfma else "mfma"
    traffic, src = pmc_traffic(dom, A, B, D, H)
    out = {"bound": bound, "kernel": dom,
           "achieved": ach_gbs if bound == "hbm" else ach_tf,
           "peak": PEAK_HBM_GBS if bound == "hbm" else engine_peak,
           "unit": "GB/s" if bound == "hbm" else "TFLOP/s",
           "frac": ach_gbs / PEAK_HBM_GBS if bound == "hbm" else ach_tf / engine_peak,
           "traffic": traffic, "traffic_unit": "bytes/launch", "traffic_source": src,
           "avg_launch_ms": res[dom]["avg_launch_ms"],
           "algorithmic_flop_per_launch": fl_launch, "algorithmic_bytes_per_launch": by_launch,
           "roof_times_us": {"mfma_engine": t_mfma * 1e6, "hbm": t_hbm * 1e6},
           "hbm": {"achieved": ach_gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": ach_gbs / PEAK_HBM_GBS},
           "mfma": {"achieved": ach_tf, "unit": "TFLOP/s (algorithmic fp32 FLOPs)", "engine_peak": engine_peak,
                    "frac_of_engine_peak": ach_tf / engine_peak, "peak_fp32_mfma": PEAK_FP32_MFMA_TFLOPS,
                    # continuity with rounds 1-2 (which priced every engine against the fp32 matrix instruction's peak;
                    # under fp32x3 that ratio can exceed 1 and is not a distance to any roof)
                    "frac_of_fp32_mfma_peak": ach_tf / PEAK_FP32_MFMA_TFLOPS},
           "stages": res}
    for k, r in res.items():
        r["frac_of_engine_peak"] = r["tflops"] / engine_peak
        r["frac_of_hbm_peak"] = r["algorithmic_GBs"] / PEAK_HBM_GBS
    if x3:
        # what the kernel executes: six bf16 MFMAs per pair of fragments on padded tiles (k_x3_fc11g: 90
        # v_mfma_f32_32x32x16_bf16 per wave and piece of 32 cells x 32 genes); against the bf16 matrix peak:
        ex_fl = A * (4 * ((B + 127) // 128)) * (2 * ((D + 63) // 64)) * 90 * 32768.0 if dom == "k_x3_fc11g" else None
        out["engine"] = "fp32x3 (fp32 operands as three exact bf16 slices, six slice products per product)"
        if ex_fl:
            out["executed"] = {"mfma_tflops": ex_fl / t_s / 1e12, "peak_bf16_tflops": PEAK_BF16_MFMA_TFLOPS,
                               "frac_of_bf16_mfma_peak": ex_fl / t_s / 1e12 / PEAK_BF16_MFMA_TFLOPS,
                               "executed_over_algorithmic_flops": ex_fl / STAGES[1][1](A, B, D, H)}
    busy = pmc_mfma_busy(dom, A, B, D, H)
    if busy is not None:
        out["mfma_busy"] = busy
    return out


PMC_ROUND = "r04"


def _pmc_rows(kind, A, B, D, H):
    """Rows of the committed PMC summary of this round (profiles/<round>_pmc_<kind>_summary.csv), or None when it was
    not collected on this shape or -- checked through the hash the collection recorded in profiles/<round>_pmc_meta.json
    -- with different kernel sources than the ones being run (counters cannot be read from inside the process, so a
    stale file would otherwise be reported as if it described the current kernels)."""
    import csv
    import hashlib
    root = os.path.dirname(os.path.abspath(__file__))
    path = os.path.join(root, "profiles", f"{PMC_ROUND}_pmc_{kind}_summary.csv")
    meta = os.path.join(root, "profiles", f"{PMC_ROUND}_pmc_meta.json")
    if (A, B, D, H) != (2, 5000, 5000, 100) or not os.path.exists(path) or not os.path.exists(meta):
        return None, None
    want = json.load(open(meta)).get("sources_sha256", {})
    for rel, sha in want.items():
        f = os.path.join(root, rel)
        if not os.path.exists(f) or hashlib.sha256(open(f, "rb").read()).hexdigest() != sha:
            return None, None
    return list(csv.DictReader(open(path))), f"profiles/{PMC_ROUND}_pmc_{kind}_summary.csv"


def pmc_mfma_busy(kernel, A, B, D, H):
    """Matrix-pipe utilisation of `kernel` from the committed PMC pass: SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x
    SQ_BUSY_CU_CYCLES).  Only for the shape and the kernel sources the pass was collected on."""
    rows, src = _pmc_rows("mfma", A, B, D, H)
    for r in rows or []:
        if r["kernel"].startswith(kernel) and r.get("SQ_VALU_MFMA_BUSY_CYCLES") and r.get("SQ_BUSY_CU_CYCLES"):
            cu = float(r["SQ_BUSY_CU_CYCLES"])
            return {"pipe_busy_frac": float(r["SQ_VALU_MFMA_BUSY_CYCLES"]) / (4.0 * cu), "busy_cu_cycles_per_launch": cu,
                    "source": src}
    return None


def pmc_traffic(kernel, A, B, D, H):
    """HBM bytes per launch of `kernel` from the committed PMC passes (two separate `rocprofv3 --pmc FETCH_SIZE` /
    `--pmc WRITE_SIZE` runs of this bench command, summarised by tools/pmc_summary.py).  FETCH_SIZE (KB) is doubled as
    the gfx950 guide prescribes for 16-byte-per-lane loads -- which is how k_fc11_zg and k_fc1_fwd_v3 read x and the
    weights -- and WRITE_SIZE (KB) is taken as is.  Other kernels: null (their load widths are uncalibrated)."""
    if kernel not in ("k_fc11_zg", "k_fc1_fwd_v3", "k_x3_fc11g"):
        return None, None
    rows, src = _pmc_rows("hbm", A, B, D, H)
    for r in rows or []:
        if r["kernel"].startswith(kernel) and r["FETCH_SIZE"] and r["WRITE_SIZE"]:
            return (2.0 * float(r["FETCH_SIZE"]) + float(r["WRITE_SIZE"])) * 1024.0, src
    return None, None


if __name__ == "__main__":
    main()
