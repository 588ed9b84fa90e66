"""Builds libmmvae_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

hipcc cross-compiles without a GPU; the resulting .so is git-ignored but travels to the GPU box
with the repo snapshot.  Usage: ``python -m distributed_vae_amd.build`` or ``build_native()``.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmmvae_hip.so")
SOURCES = ["api.hip", "gemm_big.hip", "gemm_fast.hip", "gemm_bf16.hip", "gemm_pp.hip", "chain.hip", "rowwise.hip", "consensus.hip", "augment.hip", "datapath.hip", "dp.hip"]
HEADERS = ["common.hpp", "couple.hpp", "tune.h", os.path.join("..", "..", "include", "mmvae.h")]
# -amdgpu-mfma-vgpr-form: keep MFMA accumulators in VGPRs (gfx950 has a unified register file); without it
# hipcc parks loop-carried accumulators in AGPRs and copies all 64 of them out and back every K tile.
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-fno-gpu-rdc", "-mllvm", "-amdgpu-mfma-vgpr-form=1"]


# per-source extra flags.  gemm_bf16.hip: hipcc's SLP vectoriser packs the fp32 subtractions of the operand split into
# v_pk_add_f32, which issues very slowly beside a partner wave's MFMAs (stage phase 3100 -> 1780 cycles without it)
EXTRA = {"gemm_bf16.hip": ["-fno-slp-vectorize"]}
# gemm_bf16.hip is compiled WITHOUT -amdgpu-mfma-vgpr-form: its accumulators are touched by MFMAs only (and once by the
# epilogue), so they can live in the accumulator half of the register file and leave the 256 architectural VGPRs to
# operand fragments in flight
DROP = {"gemm_bf16.hip": ["-amdgpu-mfma-vgpr-form=1"], "gemm_pp.hip": ["-amdgpu-mfma-vgpr-form=1"]}


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need the ROCm toolchain to build libmmvae_hip.so)")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_native(force: bool = False, verbose: bool = True) -> str:
    hipcc = _hipcc()
    objdir = os.path.join(CSRC, "_obj")
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]

    def compile_one(src):
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + hdrs):
            flags = list(FLAGS)
            for f in DROP.get(src, []):
                i = flags.index(f)
                del flags[i - 1:i + 1]          # the option and its "-mllvm"
            cmd = [hipcc] + flags + EXTRA.get(src, []) + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
        return o

    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    if force or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build_native(force="--force" in sys.argv)
    print(LIB)
