"""CPU (hipcc cross-compiles without a GPU): a static check of the generated gfx950 code for the store-data hazard that
once corrupted dZ11 -- a VALU write of the data registers of a 128-bit buffer store within two wait states of the
store.  hipcc pads it for global stores and for buffer stores with an immediate offset, NOT for
`buffer_store_dwordx4 v[..], v, s[..], sN offen` (csrc/gemm_fast.hip pads it by hand)."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "distributed-vae_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


# (source, the flags distributed-vae_amd/build.py compiles it with): the fp32 matrix-instruction kernels and the bf16 /
# fp32x3 engine, whose fused fc11 kernel (k_x3_fc11g) stores dZ11 the same way
SOURCES = [("gemm_fast.hip", ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]), ("gemm_bf16.hip", ["-fno-slp-vectorize"])]


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
@pytest.mark.parametrize("src,flags", SOURCES)
def test_no_valu_write_within_two_wait_states_of_a_wide_buffer_store(src, flags):
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, src.replace(".hip", ".s"))
        cmd = [HIPCC, "-O3", "--offload-arch=gfx950", "-std=c++17", "-fno-gpu-rdc"] + flags + [
               "-S", "--cuda-device-only", "-o", asm, os.path.join(CSRC, src)]
        subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        lines = open(asm).read().splitlines()
    stores = hazards = 0
    for i, line in enumerate(lines):
        m = re.search(r"buffer_store_dwordx[34] v\[(\d+):(\d+)\]", line)
        if not m:
            continue
        stores += 1
        lo, hi = int(m.group(1)), int(m.group(2))
        j, wait = i + 1, 0
        while j < len(lines) and wait < 2:
            t = lines[j].strip()
            j += 1
            if not t or t[0] in ";.":
                continue
            if t.startswith("s_nop"):
                wait += int(t.split()[1]) + 1
                continue
            w = re.match(r"v_\w+\s+v\[?(\d+)(?::(\d+))?\]?", t)
            if w:
                a, b = int(w.group(1)), int(w.group(2) or w.group(1))
                if not (b < lo or a > hi):
                    hazards += 1
            wait += 1
    assert stores >= 8, "the fused fc11 kernels no longer use buffer stores? update this test"
    assert hazards == 0, f"{hazards} of {stores} wide buffer stores are followed by a VALU write of their data registers"
