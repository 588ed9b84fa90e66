#!/bin/bash
# Build a variant of the WHOLE library with extra flags: tools/mkvariant_all.sh NAME "-DX=1 ..."  -> ab/libNAME.so
set -e
cd "$(dirname "$0")/.."
N=$1; F=$2
C=distributed-vae_amd/csrc
mkdir -p ab/$N
for S in api gemm_big gemm_fast gemm_bf16 chain rowwise consensus augment datapath; do
  if [ $S = gemm_bf16 ]; then X="-fno-slp-vectorize"; else X="-mllvm -amdgpu-mfma-vgpr-form=1"; fi
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -fno-gpu-rdc $X $F -c $C/$S.hip -o ab/$N/$S.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab/lib$N.so ab/$N/*.o
echo ab/lib$N.so
