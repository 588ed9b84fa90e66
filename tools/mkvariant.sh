#!/bin/bash
# Build a variant of libmmvae_hip.so with extra compiler flags for ONE source: tools/mkvariant.sh NAME SRC.hip "-DX=1 ..."
# -> ab/libNAME.so (run with MMVAE_LIB=$PWD/ab/libNAME.so); the other objects come from the regular build.
set -e
cd "$(dirname "$0")/.."
N=$1; S=$2; F=$3
C=distributed-vae_amd/csrc
python distributed-vae_amd/build.py > /tmp/mmvae_build.log 2>&1 || { tail -30 /tmp/mmvae_build.log; exit 1; }
mkdir -p ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -fno-gpu-rdc -fno-slp-vectorize $F -c $C/$S -o ab/$N.o
OBJS=$(ls $C/_obj/*.o | grep -v "/${S%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab/lib$N.so $OBJS ab/$N.o
echo ab/lib$N.so
