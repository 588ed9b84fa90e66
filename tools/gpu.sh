#!/bin/bash
# Build the library (so that the .so that travels is current), then run a command on the GPU box.
#   tools/gpu.sh [--timeout S] -- '<command>'
set -e
cd "$(dirname "$0")/.."
python distributed-vae_amd/build.py > /tmp/mmvae_build.log 2>&1 || { tail -30 /tmp/mmvae_build.log; exit 1; }
make -s -C tools/micro 2>/dev/null || true
exec /usr/local/graft/bin/gpurun "$@"
