#!/bin/bash
# per-layer timeline of the augmenter forward for a list of library builds (diagnostic ablations): tools/aug_abl.sh fp32|bf16 lib1.so lib2.so ...
# run on the GPU box from the repo root
MODE=$1; shift
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for L in "$@"; do
  rm -rf /tmp/augp
  MMVAE_LIB=$R/$L rocprofv3 --kernel-trace --output-format csv -d /tmp/augp -- python3 $R/tools/aug_time.py $MODE > /tmp/augp.log 2>&1
  echo "== $L: $(grep 'ms per batch' /tmp/augp.log)"
  python3 $R/tools/aug_timeline.py /tmp/augp | grep -v "at::native"
done
