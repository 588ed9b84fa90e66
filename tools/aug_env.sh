#!/bin/bash
# per-layer timeline of the augmenter forward under a list of environment assignments: tools/aug_env.sh fp32|bf16 "MMVAE_AUG_TILE=1" ...
MODE=$1; shift
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for kv in "$@"; do
  rm -rf /tmp/augp
  env $kv rocprofv3 --kernel-trace --output-format csv -d /tmp/augp -- python3 $R/tools/aug_time.py $MODE > /tmp/augp.log 2>&1
  echo "== $kv: $(grep 'ms per batch' /tmp/augp.log)"
  python3 $R/tools/aug_timeline.py /tmp/augp | grep -v "at::native"
done
