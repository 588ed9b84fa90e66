#!/bin/bash
# A/B of the augmented shuffled epoch: rows read in place against gathered batches, with per-kernel average durations
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for rows in 1 0 1 0; do MMVAE_ROWS=$rows python3 $R/tools/aug_epoch_time.py fp32 2>&1 | grep "per augmented"; done
for rows in 1 0; do
  rm -rf /tmp/ab$rows
  MMVAE_ROWS=$rows rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab$rows -- python3 $R/tools/aug_epoch_time.py fp32 > /tmp/ab$rows.log 2>&1
  echo "== MMVAE_ROWS=$rows (under the profiler): $(grep 'per augmented' /tmp/ab$rows.log)"
  python3 - <<PY
import csv, glob
f = glob.glob("/tmp/ab$rows/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(f"  {float(r['AverageNs'])/1e3:8.1f} us x {r['Calls']:>4}  {r['Name'][:90]}")
PY
done
