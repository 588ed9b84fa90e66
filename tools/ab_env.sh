#!/bin/bash
# step time with an environment switch off / on, alternating, R rounds:  tools/ab_env.sh MMVAE_BN_PARTIALS 3 [bench args]
V=$1; R=${2:-2}; shift; shift
for i in $(seq 1 $R); do
  for X in 0 1; do
    env $V=$X python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline --no-bf16 --no-eval "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$V=$X', round(d['ms_per_step']*1000,1), 'us; median', round(d['ms_per_step_hip_events']['median']*1000,1))"
  done
done
