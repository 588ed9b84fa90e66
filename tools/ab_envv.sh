#!/bin/bash
# step time for several values of an environment switch, alternating, R rounds:  tools/ab_envv.sh VAR "0 1 2" 3 [bench args]
V=$1; VALS=$2; R=${3:-2}; shift; shift; shift
for i in $(seq 1 $R); do
  for X in $VALS; do
    env $V=$X python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline --no-bf16 --no-eval "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$V=$X', round(d['ms_per_step']*1000,1), 'us; median', round(d['ms_per_step_hip_events']['median']*1000,1))"
  done
done
