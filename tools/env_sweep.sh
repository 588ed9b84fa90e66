#!/bin/bash
# step time under a list of environment assignments (one per argument, "NAME=V NAME2=V2" allowed), alternating rounds
R=${ROUNDS:-2}
for i in $(seq 1 $R); do
  for kv in "$@"; do
    env $kv timeout -k 10 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-roofline --no-eval --no-bf16 $BENCH_ARGS 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$kv', round(d['ms_per_step']*1000,1), 'us  median', round(d['ms_per_step_hip_events']['median']*1000,1))" || echo "$kv FAILED"
  done
done
