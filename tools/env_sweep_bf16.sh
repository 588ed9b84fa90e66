#!/bin/bash
# step time of the bf16 configuration (bench.py's bf16_config object) under a list of environment assignments, alternating rounds
R=${ROUNDS:-2}
for i in $(seq 1 $R); do
  for kv in "$@"; do
    env $kv timeout -k 10 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-roofline --no-eval $BENCH_ARGS 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read()); b=d['bf16_config']; print('$kv', 'bf16', round(b['ms_per_step']*1000,1), 'us  fp32', round(d['ms_per_step']*1000,1))" || echo "$kv FAILED"
  done
done
