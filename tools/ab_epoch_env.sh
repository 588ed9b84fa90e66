#!/bin/bash
# shuffled-epoch step time (bench.py data_path section) with an environment switch off / on:  tools/ab_epoch_env.sh VAR R
V=$1; R=${2:-2}
for i in $(seq 1 $R); do for X in 1 0; do
  env $V=$X python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-roofline --no-bf16 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); p=d['data_path']; print('$V=$X', 'epoch step back-to-back', round(p['shuffled_epoch_ms_per_step_back_to_back']*1000,1), 'pipelined', round(p['shuffled_epoch_ms_per_step_pipelined']*1000,1), '; fixed-batch step', round(d['ms_per_step']*1000,1))"
done; done
