#!/bin/bash
for i in 1 2; do
for L in "" $(ls ab/lib*.so 2>/dev/null); do
  if [ -n "$L" ]; then export MMVAE_LIB=$PWD/$L; else unset MMVAE_LIB; fi
  python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-roofline --no-bf16 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); p=d['data_path']; print('${L:-default}', 'gather', round(p['us_per_batch'],1), 'us; epoch step pipelined', round(p['shuffled_epoch_ms_per_step_pipelined']*1000,1), '; fixed-batch step', round(d['ms_per_step']*1000,1))"
done; done
