#!/bin/bash
# step time under each variant library in ab/ (and the regular build), alternating, R rounds
R=${1:-2}
for i in $(seq 1 $R); do
  for L in "" $(ls ab/lib*.so 2>/dev/null); do
    if [ -n "$L" ]; then export MMVAE_LIB=$PWD/$L; else unset MMVAE_LIB; fi
    python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-roofline --no-bf16 --no-eval 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('${L:-default}', round(d['ms_per_step']*1000,1), 'us; median', round(d['ms_per_step_hip_events']['median']*1000,1))"
  done
done
