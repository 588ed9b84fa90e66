#!/bin/bash
# stage timings of the split engine under each variant library in ab/
for L in "" $(ls ab/lib*.so); do
  echo "== ${L:-default}"
  if [ -n "$L" ]; then export MMVAE_LIB=$PWD/$L; else unset MMVAE_LIB; fi
  python tools/bf16_stage_time.py fp32x3 2>/dev/null
done
