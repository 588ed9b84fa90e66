#!/bin/bash
# A/B timing of two builds of libmmvae_hip.so on one box: tools/ab_bench.sh ab/lib_base.so ab/lib_b.so [rounds]
# (alternating runs; box-to-box variance is larger than most kernel-level gains)
A=$1; B=$2; R=${3:-3}
for i in $(seq 1 $R); do
  for L in $A $B; do
    MMVAE_LIB=$PWD/$L timeout -k 10 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --no-eval 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', round(d['ms_per_step']*1000,1), 'us')" || exit 1
  done
done
