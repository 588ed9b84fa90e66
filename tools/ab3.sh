#!/bin/bash
# three-way A/B on one box: old-kernel library, new library with tickets off, new with tickets on (alternating rounds)
R=${1:-3}
run() { # label, env...
  lbl=$1; shift
  env "$@" timeout -k 10 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-roofline --no-eval 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lbl', round(d['ms_per_step']*1000,1), 'us  median', round(d['ms_per_step_hip_events']['median']*1000,1))" || echo "$lbl FAILED"
}
for i in $(seq 1 $R); do
  run old   MMVAE_LIB=$PWD/gpurun_in/libmmvae_r01kernels.so
  run new_t0 MMVAE_TICKET=0
  run new_t1 MMVAE_TICKET=1
done
