#!/bin/bash
# ablations of k_bf16_fc11g: kernel averages of the bf16 step for diagnostic builds distributed-vae_amd/ab/libmmvae_abl<mask>.so, made in
# the build container first (they travel with the snapshot):
#   for m in 1 2 4 7; do hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -fno-gpu-rdc -fno-slp-vectorize -DBF16FC_ABL=$m -c csrc/gemm_bf16.hip -o /tmp/gb_$m.o;
#     hipcc --offload-arch=gfx950 -shared -fPIC -o ab/libmmvae_abl$m.so $(ls csrc/_obj/*.o | grep -v gemm_bf16.o) /tmp/gb_$m.o -ldl; done
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for m in 0 1 2 4 7; do
  lib=$R/distributed-vae_amd/libmmvae_hip.so; [ $m != 0 ] && lib=$R/distributed-vae_amd/ab/libmmvae_abl$m.so
  rm -rf /tmp/abl$m
  MMVAE_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abl$m -- python3 $R/bench.py --gemm-dtype bf16 --steps 30 --warmup 5 --no-cpu-baseline --no-eval --no-roofline --no-bf16 --no-other-configs > /tmp/abl$m.json 2> /tmp/abl$m.err
  python3 - <<PY
import csv, glob, json
f = glob.glob("/tmp/abl$m/**/*kernel_stats.csv", recursive=True)[0]
row = [r for r in csv.DictReader(open(f)) if "k_bf16_fc11g" in r["Name"]][0]
print(f"BF16FC_ABL=$m  k_bf16_fc11g {float(row['AverageNs'])/1e3:7.1f} us   step {json.load(open('/tmp/abl$m.json'))['ms_per_step']:.4f} ms")
PY
done
