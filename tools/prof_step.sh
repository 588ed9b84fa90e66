#!/bin/bash
# rocprofv3 kernel stats of the train step only: tools/prof_step.sh <outdir-name> [env assignments...]
# (run ON the GPU box, from the repo root)
name=$1; shift
out=$PWD/gpurun_out/$name
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 $OLDPWD/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-eval --no-roofline $BENCH_ARGS > $out/bench.json 2> $out/prof.err
f=$(find $out/prof -name "*kernel_stats.csv" | head -1)
cp "$f" $out/kernel_stats.csv
python3 - "$out/kernel_stats.csv" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=0
for r in rows:
    n=r["Name"]; 
    if "mmvae::" not in n: continue
    short=n.split("mmvae::")[1].split("(")[0][:40]
    calls=int(r["Calls"]); avg=float(r["AverageNs"])/1e3
    per_step=float(r["TotalDurationNs"])/1e3/60.0
    tot+=per_step
    print(f"{short:42s} calls {calls:5d} avg {avg:8.1f} us  per-step {per_step:7.1f} us")
print("sum per step", round(tot,1))
PY
