#!/bin/bash
# one train step's kernel timeline under a list of environment assignments: tools/step_tl.sh "X=0" "MMVAE_CHAIN2=0" ...   (GPU box, repo root)
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do
  rm -rf /tmp/stl
  env $kv rocprofv3 --kernel-trace --output-format csv -d /tmp/stl -- python3 $R/bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-eval --no-roofline --no-bf16 --no-other-configs $BENCH_ARGS > /tmp/stl.json 2>/tmp/stl.err
  echo "== $kv: $(python3 -c "import json;print(round(json.load(open('/tmp/stl.json'))['ms_per_step']*1000,1),'us per step under the profiler')")"
  python3 $R/tools/step_timeline.py /tmp/stl
done
