#!/bin/bash
# GPU: split-engine tests, stage timings and the step time (the build must be current: tools/gpu.sh)
timeout -k 10 200 python -m pytest tests/test_gpu_fp32x3.py -x -q 2>&1 | tail -2
timeout -k 10 60 python tools/bf16_stage_time.py fp32x3 2>/dev/null
python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-roofline --no-bf16 --no-eval 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('step', round(d['ms_per_step']*1000,1), 'us; median', round(d['ms_per_step_hip_events']['median']*1000,1))"
