#!/bin/bash
# GPU: split-engine tests, stage timings of the three engines, and the step time under each fp32 engine
set -x
mkdir -p gpurun_out
python -m pytest tests/test_gpu_fp32x3.py -x -q -s > gpurun_out/x3_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/x3_tests.log
tail -5 gpurun_out/x3_tests.log
for dt in fp32_mfma bf16 fp32x3; do python tools/bf16_stage_time.py $dt; done 2>&1 | tee gpurun_out/x3_stage.log
for e in fp32_mfma fp32x3; do MMVAE_FP32_ENGINE=$e python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-roofline --no-bf16 --no-eval 2>gpurun_out/x3_bench_$e.err | tee gpurun_out/x3_bench_$e.json | cut -c1-400; done
