#!/bin/bash
# Collect this round's measurement artefacts ON the GPU box (run from the repo root through gpurun); writes
# gpurun_out/<round>/..., which the builder then copies into profiles/.
#   tools/collect_profiles.sh r02
R=${1:-r04}
ROOT=$PWD
OUT=$ROOT/gpurun_out/$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-eval --no-roofline --no-bf16 --no-other-configs"
# 1. kernel stats of the fp32 train step (default engine: fp32x3), of the same step on the fp32 matrix instruction and of
#    the bf16 configuration (on bf16 storage, its default, and on the fp32 matrix)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kstats_fp32 -- $BENCH > $OUT/kstats_fp32.json 2> $OUT/kstats_fp32.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kstats_mfma -- $BENCH --gemm-dtype fp32_mfma > $OUT/kstats_mfma.json 2> $OUT/kstats_mfma.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kstats_bf16 -- $BENCH --gemm-dtype bf16 > $OUT/kstats_bf16.json 2> $OUT/kstats_bf16.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kstats_bf16f -- $BENCH --gemm-dtype bf16 --bf16-fp32-storage > $OUT/kstats_bf16f.json 2> $OUT/kstats_bf16f.err
cp $(find $OUT/kstats_fp32 -name "*kernel_stats.csv" | head -1) $OUT/${R}_bench_kernel_stats.csv
# one step's kernel timeline (start offset, duration, queue)
python3 $ROOT/tools/step_timeline.py $OUT/kstats_fp32 > $OUT/${R}_step_timeline.txt
cp $(find $OUT/kstats_mfma -name "*kernel_stats.csv" | head -1) $OUT/${R}_bench_fp32mfma_kernel_stats.csv
cp $(find $OUT/kstats_bf16 -name "*kernel_stats.csv" | head -1) $OUT/${R}_bench_bf16_kernel_stats.csv
cp $(find $OUT/kstats_bf16f -name "*kernel_stats.csv" | head -1) $OUT/${R}_bench_bf16_fp32storage_kernel_stats.csv
# the augmenter forward (production path: the reference's default --augmentation True): kernel stats and one forward's per-layer
# timeline, fp32x3 and bf16 operands
for m in fp32 bf16; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/aug_$m -- python3 $ROOT/tools/aug_time.py $m > $OUT/aug_$m.log 2> $OUT/aug_$m.err
  cp $(find $OUT/aug_$m -name "*kernel_stats.csv" | head -1) $OUT/${R}_augmenter_${m}_kernel_stats.csv
  { grep "ms per batch" $OUT/aug_$m.log; python3 $ROOT/tools/aug_timeline.py $OUT/aug_$m; } > $OUT/${R}_augmenter_${m}_timeline.txt
done
cp $OUT/${R}_augmenter_fp32_kernel_stats.csv $OUT/${R}_augmenter_kernel_stats.csv
# 2. PMC passes (own runs, counters only): HBM traffic, matrix-pipe utilisation
SHORT="python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-eval --no-roofline --no-bf16 --no-other-configs"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -- $SHORT > /dev/null 2> $OUT/pmc_$c.err
  rocprofv3 --pmc $c --output-format csv -d $OUT/pmcb_$c -- $SHORT --gemm-dtype bf16 > /dev/null 2> $OUT/pmcb_$c.err
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_mfma1 -- $SHORT > /dev/null 2> $OUT/pmc_mfma1.err
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $OUT/pmc_mfma2 -- $SHORT > /dev/null 2> $OUT/pmc_mfma2.err
cd $ROOT
python3 tools/pmc_summary.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE > $OUT/${R}_pmc_hbm_summary.csv
python3 tools/pmc_summary.py $OUT/pmcb_FETCH_SIZE $OUT/pmcb_WRITE_SIZE > $OUT/${R}_pmc_hbm_bf16_summary.csv
python3 tools/pmc_summary.py $OUT/pmc_mfma1 $OUT/pmc_mfma2 > $OUT/${R}_pmc_mfma_summary.csv
python3 - <<PY
import hashlib, json
srcs = ["distributed-vae_amd/csrc/gemm_fast.hip", "distributed-vae_amd/csrc/gemm_bf16.hip", "distributed-vae_amd/csrc/common.hpp", "distributed-vae_amd/csrc/chain.hip", "distributed-vae_amd/csrc/api.hip"]
json.dump({"sources_sha256": {s: hashlib.sha256(open(s, "rb").read()).hexdigest() for s in srcs},
           "command": "tools/collect_profiles.sh $R"}, open("$OUT/${R}_pmc_meta.json", "w"), indent=1)
PY
# drop the bulky raw traces from what travels back (keep the summaries)
rm -rf $OUT/aug_fp32 $OUT/aug_bf16 $OUT/kstats_fp32 $OUT/kstats_mfma $OUT/kstats_bf16 $OUT/kstats_bf16f $OUT/pmc_* $OUT/pmcb_*
ls -la $OUT
