#!/bin/bash
# PMC passes over a replay of one stage (tools/stage_time.py <stage>): tools/pmc_stage.sh <name> <stage> [lib]
# (run ON the GPU box from the repo root; counters only, one pass per group)
name=$1; stage=$2; lib=$3
ROOT=$PWD; OUT=$ROOT/gpurun_out/$name; mkdir -p $OUT
[ -n "$lib" ] && export MMVAE_LIB=$ROOT/$lib
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS" ; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -- python3 $ROOT/tools/stage_time.py $stage > $OUT/p$i.out 2> $OUT/p$i.err
done
cd $ROOT
python3 tools/pmc_summary.py $OUT/p1 > $OUT/summary.csv
python3 - $OUT <<'P'
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/p1/**/*kernel_trace.csv',recursive=True)
if f:
    d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in csv.DictReader(open(f[0])) if 'fc11g' in r['Kernel_Name']]
    print('fc11g launches',len(d),'median us',sorted(d)[len(d)//2])
P
rm -rf $OUT/p1
grep "kernel\|fc11g" $OUT/summary.csv
