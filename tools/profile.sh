#!/bin/bash
# Collects the rocprofv3 evidence for one stage on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1200 -- 'bash tools/profile.sh'
# then locally:  python tools/prof_summary.py <tag>
# Counters are collected in their own passes with --kernel-trace only (no sys/hip/hsa trace domains).
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p $OUT
BENCH="python $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-parity"
cd /tmp && export TMPDIR=/tmp
python $REPO/bench.py > $OUT/bench_line.json 2> $OUT/bench_err.log
rm -rf $OUT/prof_kt $OUT/prof_fetch $OUT/prof_write $OUT/prof_l2 $OUT/prof_sq $OUT/prof_sq2
rocprofv3 --kernel-trace --stats -d $OUT/prof_kt -- $BENCH > $OUT/prof_kt.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/prof_fetch -- $BENCH > $OUT/prof_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/prof_write -- $BENCH > $OUT/prof_write.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d $OUT/prof_l2 -- $BENCH > $OUT/prof_l2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d $OUT/prof_sq -- $BENCH > $OUT/prof_sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA -d $OUT/prof_sq2 -- $BENCH > $OUT/prof_sq2.log 2>&1
tail -1 $OUT/bench_line.json
ls $OUT
