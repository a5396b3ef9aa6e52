#!/bin/bash
# Collects the rocprofv3 evidence of one stage on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 2400 -- 'bash tools/profile.sh'
# then locally:  bash tools/profile_summaries.sh <tag>      (e.g. r03b; writes profiles/<tag>_* and profiles/counters.json)
# Five commands, each under --kernel-trace --stats and under five --pmc passes of their own (no sys/hip/hsa trace domains):
#   (default dirs prof_*)  bench.py's timed loop alone: the C3 forward kernels
#   bwd_     reverse mode on C3 alone (tools/bwd_only.py c3)
#   c5_      BASELINE config 5 at full size (bench.py --config 5: 1024 x 1024, 64 / 64 / 64, guiding grid)
#   c5bwd_   reverse mode on config 5 at full size with its guiding grid (tools/bwd_only.py c5: 1024 x 1024 x 64)
#   sph_     the sphere tutorial box at depth 3 (tools/bench_scene.py sphere --terms-only: one term per launch, every sample traced)
# The config-5 and sphere passes run with PSDR_NO_FORK=1: the three terms of a renderD one after the other instead of on forked streams, so that a kernel's
# duration in the trace is its own (overlapping launches each show the whole call's time); bench.py's own numbers are taken WITH the fork.
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p $OUT
BENCH="python $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-parity --no-backward --no-config5 --no-api --no-static-skip"
cd /tmp && export TMPDIR=/tmp
python $REPO/bench.py > $OUT/bench_line.json 2> $OUT/bench_err.log
rm -rf $OUT/prof_kt $OUT/prof_fetch $OUT/prof_write $OUT/prof_l2 $OUT/prof_sq $OUT/prof_sq2
rocprofv3 --kernel-trace --stats -d $OUT/prof_kt -- $BENCH > $OUT/prof_kt.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/prof_fetch -- $BENCH > $OUT/prof_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/prof_write -- $BENCH > $OUT/prof_write.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d $OUT/prof_l2 -- $BENCH > $OUT/prof_l2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d $OUT/prof_sq -- $BENCH > $OUT/prof_sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA -d $OUT/prof_sq2 -- $BENCH > $OUT/prof_sq2.log 2>&1
cd $REPO
bash tools/profile_cmd.sh bwd python tools/bwd_only.py c3 5
PSDR_NO_FORK=1 bash tools/profile_cmd.sh c5 python bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-backward --no-roofline --no-api --no-static-skip
bash tools/profile_cmd.sh c5bwd python tools/bwd_only.py c5 2 1024 64
PSDR_NO_FORK=1 bash tools/profile_cmd.sh sph python tools/bench_scene.py sphere --terms-only
python tools/bench_scene.py sphere --skip > $OUT/sphere_skip.log 2>&1
# the databases travel back through gpurun_out (64 MiB): keep the *_results.db files only
find $OUT -name "*.csv" -size +2M -delete 2> /dev/null
tail -1 $OUT/bench_line.json | cut -c1-400
du -sh $OUT
