#!/bin/bash
# rocprofv3 evidence for an arbitrary command (run through gpurun from the repo root):
#   gpurun --timeout 1500 -- 'bash tools/profile_cmd.sh c5 python tools/config5.py --res 512 --spp 16 --steps 2'
# writes gpurun_out/<tag>_{kt,fetch,write,l2,sq,sq2}/ ; summarise with: python tools/prof_summary.py <tag> <tag>_
set -u
TAG=$1; shift
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="$@"
cd $REPO
run() { d=$1; shift; rm -rf $OUT/${TAG}_$d; ( cd /tmp && rocprofv3 "$@" -d $OUT/${TAG}_$d -- bash -c "cd $REPO && $CMD" > $OUT/${TAG}_$d.log 2>&1 ); }
run kt --kernel-trace --stats
run fetch --kernel-trace --pmc FETCH_SIZE
run write --kernel-trace --pmc WRITE_SIZE
run l2 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum
run sq --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES
run sq2 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM
tail -3 $OUT/${TAG}_kt.log
