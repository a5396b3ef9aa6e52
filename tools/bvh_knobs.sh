#!/bin/bash
# BVH builder knobs (bvh.h: PSDR_BVH_BINS / PSDR_BVH_SWEEP / PSDR_BVH_LEAFCOST / PSDR_BVH_LEAF), each timed on config 5 at full size in one GPU call:
#   gpurun --timeout 1200 -- 'bash tools/bvh_knobs.sh > gpurun_out/bvh_knobs.log 2>&1'
run() {
    echo "== $*"
    env "$@" python bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-backward 2>/dev/null | python -c '
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d["roofline"]
print("ms_per_step %.1f  primary %.1f ms  nodes/ray %.2f tris/ray %.2f  bvh %s" % (d["ms_per_step"], r["avg_launch_ms"], r["nodes_per_ray"], r["tris_per_ray"], r["bvh"]))
print("   ", [(k["kernel"], round(k["avg_launch_ms"], 1)) for k in r["kernels"]])'
}
run PSDR_X=0
run PSDR_BVH_BINS=32
run PSDR_BVH_BINS=64
run PSDR_BVH_SWEEP=64
run PSDR_BVH_SWEEP=1024
run PSDR_BVH_SWEEP=1000000
run PSDR_BVH_LEAFCOST=1
run PSDR_BVH_LEAFCOST=1 PSDR_BVH_SWEEP=1024
run PSDR_BVH_LEAF=1
