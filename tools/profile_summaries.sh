#!/bin/bash
# local half of tools/profile.sh:  bash tools/profile_summaries.sh r03b
set -eu
TAG=$1
cd "$(dirname "$0")/.."
python tools/prof_summary.py $TAG > /dev/null
python tools/prof_summary.py ${TAG}_bwd bwd_ > /dev/null
python tools/prof_summary.py ${TAG}_config5 c5_ config5 "BASELINE config 5, 1024 x 1024, spp = sppe = sppse = 64, guiding grid (python bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-backward --no-roofline)" > /dev/null
python tools/prof_summary.py ${TAG}_config5_bwd c5bwd_ > /dev/null
python tools/prof_summary.py ${TAG}_sphere sph_ > /dev/null
cp gpurun_out/bench_line.json profiles/${TAG}_bench.json
ls profiles | grep "^$TAG"
