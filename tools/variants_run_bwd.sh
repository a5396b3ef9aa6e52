#!/bin/bash
# on the GPU box: reverse-mode timings (bench.py's `backward` object) of config 5 (class-2 variants) or C3 (class-1 variants)
#   bash tools/variants_run_bwd.sh c5|c3 name...
what=$1; shift
mkdir -p gpurun_out
for v in "$@"; do
  if [ "$what" = c5 ]; then args="--config 5 --steps 2 --warmup 1"; else args="--no-config5"; fi
  python tools/variants.py run $v python bench.py $args --no-cpu-baseline --no-parity > gpurun_out/var_$v.json 2> gpurun_out/var_$v.err
  python - $v <<'PY'
import json, sys
v = sys.argv[1]
try:
    d = json.loads(open("gpurun_out/var_%s.json" % v).read().strip().splitlines()[-1])
    b = d.get("backward") or d.get("config5", {}).get("backward")
    print(v, "fwd ms", d.get("ms_per_step"), "bwd ms", b["ms"], {k: round(x, 2) for k, x in b["ms_per_term"].items()})
except Exception as e:
    print(v, "failed", e, open("gpurun_out/var_%s.err" % v).read()[-400:])
PY
done
