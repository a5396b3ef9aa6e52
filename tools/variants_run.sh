#!/bin/bash
# on the GPU box: times config 5 (full size) and the sphere box for each named variant of _dev/variants (tools/variants.py)
#   gpurun -- 'bash tools/variants_run.sh base4 sm36 ...'   -> gpurun_out/var_<name>.{json,sphere}
mkdir -p gpurun_out
for v in "$@"; do
  python tools/variants.py run $v python bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-backward > gpurun_out/var_$v.json 2> gpurun_out/var_$v.err
  python tools/variants.py run $v python tools/bench_scene.py sphere > gpurun_out/var_$v.sphere 2>> gpurun_out/var_$v.err
  python - "$v" <<'PY'
import json, sys
v = sys.argv[1]
try:
    d = json.loads(open("gpurun_out/var_%s.json" % v).read().strip().splitlines()[-1])
    c = d.get("config5", d)
    ks = c.get("roofline", {}).get("kernels") or []
    print(v, "c5 ms", d.get("ms_per_step"), [(k["kernel"], round(k["avg_launch_ms"], 1)) for k in ks])
except Exception as e:
    print(v, "failed", e)
print(open("gpurun_out/var_%s.sphere" % v).read().strip().splitlines()[-1:])
PY
done
