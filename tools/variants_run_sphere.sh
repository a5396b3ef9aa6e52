# on the GPU box: the sphere tutorial box at depth 3 (tools/bench_scene.py sphere) for the named variants
for v in "$@"; do echo -n "$v "; timeout 150 python tools/variants.py run $v python tools/bench_scene.py sphere 2>/dev/null | tail -1; done
python tools/variants.py restore
