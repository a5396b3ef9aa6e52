#!/bin/bash
# same-box A/B of measurement variants (tools/variants.py build NAME "-D...") on the C3 timed loop, with the shader clock sampled while each runs:
#   bash tools/ab_c3.sh NAME [NAME ...]          ("default" = the in-tree library)
B="python bench.py --steps 1500 --warmup 3 --no-cpu-baseline --no-parity --no-backward --no-config5 --no-roofline"
for v in "$@"; do
    if [ "$v" = default ]; then python tools/variants.py restore > /dev/null 2>&1; CMD="$B"; else CMD="python tools/variants.py run $v $B"; fi
    ( $CMD 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('%-12s %.3f ms' % ('$v', d['ms_per_step']))" ) &
    sleep 6; rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power (W)" | tr -s ' \t' ' ' | tr '\n' '|'; echo; wait
done
python tools/variants.py restore > /dev/null 2>&1
