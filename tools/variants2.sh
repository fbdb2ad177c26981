#!/bin/bash
for SO in "$@"; do
  for ARGS in "--envs 2731 --drones 96 --map 60 60 10" "--envs 1310 --drones 200 --map 90 90 10" "--envs 2048 --drones 100 --map 60 60 10"; do
    python tools/bench_variant.py $SO --no-cpu-baseline --steps 200 $ARGS 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('%-30s %-50s kernel %.2f us frac %.3f' % (sys.argv[1], sys.argv[2], r['kernel_ms'] * 1e3, r['frac']))" "$(basename $SO)" "$ARGS"
  done
done
