#!/bin/bash
# usage: tools/variants.sh <so> ...   -- kernel time of the three benchmark shapes for each library variant
for SO in "$@"; do
  for ARGS in "--envs 4096 --drones 64" "--envs 1024 --drones 256 --buildings 50 --map 100 100 10" "--envs 2048 --drones 128 --map 70 70 10"; do
    python tools/bench_variant.py $SO --no-cpu-baseline --steps 200 $ARGS 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('%-40s %-60s kernel %.2f us' % (sys.argv[1], sys.argv[2], r['kernel_ms'] * 1e3))" "$(basename $SO)" "$ARGS"
  done
done
