#!/bin/bash
# usage: tools/variants_wide.sh <so> ...   -- kernel time of the multi-wave shapes per library variant
for ARGS in "--envs 1024 --drones 256 --buildings 50 --map 100 100 10" "--envs 1310 --drones 200 --map 90 90 10" "--envs 2048 --drones 128 --map 70 70 10" "--envs 512 --drones 512 --map 140 140 10" "--envs 4096 --drones 64"; do
  for SO in "$@"; do
    python tools/bench_variant.py $SO --no-cpu-baseline --steps 300 $ARGS 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('%-16s %-58s kernel %.2f us' % (sys.argv[1], sys.argv[2], r['kernel_ms'] * 1e3))" "$(basename $SO)" "$ARGS"
  done
done
