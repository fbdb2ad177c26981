#!/bin/bash
# experiment: early zero fill (diag build, RVO3D_ABLATE=128) vs normal, several shapes
for AB in 0 128; do
  for ARGS in "--envs 4096 --drones 64" "--envs 32768 --drones 64" "--envs 1024 --drones 256 --buildings 50 --map 100 100 10"; do
    RVO3D_ABLATE=$AB python tools/bench_diag.py --no-cpu-baseline --steps 200 $ARGS 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('ablate', d['config']['ablate'], sys.argv[1], 'kernel %.2f us' % (r['kernel_ms'] * 1e3))" "$ARGS"
  done
done
