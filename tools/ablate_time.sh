#!/bin/bash
# usage: tools/ablate_time.sh [bench args]  -- kernel time of the diag build with phases switched off
for AB in 0 1 2 4 7 8 16 24 31 32 64; do
  RVO3D_ABLATE=$AB python tools/bench_diag.py --no-cpu-baseline --steps 200 "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
names = {0: 'full', 1: 'no sweep A', 2: 'no collision sweep', 4: 'no final sweep', 7: 'no sweeps', 8: 'no kept rows / proprio staging', 16: 'no row fill', 24: 'no obs writes', 31: 'no sweeps, no obs writes', 32: 'no X2', 64: 'no X1+X2'}
print('%-34s kernel %.2f us' % (names[d['config']['ablate']], r['kernel_ms'] * 1e3))"
done
