#!/bin/bash
# usage: tools/bench_all.sh [tag]  (GPU box, repo root) -- kernel time of the fused step for the standard shapes
TAG=${1:-run}
OUT=gpurun_out/bench_all_$TAG.txt
: > $OUT
run() {
  python bench.py --no-cpu-baseline --no-rollout --steps 200 "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']
print('%-62s kernel %.2f us frac %.3f | cold %.2f us frac %.3f | value %.3e' % (' '.join(sys.argv[1:]), r['kernel_ms'] * 1e3, r['frac'], r.get('kernel_ms_cold', 0) * 1e3, r.get('frac_cold', 0), d['value']))
" "$@" | tee -a $OUT
}
run --envs 4096 --drones 64
run --envs 1024 --drones 256 --buildings 50 --map 100 100 10
run --envs 256 --drones 16 --map 20 20 8
run --envs 2048 --drones 128 --map 70 70 10
run --envs 8192 --drones 32 --map 35 35 10
run --envs 16384 --drones 16 --map 20 20 8
run --envs 32768 --drones 64
run --envs 2731 --drones 96 --map 60 60 10
run --envs 5461 --drones 48 --map 45 45 10
run --envs 2048 --drones 100 --map 60 60 10
run --envs 1310 --drones 200 --map 90 90 10
run --envs 1638 --drones 160 --map 80 80 10
run --envs 1365 --drones 192 --map 85 85 10
