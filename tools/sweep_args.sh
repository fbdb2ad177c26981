#!/bin/bash
# usage: tools/sweep_args.sh "--envs 1024" "--envs 2048" ...  -- one bench line per argument set
for cfg in "$@"; do
  out=$(timeout -k 10 120 python bench.py --steps 200 --warmup 20 --no-cpu-baseline $cfg) || exit 1
  echo "$cfg: $(echo "$out" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["roofline"]["kernel_ms"], d["value"], d["roofline"]["frac"])')"
done
