#!/bin/bash
# usage: tools/sweep_env.sh "VAR=val VAR2=val" ...   -- one bench line per environment setting
for cfg in "$@"; do
  out=$(env $cfg timeout -k 10 120 python bench.py --steps 200 --warmup 20 --no-cpu-baseline) || exit 1
  echo "$cfg: $(echo "$out" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["roofline"]["kernel_ms"], d["value"], d["roofline"]["frac"])')"
done
