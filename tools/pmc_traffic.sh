#!/bin/bash
# usage: tools/pmc_traffic.sh <tag> [ENV=val ...]  -- FETCH_SIZE / WRITE_SIZE per launch (two PMC passes)
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmct_$TAG
mkdir -p $OUT
for kv in "$@"; do export "$kv"; done
BENCH_ARGS=${BENCH_ARGS:-}
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $C --output-format csv -d $OUT/$C -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --prewarm 20 --no-cpu-baseline $BENCH_ARGS > $OUT/$C.log 2>&1
done
python3 - <<PY
import csv, glob
r = {}
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    v = [float(x["Counter_Value"]) for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % C, recursive=True)
         for x in csv.DictReader(open(f)) if "env_kernel<2" in x["Kernel_Name"] and x["Counter_Name"] == C]
    r[C] = sum(v) / len(v)
print("$TAG: FETCH_SIZE %.0f KB (x2 on gfx950)  WRITE_SIZE %.0f KB  -> %.1f MB per launch" %
      (r["FETCH_SIZE"], r["WRITE_SIZE"], (2 * r["FETCH_SIZE"] + r["WRITE_SIZE"]) * 1024 / 1e6))
PY
