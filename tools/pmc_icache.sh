#!/bin/bash
# usage: tools/pmc_icache.sh <tag> [bench args]  -- instruction-cache hits / misses per launch of the step kernel
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmci_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES \
  --output-format csv -d $OUT/p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > $OUT/p.log 2>&1
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("$OUT/p/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "env_kernel<2" in r["Kernel_Name"]:
            a = agg[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
r = {k: v / n for k, (v, n) in agg.items()}
print("$TAG:", {k: round(v) for k, v in r.items()})
if r.get("SQC_ICACHE_REQ"):
    print("  icache miss rate %.2f %% (+ duplicate %.2f %%), requests per wave %.0f" % (100 * r["SQC_ICACHE_MISSES"] / r["SQC_ICACHE_REQ"],
          100 * r.get("SQC_ICACHE_MISSES_DUPLICATE", 0) / r["SQC_ICACHE_REQ"], r["SQC_ICACHE_REQ"] / r["SQ_WAVES"]))
PY
