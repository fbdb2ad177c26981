#!/bin/bash
# usage: tools/pmc_quick.sh <tag> [ENV=val ...]  -- one PMC pass (instruction counts) of the bench
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcq_$TAG
mkdir -p $OUT
for kv in "$@"; do export "$kv"; done
BENCH_ARGS=${BENCH_ARGS:-}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY \
  --output-format csv -d $OUT/p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline $BENCH_ARGS > $OUT/p.log 2>&1
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("$OUT/p/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "env_kernel<2" in r["Kernel_Name"]:
            a = agg[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
r = {k: v / n for k, (v, n) in agg.items()}
w = r["SQ_WAVES"]
print(f"$TAG: VALU/wave {r['SQ_INSTS_VALU']/w:.0f} SALU {r['SQ_INSTS_SALU']/w:.0f} LDS {r['SQ_INSTS_LDS']/w:.0f} | per wave cycles: life {4*r['SQ_WAVE_CYCLES']/w:.0f} "
      f"VALU busy {4*r['SQ_ACTIVE_INST_VALU']/w:.0f} wait_any {4*r['SQ_WAIT_ANY']/w:.0f} wait_inst {4*r['SQ_WAIT_INST_ANY']/w:.0f}")
PY
