#!/bin/bash
# usage: tools/pmc_ablate.sh  (on the GPU box, from the repo root)
# Dynamic instruction counts per wave of the fused step under each RVO3D_ABLATE setting:
# the differences to the full run are the per-phase instruction counts.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_ablate
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for AB in 0 1 2 4 7 32 64 8 16 24; do
  export RVO3D_ABLATE=$AB
  timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR \
    --output-format csv -d $OUT/a$AB -- python3 $GRAFT_REPO_ROOT/tools/bench_diag.py --steps 10 --warmup 3 --no-cpu-baseline ${BENCH_ARGS:-} > $OUT/a$AB.log 2>&1 || echo "ablate $AB failed" >> $OUT/fail.log
done
python3 - <<PY
import csv, glob, collections
rows = {}
for ab in [0, 1, 2, 4, 7, 32, 64, 8, 16, 24]:
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob("$OUT/a%d/**/*counter_collection.csv" % ab, recursive=True):
        for r in csv.DictReader(open(f)):
            if "env_kernel<2" in r["Kernel_Name"]:
                a = agg[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    rows[ab] = {k: v / n for k, (v, n) in agg.items()}
names = {0: "full", 1: "no sweep A", 2: "no collision sweep", 4: "no final sweep", 7: "no sweeps", 32: "no X2", 64: "no X1+X2",
         8: "no obs rows/proprio", 16: "no zero fill", 24: "no obs writes"}
with open("$OUT/summary.txt", "w") as o:
    for ab, r in rows.items():
        if not r: continue
        w = r.get("SQ_WAVES", 1)
        o.write(f"{names[ab]:22s} VALU/wave {r['SQ_INSTS_VALU']/w:8.0f} SALU {r['SQ_INSTS_SALU']/w:7.0f} LDS {r['SQ_INSTS_LDS']/w:6.0f} "
                f"VMEM rd {r['SQ_INSTS_VMEM_RD']/w:5.0f} wr {r['SQ_INSTS_VMEM_WR']/w:5.0f}  VALU busy cyc/wave {4*r['SQ_ACTIVE_INST_VALU']/w:8.0f} wave cyc {4*r['SQ_WAVE_CYCLES']/w:8.0f}\n")
print(open("$OUT/summary.txt").read())
PY
