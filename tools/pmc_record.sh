#!/bin/bash
# usage: tools/pmc_record.sh <tag> [bench args...]      (GPU box, from the repo root)
# The record behind bench.py's roofline.traffic / fp64_valu_tflops for one bench command:
#   pass 0  rocprofv3 --kernel-trace --stats            -> <tag>_kernel_stats.csv (avg launch duration)
#   pass 1  --pmc FETCH_SIZE        pass 2  --pmc WRITE_SIZE      (HBM bytes; FETCH x2 on gfx950)
#   pass 3  fp64 / VALU / SALU instruction counts       pass 4  LDS activity + bank conflicts, wave cycles
# (counters in their own passes, never together with tracing flags).  Writes
# gpurun_out/pmc_record_<tag>/{summary.txt,record.json,kernel_stats.csv}; copy them to profiles/rNN/
# and merge record.json into profiles/traffic.json (tools/merge_traffic.py).
set -u
TAG=$1; shift
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/pmc_record_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $ROOT/bench.py --no-cpu-baseline $*"
echo "command: rocprofv3 ... -- $CMD" > $OUT/command.txt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1 || echo "trace pass failed" >> $OUT/fail.log
i=0
for CTRS in "FETCH_SIZE" "WRITE_SIZE" \
            "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_LDS" \
            "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $CTRS --output-format csv -d $OUT/p$i -- $CMD > $OUT/p$i.log 2>&1 || echo "pass $i failed" >> $OUT/fail.log
done
python3 - "$OUT" "$TAG" "$*" <<'PY'
import collections, csv, glob, json, os, shutil, sys
out, tag, args = sys.argv[1], sys.argv[2], sys.argv[3]
agg = collections.defaultdict(lambda: [0.0, 0])
kname = None
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "env_kernel<2" in r["Kernel_Name"] or "env_kernel<1" in r["Kernel_Name"]:
            kname = r["Kernel_Name"]
            a = agg[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
c = {k: v / n for k, (v, n) in agg.items()}
stats = None
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    shutil.copy(f, out + "/kernel_stats.csv")
    for r in csv.DictReader(open(f)):
        if "env_kernel<2" in r["Name"] or "env_kernel<1" in r["Name"]:
            stats = dict(calls=int(r["Calls"]), avg_ns=float(r["AverageNs"]), min_ns=float(r["MinNs"]), max_ns=float(r["MaxNs"]))
w = c.get("SQ_WAVES", 1.0)
hbm = (2 * c.get("FETCH_SIZE", 0) + c.get("WRITE_SIZE", 0)) * 1024
f64 = 64 * (2 * c.get("SQ_INSTS_VALU_FMA_F64", 0) + c.get("SQ_INSTS_VALU_MUL_F64", 0) + c.get("SQ_INSTS_VALU_ADD_F64", 0) + c.get("SQ_INSTS_VALU_TRANS_F64", 0))
rec = dict(kernel=kname, bench_args=args, hbm_bytes=hbm, fetch_kb_raw=c.get("FETCH_SIZE"), write_kb=c.get("WRITE_SIZE"),
           fp64_flops=f64, valu_per_wave=c.get("SQ_INSTS_VALU", 0) / w, salu_per_wave=c.get("SQ_INSTS_SALU", 0) / w,
           lds_per_wave=c.get("SQ_INSTS_LDS", 0) / w, waves=w,
           lds_bank_conflict_over_active=(c.get("SQ_LDS_BANK_CONFLICT", 0) / c["SQ_ACTIVE_INST_LDS"]) if c.get("SQ_ACTIVE_INST_LDS") else None,
           valu_active_over_wave_cycles=(c.get("SQ_ACTIVE_INST_VALU", 0) / c["SQ_WAVE_CYCLES"]) if c.get("SQ_WAVE_CYCLES") else None,
           wait_any_over_wave_cycles=(c.get("SQ_WAIT_ANY", 0) / c["SQ_WAVE_CYCLES"]) if c.get("SQ_WAVE_CYCLES") else None,
           rocprof_kernel=stats, source="tools/pmc_record.sh " + tag)
json.dump(rec, open(out + "/record.json", "w"), indent=1)
with open(out + "/summary.txt", "w") as o:
    o.write(open(out + "/command.txt").read())
    o.write("kernel: %s\n" % kname)
    if stats:
        o.write("rocprofv3 --kernel-trace --stats: %d launches, avg %.2f us (min %.2f, max %.2f)\n" % (stats["calls"], stats["avg_ns"] / 1e3, stats["min_ns"] / 1e3, stats["max_ns"] / 1e3))
    for k in sorted(c):
        o.write("%s per-launch %.6g over %d launches\n" % (k, c[k], agg[k][1]))
    o.write("HBM traffic per launch = (2 x FETCH_SIZE + WRITE_SIZE) KB = %.1f MB\n" % (hbm / 1e6))
    o.write("fp64 VALU flop per launch (64 lanes per wave-instruction, fma = 2) = %.4g" % f64)
    if stats:
        o.write(" -> %.2f TFLOP/s of 78.6 (%.1f %%)" % (f64 / stats["avg_ns"] * 1e-3, 100 * f64 / stats["avg_ns"] * 1e-3 / 78.6))
    o.write("\nper wave: VALU %.0f SALU %.0f LDS %.0f; SQ_LDS_BANK_CONFLICT / SQ_ACTIVE_INST_LDS = %s; VALU-active / wave cycles = %s; WAIT_ANY / wave cycles = %s\n"
            % (rec["valu_per_wave"], rec["salu_per_wave"], rec["lds_per_wave"], rec["lds_bank_conflict_over_active"], rec["valu_active_over_wave_cycles"], rec["wait_any_over_wave_cycles"]))
print(open(out + "/summary.txt").read())
PY
