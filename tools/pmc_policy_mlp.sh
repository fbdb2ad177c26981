#!/bin/bash
# usage: tools/pmc_policy_mlp.sh [library.so]   (GPU box, from the repo root)
# rocprofv3 --pmc passes (counters only, one group per pass) of tools/policy_mlp_check.py: matrix-core busy cycles,
# instruction counts and wait cycles of policy_mlp_kernel at 64 x 4096 rows -> gpurun_out/pmc_policy_mlp/summary.txt
LIBV=${1:-}
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/pmc_policy_mlp
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export MLP_DENSE_ONLY=1   # dense rows: every k-step of every pass
run() { # name counters...
  n=$1; shift
  timeout -k 10 200 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$n -- python3 $ROOT/tools/policy_mlp_check.py $LIBV > $OUT/$n.log 2>&1
}
run a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA && \
run b SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM && \
run c SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_VMEM GRBM_GUI_ACTIVE SQ_CYCLES
python3 - > $OUT/summary.txt <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("$OUT/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "policy_mlp_kernel" in r["Kernel_Name"] and int(r["Grid_Size"]) >= 256 * 512:
            a = agg[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
c = {k: v / n for k, (v, n) in agg.items()}
print("policy_mlp_kernel<7, 8>, 262144 dense rows (no counts: 1600 MFMAs per wave), per launch (mean over %d launches; quad-cycle counters x 4):" % agg["SQ_WAVES"][1])
w = c["SQ_WAVES"]
print("  waves %d, MFMA instructions %.0f per wave, other VALU %.0f, LDS %.0f, VMEM reads %.0f, SALU %.0f per wave" % (
    w, c["SQ_INSTS_MFMA"] / w, (c["SQ_INSTS_VALU"] - c["SQ_INSTS_MFMA"]) / w, c["SQ_INSTS_LDS"] / w, c["SQ_INSTS_VMEM_RD"] / w, c["SQ_INSTS_SALU"] / w))
life = 4 * c["SQ_WAVE_CYCLES"] / w
print("  wave life %.0f cycles (mean), of which in s_waitcnt %.0f, waiting for issue %.0f" % (life, 4 * c["SQ_WAIT_ANY"] / w, 4 * c["SQ_WAIT_INST_ANY"] / w))
simds = 1024
print("  matrix pipe busy %.0f cycles per SIMD = %.2f of the mean wave life (SQ_VALU_MFMA_BUSY_CYCLES = 32 x MFMAs: %.0f)" % (
    c["SQ_VALU_MFMA_BUSY_CYCLES"] / simds, c["SQ_VALU_MFMA_BUSY_CYCLES"] / simds / life, 32 * c["SQ_INSTS_MFMA"]))
print("  bf16 MFMA ops %.3e (x 512 = %.1f GFLOP issued), LDS bank conflicts %.0f, LDS busy %.0f cycles per CU" % (
    c["SQ_INSTS_VALU_MFMA_MOPS_BF16"], c["SQ_INSTS_VALU_MFMA_MOPS_BF16"] * 512 / 1e9, c["SQ_LDS_BANK_CONFLICT"], 4 * c["SQ_LDS_IDX_ACTIVE"] / 256))
for k in sorted(c): print(f"    {k:34s} {c[k]:16.0f}")
PY
cat $OUT/summary.txt
