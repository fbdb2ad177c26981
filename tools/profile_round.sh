#!/bin/bash
# usage: tools/profile_round.sh <rNN>      (GPU box, from the repo root; ~10 min)
# Everything profiles/<rNN>/ holds for a round, in one go:
#   pmc_record (kernel-trace stats + separate --pmc passes) of the headline command, config 5 and config 2
#   kernel-trace stats of cache-cold launches only (bench.py --cold-all) and of the rollout loop
#   kernel times of every benchmark shape, warm and cold; the memory-only kernel, warm and cold; phase stamps
set -u
R=$1
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/profiles_$R
mkdir -p $OUT
cd $ROOT
bash tools/pmc_record.sh cfg3 --no-rollout --no-cold > /dev/null 2>&1
bash tools/pmc_record.sh cfg5 --no-rollout --no-cold --envs 1024 --drones 256 --buildings 50 --map 100 100 10 > /dev/null 2>&1
bash tools/pmc_record.sh cfg2 --no-rollout --no-cold --envs 256 --drones 16 --map 20 20 8 > /dev/null 2>&1
for t in cfg3 cfg5 cfg2; do
  for f in summary.txt record.json kernel_stats.csv; do cp gpurun_out/pmc_record_$t/$f $OUT/${t}_pmc_$f 2>/dev/null; done
  mv $OUT/${t}_pmc_kernel_stats.csv $OUT/${t}_kernel_stats.csv 2>/dev/null
done
( cd /tmp && export TMPDIR=/tmp
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cold_trace -- python3 $ROOT/bench.py --no-cpu-baseline --no-rollout --no-cold --cold-all --prewarm 20 --steps 200 > $OUT/cold_trace.log 2>&1
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rollout_trace -- python3 $ROOT/tools/bench_rollout.py --amp --steps 16 --no-update --no-tune > $OUT/rollout_trace.log 2>&1 )
( cd /tmp && export TMPDIR=/tmp
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rollout_rnn_trace -- python3 $ROOT/tools/bench_rollout.py --amp --steps 16 --no-update --no-tune --policy rnn > $OUT/rollout_rnn_trace.log 2>&1 )
for k in cold rollout rollout_rnn; do f=$(find $OUT/${k}_trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -16 $f > $OUT/${k}_kernel_stats.csv; done
rm -rf $OUT/cold_trace $OUT/rollout_trace $OUT/rollout_rnn_trace
bash tools/bench_all.sh $R > /dev/null 2>&1; cp gpurun_out/bench_all_$R.txt $OUT/bench_all_shapes.txt
python bench.py > $OUT/bench_cfg3.json 2> $OUT/bench_cfg3.err
python tools/bench_rollout.py --amp --steps 16 > $OUT/rollout_bench.json 2>/dev/null
python tools/rollout_profile.py > $OUT/rollout_step_kernels.txt 2>/dev/null
python tools/rollout_profile.py --policy rnn > $OUT/rollout_step_kernels_rnn.txt 2>/dev/null
tools/ubench/hbm_rate warm > $OUT/hbm_rate_ubench.txt 2>&1; tools/ubench/hbm_rate >> $OUT/hbm_rate_ubench.txt 2>&1
python tools/stamps.py > $OUT/stamps_cfg3_warm.txt 2>&1; STAMPS_COLD=rw python tools/stamps.py > $OUT/stamps_cfg3_cold.txt 2>&1
python tools/bench_aux.py > $OUT/aux_kernels.txt 2>&1
python tools/policy_mlp_check.py > $OUT/policy_mlp_check.txt 2>/dev/null
python tools/update_profile.py --no-split > $OUT/update_kernels_before_split.txt 2>/dev/null
python tools/update_profile.py > $OUT/update_kernels.txt 2>/dev/null
bash tools/pmc_policy_mlp.sh > /dev/null 2>&1; cp gpurun_out/pmc_policy_mlp/summary.txt $OUT/policy_mlp_pmc_summary.txt
ls $OUT
