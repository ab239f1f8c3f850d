#!/bin/bash
# Round profile on the GPU box (through gpurun): per BASELINE config the bench line, rocprofv3 kernel stats of the same command
# and the HBM traffic counters of its kernels (separate --pmc passes per counter, as MI355X_MICROARCH.md prescribes; never combined
# with trace domains other than --kernel-trace).  Outputs under gpurun_out/$1; tools/prof_summary.py turns them into the files
# profiles/ keeps (r03_kernel_stats_config*.csv, r03_summary_stats_and_counters.json, traffic.json).
# usage: tools/prof_round.sh <tag> [configs, default "2 3 4 5"]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$1
CONFIGS=${2:-"2 3 4 5"}
mkdir -p $OUT
for c in $CONFIGS; do
  extra=""
  [ "$c" = "5" ] && extra="--steps 3 --warmup 1"
  timeout 300 python bench.py --config $c > $OUT/bench_line_config$c.json 2> $OUT/bench_config$c.err      # (the bench line: default steps)
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_config$c -o t -- python bench.py --config $c --no-cpu-baseline $extra > $OUT/trace_config$c.log 2>&1
  if [ "$c" = "2" ]; then
    for only in compress decompress; do
      for set in FETCH_SIZE WRITE_SIZE; do
        timeout 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc_config2_${only}_$set -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --only $only > $OUT/pmc_config2_${only}_$set.log 2>&1
      done
    done
  else
    for set in FETCH_SIZE WRITE_SIZE; do
      timeout 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc_config${c}_both_$set -o p -- python bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pmc_config${c}_both_$set.log 2>&1
    done
  fi
done
python tools/prof_summary.py $OUT $1
# the raw traces are tens of MiB per pass: only the summaries travel back
rm -rf $OUT/trace_config* $OUT/pmc_config*
