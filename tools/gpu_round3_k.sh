#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03k
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_bench_launch.py -m gpu -q -p no:cacheprovider > $OUT/pytest.log 2>&1; tail -15 $OUT/pytest.log
bash tools/prof_round.sh r03k_prof "2 3 4 5" > $OUT/prof_round.log 2>&1
tail -60 $OUT/prof_round.log
