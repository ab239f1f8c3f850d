#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03m
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q --maxfail=30 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; tail -6 $OUT/pytest_gpu.log
timeout 300 python tools/scalar_latency.py > $OUT/scalar_latency.log 2>&1; cat $OUT/scalar_latency.log
timeout 300 python tools/dec_shapes.py > $OUT/dec_shapes.log 2>&1; cat $OUT/dec_shapes.log
bash tools/prof_round.sh r03m_prof "2 3 4 5" > $OUT/prof_round.log 2>&1
grep "kernel stats" -A4 $OUT/prof_round.log | head -40
