#!/bin/bash
# What a round's GPU session runs through gpurun (everything lands under gpurun_out/<tag>):  gpurun -- 'bash tools/gpu_check.sh <tag>'
#   the GPU test suite, the scalar calls' latency, every decoder on every batch shape, a Linked frame's timing, and the round
#   profile (tools/prof_round.sh: bench lines, rocprofv3 kernel statistics and HBM counters of configs 2-5).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-check}
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q --maxfail=30 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; tail -4 $OUT/pytest_gpu.log
timeout 300 python tools/scalar_latency.py > $OUT/scalar_latency.log 2>&1; grep -v amdgpu.ids $OUT/scalar_latency.log
timeout 300 python tools/dec_shapes.py > $OUT/dec_shapes.log 2>&1; grep -v amdgpu.ids $OUT/dec_shapes.log
timeout 300 python tools/linked_timing.py 64 > $OUT/linked.log 2>&1; grep -v amdgpu.ids $OUT/linked.log
bash tools/prof_round.sh ${1:-check}_prof "2 3 4 5" > $OUT/prof_round.log 2>&1
grep "kernel stats" -A5 $OUT/prof_round.log | head -40
