#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03final
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q --maxfail=30 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; tail -4 $OUT/pytest_gpu.log
timeout 300 python tools/scalar_latency.py > $OUT/scalar_latency.log 2>&1; grep -v amdgpu.ids $OUT/scalar_latency.log
timeout 300 python tools/dec_shapes.py > $OUT/dec_shapes.log 2>&1; grep -v amdgpu.ids $OUT/dec_shapes.log
timeout 300 python tools/linked_timing.py 64 > $OUT/linked.log 2>&1; grep -v amdgpu.ids $OUT/linked.log
bash tools/prof_round.sh r03final_prof "2 3 4 5" > $OUT/prof_round.log 2>&1
grep "kernel stats" -A4 $OUT/prof_round.log | head -40
for c in 2 3 4 5; do python -c "
import json;d=json.load(open('gpurun_out/r03final_prof/bench_line_config$c.json'));print($c, d['value'],d['ms_per_step'],d.get('parts_ms'), d.get('roofline',{}).get('frac'))"; done
