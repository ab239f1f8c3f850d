#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03final
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_wave_encoder.py tests/test_gpu_fast_mode.py tests/test_gpu_configs.py tests/test_gpu_bench_launch.py -m gpu -q -x -p no:cacheprovider > $OUT/pytest_enc.log 2>&1; tail -3 $OUT/pytest_enc.log
bash tools/prof_round.sh r03final_prof "2 3 4 5" > $OUT/prof_round.log 2>&1
grep "kernel stats" -A5 $OUT/prof_round.log | head -40
for c in 2 3 4 5; do python -c "
import json;d=json.load(open('gpurun_out/r03final_prof/bench_line_config$c.json'));print($c, d['value'],d['ms_per_step'],d.get('parts_ms'), d.get('roofline',{}).get('frac'))"; done
