#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03final
mkdir -p $OUT
timeout 300 python tools/scalar_latency.py > $OUT/scalar_latency.log 2>&1; grep -v amdgpu.ids $OUT/scalar_latency.log
timeout 300 python tools/dec_shapes.py > $OUT/dec_shapes.log 2>&1; grep -v amdgpu.ids $OUT/dec_shapes.log
for c in 3 4 5; do timeout 600 python bench.py --config $c > $OUT/bench$c.json 2>$OUT/bench$c.err; python -c "
import json;d=json.load(open('$OUT/bench$c.json'));print($c, d['value'],d['ms_per_step'],d.get('parts_ms'))"; done
