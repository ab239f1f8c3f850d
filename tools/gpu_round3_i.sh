#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03j
mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q --maxfail=30 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; tail -25 $OUT/pytest_gpu.log
timeout 200 python tools/dec_shapes.py --variants 7,6 --shapes json:65536:256,log:4194304:256,log:16777216:1 > $OUT/pcd_shapes.log 2>&1; cat $OUT/pcd_shapes.log
for c in 3 4 5; do timeout 300 python bench.py --config $c --no-cpu-baseline > $OUT/bench_config$c.json 2> $OUT/bench_config$c.err; python - <<PY
import json
d=json.loads(open("$OUT/bench_config$c.json").read().strip().splitlines()[-1])
print($c, d["value"], d["unit"], d["ms_per_step"], d.get("parts_ms"), d["verified"][:80])
PY
done
timeout 300 python bench.py --config 5 --compress-mode exact --no-cpu-baseline > $OUT/bench_config5_exact.json 2>&1; tail -c 600 $OUT/bench_config5_exact.json
