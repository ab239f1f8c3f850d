#!/bin/bash
# round 3, GPU call A: the whole GPU test suite, decoder shape matrix, default bench + other configs
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03a
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -30 $OUT/pytest_gpu.log
timeout 400 python tools/dec_shapes.py > $OUT/dec_shapes.log 2>&1
cat $OUT/dec_shapes.log
timeout 300 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -c 3000 $OUT/bench_default.json
for c in 3 4 5; do
  timeout 300 python bench.py --config $c > $OUT/bench_config$c.json 2> $OUT/bench_config$c.err
  tail -c 1500 $OUT/bench_config$c.json
done
