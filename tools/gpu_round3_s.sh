#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03s
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "configs or fast_mode or bench_launch or pcd" > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
timeout 300 python bench.py --config 4 > $OUT/bench4.json 2>$OUT/bench4.err; tail -3 $OUT/bench4.err; python -c "
import json;d=json.load(open('$OUT/bench4.json'));print(d['value'],d['ms_per_step'],d['parts_ms'],d['verified'][:80])"
