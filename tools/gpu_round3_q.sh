#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03q
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "frame or fast_mode or configs or cli" > $OUT/pytest_frame.log 2>&1; tail -4 $OUT/pytest_frame.log
timeout 300 python tools/linked_timing.py 64 > $OUT/linked.log 2>&1; cat $OUT/linked.log
timeout 300 python tools/linked_timing.py 1024 > $OUT/linked1024.log 2>&1; cat $OUT/linked1024.log
timeout 300 python bench.py --config 5 > $OUT/bench5.json 2>$OUT/bench5.err; cut -c1-200 $OUT/bench5.json
