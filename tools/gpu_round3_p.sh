#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03p
mkdir -p $OUT
timeout 300 python tools/linked_timing.py 64 > $OUT/linked.log 2>&1; cat $OUT/linked.log
timeout 300 python tools/linked_timing.py 1024 > $OUT/linked1024.log 2>&1; cat $OUT/linked1024.log
