#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03ac
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_block.py -m gpu -q -x -p no:cacheprovider -k synthetic > $OUT/pytest.log 2>&1; tail -12 $OUT/pytest.log | cut -c1-300
