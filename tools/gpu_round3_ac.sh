#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03ac
mkdir -p $OUT
LZ4FLEX_LIB=lz4_flex_amd/build/variant_rl3/liblz4flex_amd.so timeout 900 python -m pytest tests/test_gpu_block.py tests/test_gpu_pcd.py -m gpu -q -x -p no:cacheprovider -k "synthetic or adversarial or large_blocks or marks or 7 or 8" > $OUT/pytest_rl3.log 2>&1; tail -5 $OUT/pytest_rl3.log | cut -c1-300
