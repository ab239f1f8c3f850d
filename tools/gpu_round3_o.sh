#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03o
mkdir -p $OUT
LZ4FLEX_LIB=lz4_flex_amd/build/variant_prof/liblz4flex_amd.so timeout 300 python tools/wave_bench.py --prof > $OUT/prof_json.log 2>&1; cat $OUT/prof_json.log
