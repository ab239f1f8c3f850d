#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03b
mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_pcd.py -m gpu -q -p no:cacheprovider > $OUT/pytest_pcd.log 2>&1; tail -5 $OUT/pytest_pcd.log
LZ4FLEX_LIB=lz4_flex_amd/build/variant_pcdprof/liblz4flex_amd.so timeout 300 python tools/dec_shapes.py --variants 7 --shapes json:65536:1,json:65536:256,text:65536:160,log:65536:256,log:4194304:256,log:16777216:1,zeros:4194304:16,random:4194304:16 > $OUT/pcd_prof.log 2>&1
cat $OUT/pcd_prof.log
for n in 1024 2304 4096 8192; do timeout 120 python tools/wave_bench.py --dec 4 --blocks $n --reps 2 2>&1 | tail -1; done > $OUT/split_check.log 2>&1
cat $OUT/split_check.log
