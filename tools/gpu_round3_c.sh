#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03c
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_pcd.py tests/test_gpu_block.py -m gpu -q -p no:cacheprovider > $OUT/pytest_pcd.log 2>&1; tail -5 $OUT/pytest_pcd.log
LZ4FLEX_LIB=lz4_flex_amd/build/variant_pcdprof/liblz4flex_amd.so timeout 300 python tools/dec_shapes.py --variants 7 --shapes json:65536:1,json:65536:256,text:65536:160,log:65536:256,log:4194304:256,log:16777216:1,zeros:4194304:16,random:4194304:16 > $OUT/pcd_prof.log 2>&1
cat $OUT/pcd_prof.log
timeout 200 python tools/dec_shapes.py --variants 7,6 --shapes json:65536:256,json:65536:512,json:65536:1024,text:65536:160,log:4194304:256 > $OUT/pcd_shapes.log 2>&1
cat $OUT/pcd_shapes.log
for a in "2304 4 0" "2304 4 64" "4096 4 0" "4096 4 8" "1024 4 16"; do timeout 100 python tools/split_diag.py $a; done > $OUT/split_diag.log 2>&1
cat $OUT/split_diag.log
