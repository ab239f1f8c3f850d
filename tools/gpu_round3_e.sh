#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03g
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_pcd.py tests/test_gpu_block.py -m gpu -q -p no:cacheprovider > $OUT/pytest_pcd.log 2>&1; tail -5 $OUT/pytest_pcd.log
LZ4FLEX_LIB=lz4_flex_amd/build/variant_pcdprof/liblz4flex_amd.so timeout 300 python tools/dec_shapes.py --variants 7 --shapes json:65536:256,text:65536:160,log:4194304:256,log:16777216:1,random:4194304:16 > $OUT/pcd_prof.log 2>&1
cat $OUT/pcd_prof.log
timeout 200 python tools/dec_shapes.py --variants 7,6,5 --shapes json:65536:256,json:65536:512,json:65536:1024,log:65536:1024,text:65536:1024,log:4194304:256,log:16777216:1 > $OUT/pcd_shapes.log 2>&1
cat $OUT/pcd_shapes.log
