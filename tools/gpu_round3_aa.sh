#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03aa
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_pcd.py tests/test_gpu_block.py tests/test_gpu_frame.py -m gpu -q -x -p no:cacheprovider > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
SH=json:65536:256,json:65536:512,text:65536:160,log:4194304:256,log:16777216:1,json:1048576:64,zeros:4194304:16,random:65536:256
timeout 300 python tools/dec_shapes.py --variants 7 --shapes $SH 2>&1 | grep -v amdgpu.ids
LZ4FLEX_LIB=lz4_flex_amd/build/variant_pprof/liblz4flex_amd.so timeout 300 python tools/dec_shapes.py --variants 7 --shapes log:4194304:256 2>&1 | grep -v amdgpu.ids | cut -c1-330
