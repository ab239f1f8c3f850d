#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03t
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_pcd.py tests/test_gpu_block.py tests/test_gpu_frame.py -m gpu -q -x -p no:cacheprovider > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
timeout 300 python tools/dec_shapes.py --variants 7 --shapes json:65536:256,json:65536:512,text:65536:160,log:4194304:256,log:16777216:1,zeros:4194304:16,random:65536:256 > $OUT/pcd_shapes.log 2>&1; grep -v amdgpu.ids $OUT/pcd_shapes.log
