#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_block.py tests/test_gpu_pcd.py tests/test_gpu_frame.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -6 | cut -c1-300
SH=json:65536:1,json:65536:64,json:65536:128,json:65536:256,text:65536:160,log:4194304:16,log:4194304:128,log:4194304:256,log:16777216:1,json:1048576:64,zeros:4194304:16
timeout 300 python tools/dec_shapes.py --variants 7 --shapes $SH 2>&1 | grep -v amdgpu.ids
timeout 300 python tools/scalar_latency.py 2>&1 | grep -v amdgpu.ids
