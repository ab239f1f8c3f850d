#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
SH=json:65536:256,json:65536:512,text:65536:160,log:4194304:256,log:16777216:1,json:1048576:64,zeros:4194304:16,log:65536:1024,text:65536:1024
echo "== default (min 20)"; timeout 300 python tools/dec_shapes.py --variants 7 --shapes $SH 2>&1 | grep -v amdgpu.ids
for v in rm28 rm12; do echo "== $v"; LZ4FLEX_LIB=lz4_flex_amd/build/variant_$v/liblz4flex_amd.so timeout 300 python tools/dec_shapes.py --variants 7 --shapes $SH 2>&1 | grep -v amdgpu.ids; done
timeout 900 python -m pytest tests/test_gpu_block.py tests/test_gpu_pcd.py -m gpu -q -x -p no:cacheprovider -k "synthetic or adversarial or large_blocks or marks" 2>&1 | tail -2
