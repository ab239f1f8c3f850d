#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03v
mkdir -p $OUT
for v in p64 p256; do
echo "== $v"; LZ4FLEX_LIB=lz4_flex_amd/build/variant_$v/liblz4flex_amd.so timeout 200 python tools/dec_shapes.py --variants 7 --shapes json:65536:256,text:65536:160,log:4194304:256,log:16777216:1 2>&1 | grep -v amdgpu.ids
done > $OUT/pcd_parts.log 2>&1
cat $OUT/pcd_parts.log
