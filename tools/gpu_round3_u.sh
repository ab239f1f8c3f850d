#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03u
mkdir -p $OUT
LZ4FLEX_LIB=lz4_flex_amd/build/variant_pprof/liblz4flex_amd.so timeout 300 python tools/dec_shapes.py --variants 7 --shapes json:65536:256,log:4194304:256,text:65536:160 > $OUT/pcd_prof.log 2>&1; grep -v amdgpu.ids $OUT/pcd_prof.log | cut -c1-420
