#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03r
mkdir -p $OUT
LZ4FLEX_LIB=lz4_flex_amd/build/variant_pprof/liblz4flex_amd.so timeout 300 python tools/dec_shapes.py --variants 7 --shapes json:65536:256,log:4194304:256,log:16777216:1 > $OUT/pcd_prof.log 2>&1; grep -v amdgpu.ids $OUT/pcd_prof.log
timeout 300 python bench.py --config 4 > $OUT/bench4.json 2>$OUT/bench4.err; python -c "
import json;d=json.load(open('$OUT/bench4.json'));print(d['value'],d['ms_per_step'],d['parts_ms'])"
