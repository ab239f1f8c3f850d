#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03n
mkdir -p $OUT
hipcc --offload-arch=gfx950 -O3 tools/ubench_lds.hip -o /tmp/ubench_lds 2>/dev/null && timeout 120 /tmp/ubench_lds > $OUT/ubench_lds.log 2>&1
grep "lanes 64\|lanes 16" $OUT/ubench_lds.log
V=lz4_flex_amd/build
LIBS="lz4_flex_amd/liblz4flex_amd.so $V/variant_ar1/liblz4flex_amd.so $V/variant_arn/liblz4flex_amd.so $V/variant_nolen/liblz4flex_amd.so"
timeout 300 python tools/enc_variants.py $LIBS > $OUT/enc_json.log 2>&1; cat $OUT/enc_json.log
timeout 300 python tools/enc_variants.py --data text $LIBS > $OUT/enc_text.log 2>&1; grep bench $OUT/enc_text.log
