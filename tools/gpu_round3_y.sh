#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03y
mkdir -p $OUT
V=lz4_flex_amd/build
timeout 300 python tools/enc_variants.py lz4_flex_amd/liblz4flex_amd.so $V/variant_plain/liblz4flex_amd.so > $OUT/enc_plain.log 2>&1; grep -v amdgpu.ids $OUT/enc_plain.log
LZ4FLEX_LIB=$V/variant_plain/liblz4flex_amd.so timeout 600 python -m pytest tests/test_gpu_wave_encoder.py -m gpu -q -x -p no:cacheprovider > $OUT/pytest_plain.log 2>&1; tail -3 $OUT/pytest_plain.log
