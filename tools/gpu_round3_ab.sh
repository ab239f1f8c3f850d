#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
V=lz4_flex_amd/build
for v in none3 nopl; do echo "== $v"; LZ4FLEX_LIB=$V/variant_$v/liblz4flex_amd.so timeout 200 python tools/wave_bench.py --dec 4 2>&1 | grep "decompress_ms" | sed 's/.*round_trip_ok/round_trip_ok/'; done
