#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03x
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_wave_encoder.py tests/test_gpu_fast_mode.py tests/test_gpu_configs.py -m gpu -q -x -p no:cacheprovider > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
echo "== without the second launch (expected: the give-up case fails with status 66)"
LZ4FLEX_LIB=lz4_flex_amd/build/variant_noredo/liblz4flex_amd.so timeout 600 python -m pytest tests/test_gpu_wave_encoder.py -m gpu -q -p no:cacheprovider -k window_mode > $OUT/pytest_noredo.log 2>&1; tail -8 $OUT/pytest_noredo.log | cut -c1-200
timeout 300 python bench.py --config 4 > $OUT/bench4.json 2>$OUT/bench4.err; python -c "
import json;d=json.load(open('$OUT/bench4.json'));print(d['value'],d['ms_per_step'],d['parts_ms'])"
