cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4b
timeout 120 tools/ubench_lds.bin > gpurun_out/r4b/ubench_lds.txt 2>&1
grep -E "16 B/lane|READ  waves/CU 8 active lanes 64" gpurun_out/r4b/ubench_lds.txt | head -60
bash tools/prof_replay.sh r4b/counters lz4_flex_amd/build/variant_r_idlein/liblz4flex_amd.so 2>&1 | tail -45
