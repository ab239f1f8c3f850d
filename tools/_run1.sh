cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4a
{
for v in onlylds onlyload onlystore nothing; do
  echo "== variant $v"; timeout 200 python tools/replay_bench.py --reps 3 --lib lz4_flex_amd/build/variant_r_$v/liblz4flex_amd.so
done
} > gpurun_out/r4a/replay.log 2>&1
grep -v amdgpu.ids gpurun_out/r4a/replay.log | tail -40
