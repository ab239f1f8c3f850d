#!/bin/bash
# rocprofv3 passes for the decode kernel (run on the GPU box through gpurun). Outputs under gpurun_out/.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_$1
mkdir -p $OUT
CMD="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --only ${2:-decompress}"
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc_$tag -o p -- $CMD > $OUT/pmc_$tag.log 2>&1
done
python - <<PY
import csv, glob, collections, os
out="$OUT"
for f in sorted(glob.glob(out+"/trace/**/*kernel_stats.csv", recursive=True)):
    print("== stats", f)
    print(open(f).read()[:1500])
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(out+"/pmc_*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k=r.get("Kernel_Name","")
        if "lz4" not in k: continue
        agg[k.split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items():
    print("== kernel", k)
    for c,vals in sorted(v.items()):
        print("  %-28s n=%d mean=%.4g" % (c, len(vals), sum(vals)/len(vals)))
PY
