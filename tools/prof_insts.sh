#!/bin/bash
# Instruction-mix counters of the encoder kernel on the benchmark workload (GPU box, through gpurun): SQ_* counters in
# separate rocprofv3 --pmc passes with --kernel-trace only.  usage: tools/prof_insts.sh <tag> [counter sets...]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$1; shift
mkdir -p $OUT
SETS=("$@")
if [ ${#SETS[@]} -eq 0 ]; then SETS=("SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" "SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_THREAD_CYCLES_VALU"); fi
i=0
for set in "${SETS[@]}"; do
  timeout 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-verify --only compress > $OUT/p$i.log 2>&1
  i=$((i+1))
done
python - <<PY
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "lz4" not in k: continue
        agg[re.sub(r"\(.*", "", k)[-60:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print("== kernel", k)
    for c, vals in sorted(v.items()):
        vals = sorted(vals)
        print("  %-28s n=%d median=%.6g" % (c, len(vals), vals[len(vals) // 2]))
PY
tail -3 $OUT/p0.log
