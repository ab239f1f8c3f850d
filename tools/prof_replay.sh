#!/bin/bash
# Hardware counters of the replay kernel (tools/replay_bench.py) on the GPU box: separate rocprofv3 --pmc passes with
# --kernel-trace only (never combined with other trace domains).  usage: tools/prof_replay.sh <tag> [lib]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$1
LIBARG=""
[ -n "$2" ] && LIBARG="--lib $2"
mkdir -p $OUT
SETS=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA"
      "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_UNALIGNED_STALL"
      "SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum"
      "TCC_HIT_sum TCC_MISS_sum" "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TCP_GATE_EN1_sum TCP_TA_DATA_STALL_CYCLES_sum" "FETCH_SIZE" "WRITE_SIZE")
i=0
for set in "${SETS[@]}"; do
  timeout 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o p -- python tools/replay_bench.py --reps 2 $LIBARG > $OUT/p$i.log 2>&1
  i=$((i+1))
done
python - <<PY
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "replay" not in k and "plan" not in k: continue
        agg[re.sub(r"\(.*", "", k)[-60:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print("== kernel", k)
    for c, vals in sorted(v.items()):
        vals = sorted(vals)
        print("  %-32s n=%d median=%.6g" % (c, len(vals), vals[len(vals) // 2]))
PY
rm -rf $OUT/p[0-9]*/
