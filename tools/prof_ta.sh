#!/bin/bash
# rocprofv3 TA/TCP counter passes (run on the GPU box through gpurun). usage: prof_ta.sh <tag> <bench args...>
# NOTE: the second counter set (TA_*_STALLED_BY_TC) aborted rocprofv3 and hung until the time limit in round 1;
# every pass is wrapped in `timeout`, but only the first set (TA_TA_BUSY_sum TA_BUSY_avr) has produced data so far.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
OUT=gpurun_out/ta_$tag
mkdir -p $OUT
CMD="python bench.py --steps 2 --warmup 1 --no-cpu-baseline $*"
k=0
for set in "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TOTAL_READ_sum" "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum"; do
  k=$((k+1))
  timeout 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$k -o p -- $CMD > $OUT/p$k.log 2>&1
done
python - <<PY
import csv, glob, collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k=r.get("Kernel_Name","")
        if "lz4" not in k: continue
        agg[k.split("(")[0][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items():
    print("== kernel", k)
    for c,vals in sorted(v.items()):
        print("  %-40s n=%d mean=%.5g" % (c, len(vals), sum(vals)/len(vals)))
PY
