#!/bin/bash
# Instruction-mix and LDS counters of one decoder variant on the benchmark workload (GPU box, through gpurun): separate rocprofv3 --pmc
# passes with --kernel-trace only.  usage: tools/prof_seq.sh <tag> <decompress_variant> [data]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$1; V=${2:-13}; DATA=${3:-json}
mkdir -p $OUT
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z_0-9]*LDS[A-Z_0-9]*" | sort -u | tr '\n' ' ' > $OUT/lds_counters.txt
SETS=("SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL" "SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL" "SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD")
i=0
for set in "${SETS[@]}"; do
  timeout 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o p -- python tools/wave_bench.py --dec $V --reps 3 --data $DATA > $OUT/p$i.log 2>&1
  i=$((i+1))
done
python - <<PY
import csv, glob, collections, json, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "decompress" not in k: continue
        name = re.sub(r"^void\s+", "", re.sub(r"\)\s*\[.*$|\(.*$", "", k)).replace(" ", "")
        name = re.sub(r"^(?:\w+::)+", "", name)
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, v in agg.items():
    out[k] = {c: {"n": len(vals), "median": sorted(vals)[len(vals) // 2]} for c, vals in sorted(v.items())}
    print("== kernel", k)
    for c, d in out[k].items():
        print("  %-28s n=%d median=%.6g" % (c, d["n"], d["median"]))
json.dump(out, open("$OUT/insts.json", "w"), indent=1)
PY
cat $OUT/lds_counters.txt; echo
rm -rf $OUT/p[0-9]*/
