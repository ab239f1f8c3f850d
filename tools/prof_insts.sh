#!/bin/bash
# Instruction-mix counters of both hot kernels on the benchmark workload (GPU box, through gpurun): SQ_* counters in separate
# rocprofv3 --pmc passes with --kernel-trace only, once with bench.py --only compress and once with --only decompress.
# Writes $OUT/insts_compress.json and $OUT/insts_decompress.json ({kernel: {counter: {n, median}}}; copy to profiles/rNN_insts_*.json).
# usage: tools/prof_insts.sh <tag> [counter sets...]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$1; shift
mkdir -p $OUT
SETS=("$@")
if [ ${#SETS[@]} -eq 0 ]; then SETS=("SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH"); fi
for only in compress decompress; do
  i=0
  for set in "${SETS[@]}"; do
    timeout 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/${only}_p$i -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify --only $only > $OUT/${only}_p$i.log 2>&1
    i=$((i+1))
  done
  python - <<PY
import csv, glob, collections, json, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("$OUT/${only}_p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "lz4" not in k: continue
        name = re.sub(r"^void\s+", "", re.sub(r"\)\s*\[.*$|\(.*$", "", k)).replace(" ", "")      # whole name, template arguments included: never cut from the left
        name = re.sub(r"^(?:\w+::)+", "", name)                                             # ... without the namespaces in front of the kernel's own name
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"_how": "rocprofv3 --pmc <two counters> --kernel-trace per pass (tools/prof_insts.sh), bench.py --only $only --steps 3 --warmup 1: "
               "16 384 JSON blocks of 64 KiB per launch; per counter the median over the kernel's launches; SQ_* instruction counters are per wavefront"}
for k, v in agg.items():
    if max(len(vals) for vals in v.values()) < 3: continue          # the untimed pass of the OTHER kernel (bench.py --only runs it once to have data)
    out[k] = {c: {"n": len(vals), "median": sorted(vals)[len(vals) // 2]} for c, vals in sorted(v.items())}
    print("== kernel", k)
    for c, d in out[k].items():
        print("  %-28s n=%d median=%.6g" % (c, d["n"], d["median"]))
json.dump(out, open("$OUT/insts_$only.json", "w"), indent=1)
PY
  rm -rf $OUT/${only}_p[0-9]*/
done
