#!/bin/bash
# Round profile on the GPU box (through gpurun): the bench line, rocprofv3 kernel stats of the default bench command and the
# HBM traffic counters of both kernels (separate --pmc passes per counter, as MI355X_MICROARCH.md prescribes; never combined
# with trace domains other than --kernel-trace).  Outputs under gpurun_out/$1; summary.json is what profiles/ keeps.
# usage: tools/prof_round.sh <tag>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$1
mkdir -p $OUT
timeout 240 python bench.py > $OUT/bench_line.json 2> $OUT/bench.err
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python bench.py --no-cpu-baseline > $OUT/trace.log 2>&1
for only in compress decompress; do
  for set in FETCH_SIZE WRITE_SIZE; do
    timeout 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc_${only}_$set -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --only $only > $OUT/pmc_${only}_$set.log 2>&1
  done
done
python - <<PY
import csv, glob, collections, json, re
out = "$OUT"
summary = {"tag": "$1", "kernel_stats": [], "counters": {}}
for f in sorted(glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "lz4" in r.get("Name", ""):
            summary["kernel_stats"].append({"kernel": re.sub(r"\(.*", "", r["Name"])[:90], "calls": int(r["Calls"]),
                                            "avg_ns": float(r["AverageNs"]), "min_ns": float(r["MinNs"]), "max_ns": float(r["MaxNs"]),
                                            "pct": float(r["Percentage"])})
    print("== stats", f)
    print(open(f).read()[:2500])
for only in ("compress", "decompress"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in sorted(glob.glob(out + "/pmc_%s_*/**/*counter_collection.csv" % only, recursive=True)):
        for r in csv.DictReader(open(f)):
            k = r.get("Kernel_Name", "")
            if "lz4" not in k and "CompareEq" not in k:
                continue
            agg[re.sub(r"\(.*", "", k)[-110:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print("== %s pass, kernel %s" % (only, k))
        for c, vals in sorted(v.items()):
            vals = sorted(vals)
            print("  %-14s n=%d median=%.6g max=%.6g" % (c, len(vals), vals[len(vals) // 2], vals[-1]))
            summary["counters"].setdefault(only, {}).setdefault(k, {})[c] = {"n": len(vals), "median_kib": vals[len(vals) // 2], "max_kib": vals[-1]}
json.dump(summary, open(out + "/summary.json", "w"), indent=1)
PY
cat $OUT/bench_line.json
