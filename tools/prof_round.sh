#!/bin/bash
# Round profile on the GPU box (through gpurun): the bench line, rocprofv3 kernel stats of the default bench command and
# the HBM traffic counters of one kernel (separate --pmc passes, as MI355X_MICROARCH.md prescribes).  Outputs under gpurun_out/$1.
# usage: tools/prof_round.sh <tag> [decompress|compress]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$1
ONLY=${2:-decompress}
mkdir -p $OUT
timeout 170 python bench.py > $OUT/bench_line.json 2> $OUT/bench.err
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python bench.py --no-cpu-baseline > $OUT/trace.log 2>&1
for set in FETCH_SIZE WRITE_SIZE; do
  timeout 100 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc_$set -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --only $ONLY > $OUT/pmc_$set.log 2>&1
done
python - <<PY
import csv, glob, collections
out="$OUT"
for f in sorted(glob.glob(out+"/trace/**/*kernel_stats.csv", recursive=True)):
    print("== stats", f)
    print(open(f).read()[:1800])
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(out+"/pmc_*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k=r.get("Kernel_Name","")
        if "lz4" not in k and "CompareEq" not in k: continue
        agg[k.split("(lz4flex_dev::")[0][-80:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items():
    print("== kernel", k)
    for c,vals in sorted(v.items()):
        print("  %-28s n=%d mean=%.6g" % (c, len(vals), sum(vals)/len(vals)))
PY
cat $OUT/bench_line.json
