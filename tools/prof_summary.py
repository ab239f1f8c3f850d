#!/usr/bin/env python3
"""Summarise what tools/prof_round.sh collected: per config the lz4 kernels' rocprofv3 statistics (calls, average / min / max ns)
and the FETCH_SIZE / WRITE_SIZE medians per kernel -> <out>/summary.json, and print them.  Counter units are KiB; HBM bytes per
launch = FETCH_SIZE x 2 x 1024 + WRITE_SIZE x 1024 (MI355X_MICROARCH.md: gfx950 tallies 128-byte read requests at 64 bytes),
calibrated in the same pass on torch's CompareEq kernel (bench.py's verification: reads 2 x the buffer, writes 1/8... see the
printed figures)."""
import collections
import csv
import glob
import json
import os
import re
import sys


def short(name):
    return re.sub(r"\(.*", "", name).strip()


def key_of(name):
    m = re.search(r"(lz4\w+kernel|lz4flex\w+kernel|xxh32\w*kernel)", name)
    return m.group(1) if m else short(name)[:60]


def main():
    out, tag = sys.argv[1], sys.argv[2]
    summary = {"tag": tag, "configs": {}}
    for tdir in sorted(glob.glob(os.path.join(out, "trace_config*"))):
        if not os.path.isdir(tdir):
            continue
        cfg = re.search(r"config(\d+)", tdir).group(1)
        rows = []
        for f in sorted(glob.glob(tdir + "/**/*kernel_stats.csv", recursive=True)):
            for r in csv.DictReader(open(f)):
                if "lz4" in r.get("Name", "") or "xxh32" in r.get("Name", ""):
                    rows.append({"kernel": short(r["Name"])[:140], "kernel_key": key_of(r["Name"]), "calls": int(r["Calls"]),
                                 "avg_ns": float(r["AverageNs"]), "min_ns": float(r["MinNs"]), "max_ns": float(r["MaxNs"]),
                                 "total_ns": float(r["TotalDurationNs"]), "pct": float(r["Percentage"])})
            # a copy of the csv itself for profiles/
            with open(os.path.join(out, "kernel_stats_config%s.csv" % cfg), "w") as g:
                g.write(open(f).read())
        # medians from the raw kernel trace (the stats file only has means)
        durs = collections.defaultdict(list)
        for f in sorted(glob.glob(tdir + "/**/*kernel_trace.csv", recursive=True)):
            for r in csv.DictReader(open(f)):
                n = r.get("Kernel_Name", "")
                if "lz4" in n or "xxh32" in n:
                    durs[key_of(n)].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
        for r in rows:
            d = sorted(durs.get(r["kernel_key"], []))
            if d:
                r["median_ns"] = d[len(d) // 2]
        summary["configs"].setdefault(cfg, {})["kernel_stats"] = rows
        print("== config %s kernel stats" % cfg)
        for r in sorted(rows, key=lambda r: -r["total_ns"]):
            print("  %-70s calls %4d avg %10.1f us median %10.1f us  %5.1f %%" % (r["kernel"][:70], r["calls"], r["avg_ns"] / 1e3, r.get("median_ns", 0) / 1e3, r["pct"]))
    for pdir in sorted(glob.glob(os.path.join(out, "pmc_config*"))):
        if not os.path.isdir(pdir):
            continue
        m = re.search(r"pmc_config(\d+)_(\w+?)_(FETCH_SIZE|WRITE_SIZE)$", pdir)
        if not m:
            continue
        cfg, only, counter = m.groups()
        agg = collections.defaultdict(list)
        for f in sorted(glob.glob(pdir + "/**/*counter_collection.csv", recursive=True)):
            for r in csv.DictReader(open(f)):
                k = r.get("Kernel_Name", "")
                if "lz4" not in k and "CompareEq" not in k and "xxh32" not in k:
                    continue
                if r["Counter_Name"] != counter:
                    continue
                agg[key_of(k) if "lz4" in k or "xxh32" in k else "CompareEq (calibration)"].append(float(r["Counter_Value"]))
        for k, vals in agg.items():
            vals = sorted(vals)
            e = summary["configs"].setdefault(cfg, {}).setdefault("counters", {}).setdefault(k, {})
            e[counter + ("" if only == "both" else "@" + only)] = {"n": len(vals), "median_kib": vals[len(vals) // 2], "max_kib": vals[-1]}
            print("== config %s %-10s %-45s %-11s n=%3d median %.6g KiB max %.6g KiB" % (cfg, only, k[:45], counter, len(vals), vals[len(vals) // 2], vals[-1]))
    for cfg in summary["configs"]:
        p = os.path.join(out, "bench_line_config%s.json" % cfg)
        if os.path.exists(p):
            try:
                summary["configs"][cfg]["bench_line"] = json.loads([ln for ln in open(p).read().splitlines() if ln.startswith("{")][-1])
            except Exception as e:
                summary["configs"][cfg]["bench_line_error"] = repr(e)
    json.dump(summary, open(os.path.join(out, "summary.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
