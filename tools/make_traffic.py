#!/usr/bin/env python3
"""profiles/traffic.json + the round's profile files from a tools/prof_round.sh summary:  python tools/make_traffic.py gpurun_out/<tag> <round>"""
import json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, rnd = sys.argv[1], int(sys.argv[2])
S = json.load(open(os.path.join(src, "summary.json")))
P = os.path.join(ROOT, "profiles")
pre = "r%02d" % rnd
shutil.copy(os.path.join(src, "summary.json"), os.path.join(P, pre + "_summary_stats_and_counters.json"))
for cfg in S["configs"]:
    for name, dst in (("kernel_stats_config%s.csv" % cfg, pre + "_kernel_stats_config%s.csv" % cfg), ("bench_line_config%s.json" % cfg, pre + "_bench_line_config%s.json" % cfg)):
        p = os.path.join(src, name)
        if os.path.exists(p):
            shutil.copy(p, os.path.join(P, dst))
old = json.load(open(os.path.join(P, "traffic.json")))
out = {"_how": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (tools/prof_round.sh; --kernel-trace only beside the counter). Units are KiB; "
               "medians over the launches of a pass. Corrected as MI355X_MICROARCH.md prescribes: FETCH_SIZE x2 (gfx950 tallies 128-B requests at 64 B), "
               "calibrated in the same pass on torch's CompareEq kernel over the 1 GiB buffers (FETCH_SIZE 1 048 6xx KiB for 2 GiB read, WRITE_SIZE exact). "
               "The counters sit on the L2's fabric side: Infinity-Cache hits are counted (an upper bound of HBM bytes). kernel_key = the kernel's plain "
               "name (bench.py matches on it); entries of earlier rounds are kept under 'history'.",
       "history": {k: v for k, v in old.items() if not k.startswith("_") and k != "history"}}
out["history"].update(old.get("history", {}))


def entry(cfg, key, kernel_key, blocks, alg, counters, note):
    c = S["configs"][str(cfg)]["counters"][kernel_key]
    f = c.get("FETCH_SIZE@" + counters, c.get("FETCH_SIZE"))["median_kib"]
    w = c.get("WRITE_SIZE@" + counters, c.get("WRITE_SIZE"))["median_kib"]
    st = [r for r in S["configs"][str(cfg)]["kernel_stats"] if r["kernel_key"] == kernel_key]
    out[key] = {"kernel_key": kernel_key, "kernel": st[0]["kernel"] if st else kernel_key, "round": rnd, "config": cfg, "blocks": blocks,
                "fetch_size_kib": f, "write_size_kib": w, "hbm_bytes_per_launch": int(f * 2048 + w * 1024), "algorithmic_bytes_per_launch": alg,
                "avg_us": st[0]["avg_ns"] / 1e3 if st else None, "median_us": st[0].get("median_ns", 0) / 1e3 if st else None,
                "source": "%s -> profiles/%s_summary_stats_and_counters.json" % (src, pre), "note": note}


b2 = S["configs"]["2"]["bench_line"]
alg2 = int(2**30 * (1 + b2["ratio"]))
entry(2, "compress_fast", "lz4_compress_wave_kernel", 16384, alg2, "compress",
      "beyond input (1 GiB read) and output (0.23 GiB written): the indexer writes 2 B of cand[] per input byte and the workers read it back (2 + 2 GiB), "
      "segment bodies go through the workspace (2 x 0.23 GiB, exact-size stores since round 6); the 166 MiB workspace exceeds the 32 MiB of L2, so this traffic crosses the fabric")
entry(2, "decompress", "lz4_decompress_split_kernel", 16384, alg2, "decompress",
      "reads: the compressed stream by the parser and again (literal pieces) by the copiers, far match pieces from the written-back output; every copier "
      "lane loads 16 B per step whether its piece needs them or not; the x2 correction is an upper bound for scattered loads")
b4 = S["configs"]["4"]["bench_line"]
alg4 = int(2**30 * (1 + b4["ratio"]))
entry(4, "config4_compress", "lz4_compress_wave_kernel", 256, alg4, "both", "256 x 4 MiB log blocks: windows of a block dealt to the workgroups")
entry(4, "config4_decompress", "lz4_decompress_pcd_kernel", 256, alg4, "both",
      "one workgroup per 4 MiB block: the compressed stream is staged tile by tile in LDS (+ literals read again from memory), the output is written "
      "once; matches older than the 48 KiB of LDS history come from the written-back output")
b3 = S["configs"]["3"]["bench_line"]
alg3 = int(160 * 65536 * (1 + b3["ratio"]))
entry(3, "config3_compress", "lz4_compress_wave_kernel", 160, alg3, "both", "160 text blocks: 10 MiB, every level of the hierarchy holds it")
entry(3, "config3_decompress", "lz4_decompress_pcd_kernel", 160, alg3, "both", "one workgroup per block")
json.dump(out, open(os.path.join(P, "traffic.json"), "w"), indent=1)
print(json.dumps({k: (v["hbm_bytes_per_launch"], v["algorithmic_bytes_per_launch"], round(v["hbm_bytes_per_launch"] / v["algorithmic_bytes_per_launch"], 2))
                  for k, v in out.items() if isinstance(v, dict) and "hbm_bytes_per_launch" in v}, indent=1))
