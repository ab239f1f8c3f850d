#!/usr/bin/env python3
"""where does a Linked frame's round trip go? (tools)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_api as O
from lz4_flex_amd import block, frame as F, workloads
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
data = workloads.json_tiles(O.fixture_plain("compression_66k_JSON"), n * 65536, device="cpu").numpy().tobytes()
for mode in ("fast", "exact"):
    block.set_compress_mode(mode)
    fi = F.FrameInfo(block_size=F.BlockSize.Max64KB, block_mode=F.BlockMode.Linked)
    for rep in range(3):
        t0 = time.perf_counter(); fr = F.compress_frame(data, fi); t1 = time.perf_counter()
        back = F.decompress_frame(fr, len(data))[0]; t2 = time.perf_counter()
        assert back == data
    print("%s: %d blocks, compress %.2f ms, decompress %.2f ms, ratio %.4f" % (mode, n, (t1 - t0) * 1e3, (t2 - t1) * 1e3, len(fr) / len(data)), flush=True)
    if mode == "exact":
        ref = O.frame_compress(data, block_mode=1, block_size=4)[1]
        t0 = time.perf_counter(); back = F.decompress_frame(ref, len(data))[0]; t1 = time.perf_counter()
        assert back == data and ref == fr
        print("   oracle-encoded Linked frame: decompress %.2f ms" % ((t1 - t0) * 1e3), flush=True)
fi = F.FrameInfo(block_size=F.BlockSize.Max64KB, block_mode=F.BlockMode.Independent)
block.set_compress_mode("fast")
for rep in range(3):
    t0 = time.perf_counter(); fr = F.compress_frame(data, fi); t1 = time.perf_counter()
    back = F.decompress_frame(fr, len(data))[0]; t2 = time.perf_counter()
print("independent fast: compress %.2f ms, decompress %.2f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
