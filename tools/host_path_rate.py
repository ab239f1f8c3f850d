#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer batch path (LZ4FLEX_MEM_HOST): pageable numpy buffers in, numpy buffers
out, staging through the context arena.  Never the benchmark's `value`."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from lz4_flex_amd import block as blk   # noqa: E402
import oracle_api as O                  # noqa: E402

plain = O.fixture_plain("compression_66k_JSON")
n, B, stride = 4096, 65536, 72128
src = (plain * (n * B // len(plain) + 2))[:n * B]
inb = np.frombuffer(src, dtype=np.uint8)
in_len = np.full(n, B, np.uint32)
in_off = np.arange(n, dtype=np.uint64) * B
out = np.zeros(n * stride, np.uint8)
out_off = np.arange(n, dtype=np.uint64) * stride
cap = np.full(n, stride, np.uint32)
back = np.zeros(n * B, np.uint8)
for rep in range(3):
    t0 = time.perf_counter()
    ol, st = blk.compress_batch(inb, in_off, in_len, out, out_off, cap)
    t1 = time.perf_counter()
    dl, dst, _ = blk.decompress_batch(out, out_off, ol, back, in_off, in_len)
    t2 = time.perf_counter()
    mib = n * B / 2**20
    print("host buffers, %d MiB: compress %.1f ms = %.0f MiB/s, decompress %.1f ms = %.0f MiB/s" %
          (mib, (t1 - t0) * 1e3, mib / (t1 - t0), (t2 - t1) * 1e3, mib / (t2 - t1)))
assert (st == 0).all() and (dst == 0).all() and bytes(back) == src
