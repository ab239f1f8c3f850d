#!/usr/bin/env python3
"""End-to-end rate of the one-shot frame C calls over host buffers (PCIe, XXH32 and host framing included;
the Python wrapper's own copies are not: the C entry points are timed directly)."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from lz4_flex_amd import _lib as L, frame as fr   # noqa: E402
import oracle_api as O                            # noqa: E402

lib = L.load()
plain = O.fixture_plain("compression_66k_JSON")
total = 256 << 20
src = np.frombuffer((plain * (total // len(plain) + 2))[:total], dtype=np.uint8)
back = np.zeros(total, np.uint8)
for bs in (fr.BlockSize.Max64KB, fr.BlockSize.Max4MB):
    fi = fr.FrameInfo(block_size=bs)._c()
    cap = int(lib.lz4flex_frame_compress_bound(total, C.byref(fi)))
    out = np.zeros(cap, np.uint8)
    d = L.ErrDetail()
    consumed = C.c_size_t(0)
    for rep in range(3):
        t0 = time.perf_counter()
        r = lib.lz4flex_frame_compress(C.c_void_p(src.ctypes.data), total, C.byref(fi), C.c_void_p(out.ctypes.data), cap, C.byref(d))
        t1 = time.perf_counter()
        assert r > 0, r
        r2 = lib.lz4flex_frame_decompress(C.c_void_p(out.ctypes.data), r, C.c_void_p(back.ctypes.data), total, C.byref(consumed), C.byref(d))
        t2 = time.perf_counter()
        assert r2 == total, r2
    assert (back == src).all()
    print("%s: frame compress %.0f MiB/s, frame decompress %.0f MiB/s (256 MiB, ratio %.3f)" %
          (bs.name, 256 / (t1 - t0), 256 / (t2 - t1), r / total))
