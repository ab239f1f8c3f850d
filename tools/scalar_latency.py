#!/usr/bin/env python3
"""Latency of the scalar drop-in calls (lz4flex_compress_into / lz4flex_decompress_into, host buffers: PCIe both ways, one
launch, one synchronisation per call) for one block of a given size.  Run on the GPU box."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import oracle_api as O
    from lz4_flex_amd import block
    plain = O.fixture_plain("compression_66k_JSON")
    for size in (1024, 65536, 1 << 20, 16 << 20):
        data = (plain * (size // len(plain) + 2))[:size]
        comp = block.compress(data)
        assert block.decompress(comp, size) == data
        reps = 200 if size <= 65536 else 20
        t0 = time.perf_counter()
        for _ in range(reps):
            block.compress(data)
        tc = (time.perf_counter() - t0) / reps
        t0 = time.perf_counter()
        for _ in range(reps):
            block.decompress(comp, size)
        td = (time.perf_counter() - t0) / reps
        print("block of %8d bytes: compress_into %8.3f ms (%7.1f MiB/s), decompress_into %8.3f ms (%7.1f MiB/s), ratio %.3f" %
              (size, tc * 1e3, size / 1048576 / tc, td * 1e3, size / 1048576 / td, len(comp) / size), flush=True)


if __name__ == "__main__":
    main()
