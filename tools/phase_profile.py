#!/usr/bin/env python3
"""Per-phase cycle breakdown of the encoder: builds a PRIVATE copy of the library with
-DLZ4FLEX_PROFILE_PHASES (wave-level s_memtime deltas per code region), runs the bench workload once and
prints cycles per region.  Diagnostic only; never part of the shipped .so."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    csrc = os.path.join(ROOT, "lz4_flex_amd", "csrc")
    out = "/tmp/liblz4flex_prof.so"
    srcs = ["lz4_decompress.hip", "lz4_decompress_lds.hip", "lz4_decompress_split.hip", "lz4_compress.hip", "lz4_compress_lds.hip", "xxh32_kernel.hip", "capi.cpp", "frame.cpp"]
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DLZ4FLEX_PROFILE_PHASES", "-x", "hip"] + \
          [os.path.join(csrc, s) for s in srcs] + ["-o", out]
    subprocess.check_call(cmd)
    import torch
    from lz4_flex_amd import _lib
    _lib.LIB_PATH = out
    lib = _lib.load()
    from lz4_flex_amd import workloads as W, sharded
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_api as O
    plain = O.fixture_plain("compression_66k_JSON")
    n = int(os.environ.get("BLOCKS", "16384"))
    variant = int(os.environ.get("COMPRESS_VARIANT", "1"))
    ctxp = C.c_void_p()
    # the device helpers use the thread's default context: set the variant there through a first call
    lib.lz4flex_ctx_create.argtypes = [C.POINTER(C.c_void_p), C.c_int]
    src = W.json_tiles(plain, n * 65536, device="cuda")
    flags = np.zeros(n, dtype=np.uint32)
    if variant != 1:
        os.environ["LZ4FLEX_COMPRESS_VARIANT"] = str(variant)
    for _ in range(2):
        comp, comp_off, comp_len, in_len = sharded.compress_blocks_device(src, 65536, flags)
    torch.cuda.synchronize()
    cyc = (C.c_ulonglong * 8)()
    cnt = (C.c_ulonglong * 8)()
    dbg = lib.lz4flex_debug_phase2 if variant == 2 else lib.lz4flex_debug_phase   # variant 3 = the instrumented plain encoder
    dbg.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    dbg(None, None, 1)
    comp, comp_off, comp_len, in_len = sharded.compress_blocks_device(src, 65536, flags)
    dbg(cyc, cnt, 0)
    if os.environ.get("DECODE"):
        out, out_len, st = sharded.decompress_blocks_device(comp, comp_off, comp_len, None, 65536)
        dbg = lib.lz4flex_debug_phase_dec
        dbg.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        dbg(None, None, 1)
        out, out_len, st = sharded.decompress_blocks_device(comp, comp_off, comp_len, None, 65536)
        dbg(cyc, cnt, 0)
        names = ["outer loop", "steady 4-step iteration", "drain (3 back-end steps)", "service", "-", "-", "-", "-"]
        tot = sum(cyc)
        for k in range(4):
            print("%-28s cycles/visit %8.0f  visits %10d  share %5.1f%%" % (names[k], cyc[k] / max(cnt[k], 1), cnt[k], 100.0 * cyc[k] / max(tot, 1)))
        return
    names = ["top: probe wait+hash+tbl issue", "tbl wait+conflict", "cand round trip+verify", "tbl stores+winner bcast", "extension round trip", "backtrack+forward math", "next requests + emit", "rest of next-step round trip"]
    if variant == 2:
        names = ["loop", "window/stage maintenance", "generic steps", "fast steps (all)", "fs: window reads+hash+table", "fs: conflict masks+load issue",
                 "fs: wait+verify+extension math", "fs: ballots+table stores"]
    tot = sum(cyc)
    for k in range(8):
        print("%-28s cycles/visit %8.0f  visits %10d  share %5.1f%%" % (names[k], cyc[k] / max(cnt[k], 1), cnt[k], 100.0 * cyc[k] / max(tot, 1)))


if __name__ == "__main__":
    main()
