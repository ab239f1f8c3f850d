#!/usr/bin/env python3
"""Kernel time of the encoder/decoder for PRIVATE builds with extra -D flags (ablation experiments; the
results of such builds are not bit-exact and never shipped).  usage: ablate_time.py "<defs>" ["<defs>" ...]
e.g. ablate_time.py "" "-DLZ4FLEX_ABL_NOSTORE"; env COMPRESS_VARIANT / BLOCKS as in phase_profile.py."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def build(defs, out):
    csrc = os.path.join(ROOT, "lz4_flex_amd", "csrc")
    srcs = ["lz4_decompress.hip", "lz4_decompress_lds.hip", "lz4_decompress_split.hip", "lz4_compress.hip", "lz4_compress_lds.hip", "xxh32_kernel.hip", "capi.cpp", "frame.cpp"]
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared"] + defs.split() + ["-x", "hip"] + \
          [os.path.join(csrc, s) for s in srcs] + ["-o", out]
    subprocess.check_call(cmd)


def run_child(lib_path):
    import numpy as np
    import torch
    from lz4_flex_amd import _lib
    _lib.LIB_PATH = lib_path
    _lib.load()
    from lz4_flex_amd import workloads as W, sharded
    import oracle_api as O
    plain = O.fixture_plain("compression_66k_JSON")
    n = int(os.environ.get("BLOCKS", "16384"))
    src = W.json_tiles(plain, n * 65536, device="cuda")
    flags = np.zeros(n, dtype=np.uint32)
    for _ in range(2):
        comp, comp_off, comp_len, in_len = sharded.compress_blocks_device(src, 65536, flags)
    torch.cuda.synchronize()
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    tc = td = 0.0
    reps = 3
    for _ in range(reps):
        e0.record()
        comp, comp_off, comp_len, in_len = sharded.compress_blocks_device(src, 65536, flags)
        e1.record()
        out, out_len, st = sharded.decompress_blocks_device(comp, comp_off, comp_len, None, 65536)
        e2.record()
        torch.cuda.synchronize()
        tc += e0.elapsed_time(e1)
        td += e1.elapsed_time(e2)
    ok = bool((out[:n * 65536] == src).all().item())
    print("RESULT compress %.3f ms  decompress %.3f ms  (host-side wrappers included)  roundtrip_ok=%s  comp_bytes=%d" %
          (tc / reps, td / reps, ok, int(comp_len.sum().item())))


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "--child":
        run_child(sys.argv[2])
        sys.exit(0)
    for k, defs in enumerate(sys.argv[1:] or [""]):
        out = "/tmp/liblz4flex_abl%d.so" % k
        build(defs, out)
        r = subprocess.run([sys.executable, __file__, "--child", out], capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
        print("%-40s %s" % (defs or "(none)", line[0] if line else "FAILED: " + r.stderr[-400:]))
