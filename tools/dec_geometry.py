#!/usr/bin/env python3
"""Kernel time and bit-exactness of every decoder geometry (and optionally encoder variant) on the BASELINE
configs[1] workload, WITHOUT importing torch (a cold `import torch` costs 1-2 min of GPU-box time): device
memory, events and copies come from libamdhip64 through ctypes, the codec from the C ABI.

usage: dec_geometry.py [--blocks 16384] [--reps 5] [--geometries 0,1] [--encoders 1]
Checks: decoded bytes == the original input (full compare on the host), encoder bytes/lengths of sampled
blocks == the oracle (tests/oracle_api.py, test infrastructure)."""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["LZ4FLEX_NO_TORCH"] = "1"

hip = None


def _hip():
    global hip
    if hip is None:
        for cand in ("libamdhip64.so", "/opt/rocm/lib/libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so.6"):
            try:
                hip = C.CDLL(cand, mode=C.RTLD_GLOBAL)
                break
            except OSError:
                continue
        if hip is None:
            raise RuntimeError("libamdhip64 not found")
    return hip


def chk(e, what=""):
    if e != 0:
        raise RuntimeError("HIP error %d %s" % (e, what))


def dmalloc(n):
    p = C.c_void_p()
    chk(_hip().hipMalloc(C.byref(p), C.c_size_t(n)), "hipMalloc %d" % n)
    return p


def h2d(dst, arr):
    arr = np.ascontiguousarray(arr)
    chk(_hip().hipMemcpy(dst, C.c_void_p(arr.ctypes.data), C.c_size_t(arr.nbytes), 1), "h2d")


def d2h(arr, src, nbytes=None):
    chk(_hip().hipMemcpy(C.c_void_p(arr.ctypes.data), src, C.c_size_t(arr.nbytes if nbytes is None else nbytes), 2), "d2h")


def dev_array(arr):
    p = dmalloc(max(arr.nbytes, 16))
    h2d(p, arr)
    return p


class Timer:
    def __init__(self):
        self.a, self.b = C.c_void_p(), C.c_void_p()
        chk(_hip().hipEventCreate(C.byref(self.a)))
        chk(_hip().hipEventCreate(C.byref(self.b)))

    def start(self):
        chk(_hip().hipEventRecord(self.a, None))

    def stop_ms(self):
        chk(_hip().hipEventRecord(self.b, None))
        chk(_hip().hipEventSynchronize(self.b))
        ms = C.c_float()
        chk(_hip().hipEventElapsedTime(C.byref(ms), self.a, self.b))
        return ms.value


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks", type=int, default=16384)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--geometries", default="0,1", help="pipelined decoder (variant 3) geometries")
    ap.add_argument("--split", default="", help="split decoder (variant 4): blocks per workgroup, e.g. 64,32")
    ap.add_argument("--encoders", default="1")
    ap.add_argument("--block-size", type=int, default=65536)
    ap.add_argument("--lib", default=None, help="private build of the library (e.g. -DLZ4FLEX_PROFILE_PHASES; see --build-prof)")
    ap.add_argument("--build-prof", action="store_true", help="build lz4_flex_amd/build/liblz4flex_prof.so (run this where hipcc is) and exit")
    ap.add_argument("--defs", default="-DLZ4FLEX_PROFILE_PHASES", help="extra -D flags of --build-prof")
    ap.add_argument("--phases", action="store_true", help="with --lib <profiling build>: print the decoder's per-phase cycle shares")
    a = ap.parse_args()
    if a.build_prof:
        import subprocess
        csrc = os.path.join(ROOT, "lz4_flex_amd", "csrc")
        srcs = ["lz4_decompress.hip", "lz4_decompress_lds.hip", "lz4_decompress_split.hip", "lz4_compress.hip", "lz4_compress_lds.hip", "xxh32_kernel.hip",
                "capi.cpp", "frame.cpp"]
        out = os.path.join(ROOT, "lz4_flex_amd", "build", "liblz4flex_prof.so")
        os.makedirs(os.path.dirname(out), exist_ok=True)
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared"] + a.defs.split() + [
                               "-x", "hip"] + [os.path.join(csrc, f) for f in srcs] + ["-o", out])
        print(out)
        return
    _hip()
    from lz4_flex_amd import _lib
    if a.lib:
        _lib.LIB_PATH = os.path.abspath(a.lib)
    import oracle_api as O
    lib = _lib.load()
    assert lib.lz4flex_device_count() >= 1, _lib.last_error()
    ctx = C.c_void_p()
    assert lib.lz4flex_ctx_create(C.byref(ctx), 0) == 0, _lib.last_error()

    n, bs = a.blocks, a.block_size
    plain = np.frombuffer(O.fixture_plain("compression_66k_JSON"), dtype=np.uint8)
    total = n * bs
    reps = total // len(plain) + 2
    src = np.tile(plain, reps)[:total].copy()           # buf[i] = json[i mod len]
    stride = (int(lib.lz4flex_get_maximum_output_size(bs)) + 63) // 64 * 64
    d_src = dev_array(src)
    d_comp = dmalloc(n * stride)
    d_out = dmalloc(total)
    ar = np.arange(n, dtype=np.uint64)
    d_in_off = dev_array(ar * bs)
    d_in_len = dev_array(np.full(n, bs, dtype=np.uint32))
    d_comp_off = dev_array(ar * stride)
    d_comp_cap = dev_array(np.full(n, stride, dtype=np.uint32))
    d_comp_len = dev_array(np.zeros(n, dtype=np.uint32))
    d_status = dev_array(np.zeros(n, dtype=np.int32))
    d_out_len = dev_array(np.zeros(n, dtype=np.uint32))
    d_detail = dev_array(np.zeros(2 * n, dtype=np.uint64))
    detail = np.zeros(2 * n, dtype=np.uint64)
    t = Timer()
    comp_len = np.zeros(n, dtype=np.uint32)
    status = np.zeros(n, dtype=np.int32)

    def compress():
        r = lib.lz4flex_compress_batch(ctx, d_src, d_in_off, d_in_len, None, n, d_comp, d_comp_off, d_comp_cap, d_comp_len,
                                       d_status, _lib.MEM_DEVICE, None)
        assert r == 0, _lib.last_error()

    def decompress():
        r = lib.lz4flex_decompress_batch(ctx, d_comp, d_comp_off, d_comp_len, n, d_out, d_in_off, d_in_len, d_out_len,
                                         d_status, d_detail, _lib.MEM_DEVICE, None)
        assert r == 0, _lib.last_error()

    first = True
    for ev in [int(x) for x in a.encoders.split(",") if x != ""]:
        assert lib.lz4flex_set_tuning(ctx, b"compress_variant", ev) == 0
        compress()
        chk(_hip().hipDeviceSynchronize())
        best = 1e9
        for _ in range(a.reps):
            t.start()
            compress()
            best = min(best, t.stop_ms())
        d2h(comp_len, d_comp_len)
        d2h(status, d_status)
        ok = bool((status == 0).all())
        # sampled blocks against the oracle's bytes
        samp = sorted(set([0, 1, n // 2, n - 1]) | set(range(7, n, max(1, n // 13))))
        buf = np.zeros(stride, dtype=np.uint8)
        for b in samp:
            want = O.compress(src[b * bs:(b + 1) * bs].tobytes())
            chk(_hip().hipMemcpy(C.c_void_p(buf.ctypes.data), C.c_void_p(d_comp.value + b * stride), C.c_size_t(stride), 2))
            ok = ok and int(comp_len[b]) == len(want) and buf[:len(want)].tobytes() == want
        print("ENC variant %d: %8.3f ms  (%7.1f GiB/s in)  ratio %.4f  oracle-exact(sampled %d)=%s" %
              (ev, best, total / 2**30 / (best / 1e3), comp_len.sum() / total, len(samp), ok), flush=True)
        first = False
    if first:
        compress()
    chk(_hip().hipDeviceSynchronize())
    d2h(comp_len, d_comp_len)
    out = np.empty(total, dtype=np.uint8)
    out_len = np.zeros(n, dtype=np.uint32)
    runs = [(3, int(x)) for x in a.geometries.split(",") if x != ""] + [(4, int(x)) for x in a.split.split(",") if x != ""]
    for variant, g in runs:
        assert lib.lz4flex_set_tuning(ctx, b"decompress_variant", variant) == 0
        key = b"decompress_geometry" if variant == 3 else b"decompress_blocks_per_wg"
        assert lib.lz4flex_set_tuning(ctx, key, g) == 0, "%s %d rejected" % (key, g)
        chk(_hip().hipMemset(d_out, 0xA5, C.c_size_t(total)))
        decompress()
        chk(_hip().hipDeviceSynchronize(), "decode %d/%d" % (variant, g))
        d2h(out, d_out)
        d2h(status, d_status)
        d2h(out_len, d_out_len)
        ok = bool((status == 0).all()) and bool((out_len == bs).all()) and bool(np.array_equal(out, src))
        best = 1e9
        for _ in range(a.reps):
            t.start()
            decompress()
            best = min(best, t.stop_ms())
        alg = (total + int(comp_len.sum())) / 1e9
        print("DEC variant %d/%d: %8.3f ms  (%7.1f GiB/s out, %6.1f GB/s algorithmic = %.2f%% of 8 TB/s)  exact=%s" %
              (variant, g, best, total / 2**30 / (best / 1e3), alg / (best / 1e3), alg / (best / 1e3) / 80.0, ok), flush=True)
        if a.phases and variant == 3:
            cyc = (C.c_ulonglong * 8)()
            cnt = (C.c_ulonglong * 8)()
            dbg = lib.lz4flex_debug_phase_dec
            dbg.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
            dbg(None, None, 1)
            decompress()
            dbg(cyc, cnt, 0)
            names = ["outer loop", "steady 4-step iteration", "drain (3 back-end steps)", "service", "p4", "p5", "p6", "p7"]
            tot = sum(cyc)
            for k in range(8):
                if cnt[k]:
                    print("   %-28s cycles/visit %8.0f  visits/wave %8.1f  share %5.1f%%" %
                          (names[k], cyc[k] / cnt[k], cnt[k] / max(cnt[0] and (n * [8, 4][g] // 64), 1), 100.0 * cyc[k] / max(tot, 1)))
        if a.phases and variant == 4:
            v = (C.c_ulonglong * 16)()
            dbg = lib.lz4flex_debug_phase_split
            dbg.argtypes = [C.c_void_p, C.c_int]
            dbg(None, 1)
            decompress()
            dbg(v, 0)
            nwg = (n + g - 1) // g
            ncw = nwg * (g // 8)
            print("   parser: %.0f cycles/wave, %.0f wave-steps (%.0f cycles/step); per block: live steps %.0f, queue-full %.0f, "
                  "window bubbles %.0f, exact path %.0f, records %.0f" %
                  (v[0] / nwg, v[1] / nwg, v[0] / max(v[1], 1), v[2] / n, v[3] / n, v[4] / n, v[5] / n, v[6] / n))
            print("   copier: %.0f cycles/wave, %.0f 4-step iterations (%.0f cycles/step), %.1f services/wave (%.0f cycles each, "
                  "%.1f%% of the wave); per block: steps with a piece %.0f, idle on empty queue %.0f, blocked/done %.0f" %
                  (v[8] / ncw, v[9] / ncw, v[8] / max(4 * v[9], 1), v[10] / ncw, v[14] / max(v[10], 1), 100.0 * v[14] / max(v[8], 1),
                   v[11] / n, v[12] / n, v[13] / n))
        if not ok:
            bad = np.nonzero(status != 0)[0]
            print("   failing status blocks:", bad[:8], status[bad[:8]] if len(bad) else "", flush=True)
            d2h(detail, d_detail)
            for b in bad[:8]:
                print("     block %d: comp_len %d detail %x %x" % (b, comp_len[b], detail[2 * b], detail[2 * b + 1]))
                dip = int(detail[2 * b]) >> 32
                if dip < comp_len[b]:
                    buf = np.zeros(stride, dtype=np.uint8)
                    chk(_hip().hipMemcpy(C.c_void_p(buf.ctypes.data), C.c_void_p(d_comp.value + int(b) * stride), C.c_size_t(stride), 2))
                    print("       bytes at ip %d:" % dip, buf[dip:dip + 24].tobytes().hex(), " oracle:",
                          O.compress(src[int(b) * bs:(int(b) + 1) * bs].tobytes())[dip:dip + 24].hex())
            if not len(bad):
                diff = np.nonzero(out != src)[0]
                print("   first diffs at", diff[:8], "block", diff[:1] // bs, "count", len(diff), flush=True)


if __name__ == "__main__":
    t0 = time.time()
    main()
    print("harness wall %.1f s" % (time.time() - t0))
