"""ctypes view of tests/sim/fused_model.cpp: the host model of the fused decoder (lz4_decompress_fused.hip) -- the real parser and the
real emitter (host builds of lz4_split_parser.h / lz4_fused_common.h) and a lane-exact model of a block's quad.  Test infrastructure."""
import ctypes as C
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "sim", "fused_model.cpp")
HDRS = [os.path.join(ROOT, "lz4_flex_amd", "csrc", h) for h in ("lz4_split_parser.h", "lz4_fused_common.h")]
SO = os.path.join(ROOT, "tests", "sim", "libfused_model.so")
MAX_FIELD = (1 << 19) - 1

_m = None


def clangxx():
    for cand in ("/opt/rocm/lib/llvm/bin/clang++", "/opt/rocm/llvm/bin/clang++"):
        if os.path.exists(cand):
            return cand
    return None


def lib():
    global _m
    if _m is None:
        if not os.path.exists(SO) or max(os.path.getmtime(p) for p in [SRC] + HDRS) > os.path.getmtime(SO):
            cxx = clangxx()                      # (the parser uses ext_vector_type: clang)
            if cxx is None:
                raise RuntimeError("no clang++ to build the host model")
            subprocess.check_call([cxx, "-O1", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function", "-DLZ4FLEX_HOST_SIM", SRC, "-o", SO])
        m = C.CDLL(SO)
        m.fused_model_run.restype = C.c_int
        m.fused_model_run.argtypes = [C.c_char_p, C.c_uint32, C.c_char_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.c_uint32,
                                      C.c_uint32, C.POINTER(C.c_uint64)]
        _m = m
    return _m


def decode(comp, cap, misalign=0, seed=0):
    """-> (code, bytes, detail, stats): code 0 / 1..5 = the reference's outcome, negative = a guard of the model fired;
    stats = parser steps, emitter iterations, steps executed, quad turns, pieces, resting turns"""
    comp = bytes(comp)
    out = C.create_string_buffer(max(cap, 1) + 64)
    n = C.c_uint32(0)
    detail = (C.c_uint64 * 2)()
    stats = (C.c_uint64 * 8)()
    r = lib().fused_model_run(comp, len(comp), out, cap, C.byref(n), detail, misalign, seed, stats)
    return r, out.raw[:n.value], (detail[0], detail[1]), list(stats)
