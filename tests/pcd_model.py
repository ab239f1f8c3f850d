"""ctypes view of tests/sim/pcd_model.cpp: the host model of the parallel-chain decoder (lz4_decompress_pcd.hip).  Test infrastructure."""
import ctypes as C
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "sim", "pcd_model.cpp")
HDR = os.path.join(ROOT, "lz4_flex_amd", "csrc", "lz4_pcd_common.h")
SO = os.path.join(ROOT, "tests", "sim", "libpcd_model.so")


class Params(C.Structure):
    _fields_ = [(k, C.c_uint32) for k in ("ct", "p", "batch", "hist", "wnew", "max_iters")]


class Stats(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in ("tiles", "iters", "part_walks", "hops", "batches", "seqs", "giants", "far_bytes", "near_bytes",
                                          "depth_sum", "depth_max", "dirty_after_first")]


_m = None


def lib():
    global _m
    if _m is None:
        if not os.path.exists(SO) or max(os.path.getmtime(SRC), os.path.getmtime(HDR)) > os.path.getmtime(SO):
            subprocess.check_call(["g++", "-O2", "-fPIC", "-shared", "-std=c++17", "-Wall", SRC, "-o", SO])
        m = C.CDLL(SO)
        m.pcd_model_decode.restype = C.c_long
        m.pcd_model_decode.argtypes = [C.c_char_p, C.c_uint32, C.c_char_p, C.c_uint64, C.POINTER(Params), C.POINTER(Stats), C.c_uint64]
        m.pcd_model_defaults.argtypes = [C.POINTER(Params)]
        _m = m
    return _m


def defaults():
    p = Params()
    lib().pcd_model_defaults(C.byref(p))
    return p


def decode(comp, cap, params=None, seed=1):
    """-> (decoded bytes or None when the model calls the block irregular, Stats)"""
    comp = bytes(comp)
    p = params or defaults()
    st = Stats()
    out = C.create_string_buffer(max(cap, 1) + 64)
    r = lib().pcd_model_decode(comp, len(comp), out, cap, C.byref(p), C.byref(st), seed)
    return (None if r < 0 else out.raw[:r]), st
