"""ctypes view of tests/sim/wave_encoder_model.c: the scalar model of the throughput ("wave") encoder
(lz4_flex_amd/csrc/lz4_compress_wave.hip).  Test infrastructure."""
import ctypes as C
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "sim", "wave_encoder_model.c")
SO = os.path.join(ROOT, "tests", "sim", "libwave_encoder_model.so")
NSEG, CAP, SKIPD = 11, 1024, 64    # the kernel's constants (lz4_compress_wave.hip: WORKERS, CAP, SKIPD)


class Params(C.Structure):
    _fields_ = [("nseg", C.c_uint32), ("cap", C.c_uint32), ("skipd", C.c_uint32), ("hist", C.c_uint32), ("slide", C.c_uint32), ("sub", C.c_uint32)]


_m = None


def lib():
    global _m
    if _m is None:
        if not os.path.exists(SO) or os.path.getmtime(SRC) > os.path.getmtime(SO):
            subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-std=c11", "-Wall", SRC, "-o", SO])
        m = C.CDLL(SO)
        m.lz4w_compress.restype = C.c_size_t
        m.lz4w_compress.argtypes = [C.c_char_p, C.c_uint32, C.c_char_p, C.POINTER(Params), C.POINTER(C.c_uint32)]
        _m = m
    return _m


HIST = 32768                      # lz4_compress_wave.hip: HIST


SLIDE_DEFAULT = 2        # "compress_sliding_window": 0 = windows advance by 64 KiB, 1 = by 32 KiB, 2 = by 48 KiB (the default since round 5)


def auto_sub(n_blocks, workgroups):
    """what the library's default ("compress_subwindows" 0) picks for a batch of n_blocks with `workgroups` persistent workgroups
    (lz4flex_get_tuning "compress_workgroups")"""
    return 4 if n_blocks * 4 <= workgroups else (3 if n_blocks * 3 <= workgroups else (2 if n_blocks * 2 <= workgroups else 1))


def compress(data, nseg=NSEG, cap=CAP, skipd=SKIPD, hist=0, slide=None, sub=None):
    """sub: 1, or 2 / 4 = sub-windows (what the library does to the blocks of small batches: auto_sub); slide: None = the library's default; hist: 0, or HIST -- `data` starts with HIST bytes of history (the stream in front of the block: a Linked frame), which are
    not emitted; the result is the block alone and needs them as its dictionary"""
    data = bytes(data)
    assert hist == 0 or (hist == HIST and len(data) > hist)
    out = C.create_string_buffer(20 + len(data) * 110 // 100 + 16)
    if slide is None:           # the library's default ("compress_sliding_window" 2): the windows of a block longer than 64 KiB advance by 48 KiB (1: by 32 KiB)
        slide = SLIDE_DEFAULT if len(data) - hist > 65536 else 0
    if sub is None:             # the library's default for a block that travels alone (a scalar call: auto_sub(1, ...) = 4); a block of a batch: auto_sub(n, workgroups)
        sub = 4
    p = Params(nseg, cap, skipd, hist, slide, sub)
    ns = C.c_uint32(0)
    n = lib().lz4w_compress(data, len(data), out, C.byref(p), C.byref(ns))
    return out.raw[:n]
