"""ctypes view of tests/sim/plan_model.cpp: the host model of the plan / replay decoder (lz4_decompress_plan.hip,
lz4_decompress_replay.hip).  Test infrastructure."""
import ctypes as C
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "sim", "plan_model.cpp")
HDRS = [os.path.join(ROOT, "lz4_flex_amd", "csrc", h) for h in ("lz4_pcd_common.h", "lz4_plan_common.h")]
SO = os.path.join(ROOT, "tests", "sim", "libplan_model.so")
TURN_WORDS, END_TURNS, W = 96, 3, 2048

_m = None


def lib():
    global _m
    if _m is None:
        if not os.path.exists(SO) or max(os.path.getmtime(p) for p in [SRC] + HDRS) > os.path.getmtime(SO):
            subprocess.check_call(["g++", "-O2", "-fPIC", "-shared", "-std=c++17", "-Wall", SRC, "-o", SO])
        m = C.CDLL(SO)
        u32p = C.POINTER(C.c_uint32)
        m.plan_compile.restype = C.c_int64
        m.plan_compile.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, u32p, C.c_uint32, u32p, u32p, u32p, u32p]
        m.plan_replay.restype = C.c_int
        m.plan_replay.argtypes = [C.c_char_p, C.c_uint32, u32p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_char_p, C.c_uint32]
        m.plan_compile_batch.restype = C.c_int64
        m.plan_compile_batch.argtypes = [C.c_void_p] * 5 + [C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.POINTER(C.c_uint64)]
        _m = m
    return _m


def compile_block(comp, cap):
    """-> None (irregular block) or dict(words, n_steps, tail_word, n_tail, E)"""
    comp = bytes(comp)
    max_words = 16 * len(comp) + cap // 4 + 1024
    words = (C.c_uint32 * max_words)()
    n_steps, tail_word, n_tail, E = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
    r = lib().plan_compile(comp, len(comp), cap, words, max_words, C.byref(n_steps), C.byref(tail_word), C.byref(n_tail), C.byref(E))
    assert r >= 0, "plan does not fit %d words" % max_words
    if r == 0:
        return None
    return dict(words=words, n_words=int(r), n_steps=n_steps.value, tail_word=tail_word.value, n_tail=n_tail.value, E=E.value)


def replay(comp, plan, cap):
    """-> (code, bytes): the replay kernel's lanes on the host; code 0 = every guard held"""
    comp = bytes(comp)
    out = C.create_string_buffer(max(cap, 1))
    r = lib().plan_replay(comp, len(comp), plan["words"], plan["tail_word"], plan["n_tail"], plan["E"], out, cap)
    return r, out.raw[:plan["E"]]


def decode(comp, cap):
    """compile + replay: None for an irregular block, else the decoded bytes (asserting the guards)"""
    p = compile_block(comp, cap)
    if p is None:
        return None
    code, data = replay(comp, p, cap)
    assert code == 0, "replay guard %d" % code
    return data
