"""tests/sim/parallel_chain_model.c: K lanes that start at arbitrary bytes of a block's compressed stream recover the block's
token chain.  The model must return exactly the serial chain -- same sequences, same decoded length --
for every block and every way of cutting it, and must call a block irregular exactly when the serial walk does.  CPU only;
test infrastructure for a decoder that is not built yet."""
import ctypes as C
import os
import random
import subprocess

import pytest

import corpus
import oracle_api as O

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "sim", "parallel_chain_model.c")
SO = os.path.join(HERE, "sim", "libparallel_chain_model.so")


class Seq(C.Structure):
    _fields_ = [("ip", C.c_uint32), ("lit", C.c_uint32), ("ml", C.c_uint32), ("off", C.c_uint32)]


@pytest.fixture(scope="module")
def pcm():
    if not os.path.exists(SO) or os.path.getmtime(SRC) > os.path.getmtime(SO):
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-std=c11", "-Wall", SRC, "-o", SO])
    m = C.CDLL(SO)
    m.pcm_serial.restype = C.c_long
    m.pcm_serial.argtypes = [C.c_char_p, C.c_uint32, C.POINTER(Seq), C.POINTER(C.c_uint64)]
    m.pcm_parallel.restype = C.c_long
    m.pcm_parallel.argtypes = [C.c_char_p, C.c_uint32, C.POINTER(C.c_uint32), C.c_uint32, C.POINTER(Seq), C.POINTER(C.c_uint64),
                               C.POINTER(C.c_uint32)]
    m.pcm_decode.restype = C.c_long
    m.pcm_decode.argtypes = [C.c_char_p, C.c_uint32, C.POINTER(C.c_uint32), C.c_uint32, C.c_uint32, C.c_char_p, C.c_uint64,
                             C.POINTER(C.c_uint64)]
    return m


def _decode(m, c, starts, group, cap):
    out = C.create_string_buffer(max(cap, 1))
    st = (C.c_uint32 * len(starts))(*starts)
    rounds = C.c_uint64(0)
    r = m.pcm_decode(c, len(c), st, len(starts), group, out, cap, C.byref(rounds))
    return r, out.raw[:max(r, 0)], rounds.value


def _serial(m, c):
    out = (Seq * (len(c) + 2))()
    ol = C.c_uint64(0)
    k = m.pcm_serial(c, len(c), out, C.byref(ol))
    return k, [(s.ip, s.lit, s.ml, s.off) for s in out[:max(k, 0)]], ol.value


def _parallel(m, c, starts):
    out = (Seq * (len(c) + 2))()
    ol = C.c_uint64(0)
    st = (C.c_uint32 * len(starts))(*starts)
    walked = (C.c_uint32 * len(starts))()
    k = m.pcm_parallel(c, len(c), st, len(starts), out, C.byref(ol), walked)
    return k, [(s.ip, s.lit, s.ml, s.off) for s in out[:max(k, 0)]], ol.value, list(walked)


def _cuts(n, k, rnd):
    """lane starts: 0 and k - 1 further positions, evenly spaced or random"""
    if n <= 1:
        return [0]
    even = sorted(set([0] + [n * j // k for j in range(1, k) if 0 < n * j // k < n]))
    rand = sorted(set([0] + [rnd.randrange(1, n) for _ in range(k - 1)]))
    return even, rand


def _check(m, c, rnd, ks=(2, 4, 64, 255)):
    ser = _serial(m, c)
    for k in ks:
        for starts in _cuts(len(c), k, rnd) if len(c) > 1 else ([0],):
            par = _parallel(m, c, starts)
            assert par[:3] == ser, (len(c), k, starts[:8])
    return ser


def test_valid_blocks_every_cut(pcm):
    rnd = random.Random(3)
    inputs = corpus.roundtrip_inputs() + [O.fixture_plain(s) for s in corpus.FIXTURES] + [
        bytes(70000), corpus.lcg_bytes(70000, 5, 4, 9), corpus.lcg_bytes(70000, 6, 256, 1), corpus.lcg_bytes(3000, 8, 2, 300),
        corpus.lcg_bytes(300, 23, 256, 1) + bytes(70000) + corpus.lcg_bytes(70000, 24, 256, 1) + bytes(5000)]
    for p in inputs:
        if not p:
            continue
        for comp in (O.compress(p), O.c_compress(p)):
            k, seqs, out_len = _check(pcm, comp, rnd)
            assert k > 0 and out_len == len(p)
            # the sequences really are the block's: replaying them gives the input back
            out = bytearray()
            for ip, lit, ml, off in seqs:
                tok_ext = 1
                if lit >= 15:
                    tok_ext += (lit - 15) // 255 + 1
                out += comp[ip + tok_ext:ip + tok_ext + lit]
                for _ in range(ml):
                    out.append(out[-off])
            assert bytes(out) == p


def test_irregular_blocks_are_flagged_like_the_serial_walk(pcm):
    rnd = random.Random(4)
    n_irregular = 0
    for c, _cap in corpus.adversarial_blocks():
        if len(c) == 0:
            continue
        k, _, _ = _check(pcm, c, rnd, ks=(3, 64))
        n_irregular += k < 0
    assert n_irregular > 300           # the batch holds many truncated / corrupted blocks


def test_big_block_overlap_is_small(pcm):
    """a 1 MiB block cut into 64 parts: every lane walks its own part plus a few dozen bytes of the next one"""
    p = (O.fixture_plain("compression_66k_JSON") * 17)[:1 << 20]
    c = O.compress(p)
    starts = [len(c) * j // 64 for j in range(64)]
    k, seqs, out_len, walked = _parallel(pcm, c, starts)
    assert (k, seqs, out_len) == _serial(pcm, c) and out_len == len(p)
    part = len(c) / 64
    assert max(walked) < part + 600 and sum(walked) < 1.1 * len(c)


def test_whole_decoder_equals_the_oracle(pcm):
    """parallel chain + prefix-summed positions + literals at once + matches in rounds == the oracle's bytes; whatever the oracle
    rejects is handed back (negative): the reference-order path names the error"""
    rnd = random.Random(6)
    n_ok = n_rej = 0
    for c, cap in corpus.adversarial_blocks():
        if len(c) == 0:
            continue
        want = O.decompress(c, cap)
        for k, group in ((1, 64), (64, 64), (7, 16)):
            starts = _cuts(len(c), k, rnd)[1] if len(c) > 1 else [0]
            r, got, _ = _decode(pcm, c, starts, group, cap)
            if want[0] == "ok":
                assert r == len(want[1]) and got == want[1], (len(c), cap, k, r)
                n_ok += 1
            else:
                assert r < 0, (len(c), cap, k, want[0], r)
                n_rej += 1
    assert n_ok > 1000 and n_rej > 1000


def test_rounds_per_group(pcm):
    """the copies' dependency depth on the benchmark's data (tools/spec_parse_study.py): a group of 64 sequences of JSON needs 11
    rounds when the reference's encoder wrote the block, 7.4 when the throughput encoder's model did"""
    import wave_model
    p = (O.fixture_plain("compression_66k_JSON") * 5)[:1 << 18]
    for enc, lo, hi in ((O.compress, 9.0, 13.0), (wave_model.compress, 6.0, 9.0)):
        c = enc(p)
        r, got, rounds = _decode(pcm, c, [0], 64, len(p))
        assert r == len(p) and got == p
        nseq = _serial(pcm, c)[0]
        assert lo < rounds / (nseq / 64.0) < hi, rounds / (nseq / 64.0)
