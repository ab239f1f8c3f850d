"""The split decoder's PARSER (lz4_flex_amd/csrc/lz4_split_parser.h) compiled for the host (tests/sim/) and run
against the oracle: output bytes, length, error variant and OutputTooSmall{expected, actual} on the reference's
KATs (src/block/decompress.rs:534-622), the fixtures, every prefix and single-byte corruptions of real blocks,
tiny / tail-sized blocks and all four source alignments.  The byte-wise copier in the sim also checks the record
protocol (no literal read behind the block unless the record is marked careful).  CPU only."""
import ctypes as C
import os
import subprocess

import pytest

import corpus
import oracle_api as O

HERE = os.path.dirname(os.path.abspath(__file__))
SIM_DIR = os.path.join(HERE, "sim")
SIM_SO = os.path.join(SIM_DIR, "libsplit_parser_sim.so")
SIM_SRC = os.path.join(SIM_DIR, "split_parser_sim.cpp")
PARSER_H = os.path.join(HERE, "..", "lz4_flex_amd", "csrc", "lz4_split_parser.h")


def _clangxx():
    for cand in ("/opt/rocm/lib/llvm/bin/clang++", "/opt/rocm/llvm/bin/clang++"):
        if os.path.exists(cand):
            return cand
    return None


@pytest.fixture(scope="module")
def sim():
    stale = (not os.path.exists(SIM_SO) or
             os.path.getmtime(SIM_SO) < max(os.path.getmtime(SIM_SRC), os.path.getmtime(PARSER_H)))
    if stale:
        cxx = _clangxx()
        if cxx is None:
            pytest.skip("no clang++ (ext_vector_type) to build the host simulation")
        subprocess.check_call([cxx, "-O1", "-std=c++17", "-fPIC", "-shared", "-DLZ4FLEX_HOST_SIM", SIM_SRC, "-o", SIM_SO])
    lib = C.CDLL(SIM_SO)
    lib.split_parser_sim.restype = C.c_int
    lib.split_parser_sim.argtypes = [C.c_char_p, C.c_uint32, C.c_char_p, C.c_uint32, C.POINTER(C.c_uint32),
                                     C.POINTER(C.c_uint64), C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    return lib


def _sim_decode(lib, data, cap, misalign=0):
    data = bytes(data)
    out = C.create_string_buffer(max(cap, 1) + 8)
    ol = C.c_uint32(0)
    det = (C.c_uint64 * 2)()
    recs, steps = C.c_uint32(0), C.c_uint32(0)
    st = lib.split_parser_sim(data, len(data), out, cap, C.byref(ol), det, misalign, C.byref(recs), C.byref(steps))
    assert st >= 0, "simulation protocol violation %d (len %d cap %d)" % (st, len(data), cap)
    if st:
        return O.ERR_NAMES[st], (int(det[0]), int(det[1]))
    return "ok", out.raw[:ol.value]


def _same_as_oracle(lib, data, cap, misalign=0):
    want = O.decompress(data, cap)
    got = _sim_decode(lib, data, cap, misalign)
    if want[0] == "ok":
        assert got == want
    else:
        assert got[0] == want[0]
        if want[0] == "OutputTooSmall":
            assert got[1] == want[1]


def test_kats(sim):
    for data, cap, d, (exp, payload) in corpus.DECODER_KATS:
        if d is not None:
            continue
        st, got = _sim_decode(sim, data, cap)
        assert st == exp, (data, st, exp)
        if exp == "ok" or payload is not None:
            assert got == payload


@pytest.mark.parametrize("stem", corpus.FIXTURES)
def test_fixtures_all_alignments(sim, stem):
    m = O.manifest()[stem]
    blk = O.golden_block(stem)
    # bits 0-1: source misalignment; bit 2: a wide copier (offsets < 8 "rare", 15 bytes of literal slack); bit 3: small LDS layout;
    # bit 4: the parser's wave-level tests ("some lane reads its tail copy / has no token / left the fast path") fire at random
    for mis in range(32):
        _same_as_oracle(sim, blk, m["plain_len"], mis)
    _same_as_oracle(sim, blk, m["plain_len"] + 1000)
    _same_as_oracle(sim, blk, m["plain_len"] - 1)
    _same_as_oracle(sim, O.c_compress(O.fixture_plain(stem)), m["plain_len"])


def test_roundtrip_corpus_and_entropies(sim):
    inputs = corpus.roundtrip_inputs() + [corpus.lcg_bytes(70000, 5, 4, 9), corpus.lcg_bytes(70000, 6, 256, 1),
                                          corpus.lcg_bytes(70000, 7, 3, 40), bytes(70000), corpus.lcg_bytes(3000, 8, 2, 300)]
    inputs += [corpus.lcg_bytes(n, 11 + n, 5, 3) for n in list(range(0, 130)) + [255, 256, 257, 271, 272, 300, 1000]]
    for p in inputs:
        for comp in (O.compress(p), O.c_compress(p) if p else None):
            if comp is None:
                continue
            _same_as_oracle(sim, comp, len(p), len(p) % 32)
            if len(p):
                _same_as_oracle(sim, comp, len(p) - 1)


def test_every_prefix_and_corruptions(sim):
    blk = O.golden_block("compression_1k")
    n = O.manifest()["compression_1k"]["plain_len"]
    for cut in range(len(blk)):
        _same_as_oracle(sim, blk[:cut], n, cut % 4)
        _same_as_oracle(sim, blk[:cut], n, 16 + cut % 16)
    big = O.golden_block("compression_66k_JSON")
    nb = O.manifest()["compression_66k_JSON"]["plain_len"]
    for pos in list(range(0, 600)) + list(range(600, len(big), 37)) + list(range(len(big) - 80, len(big))):
        for val in (0x00, 0xFF, big[pos] ^ 0x10):
            bad = bytearray(big)
            bad[pos] = val
            _same_as_oracle(sim, bytes(bad), nb, 16 * (pos & 1))
    for junk in corpus.NO_PANIC_SIZE_PREPENDED + corpus.BUG_FUZZ:
        _same_as_oracle(sim, junk, 4096)
        _same_as_oracle(sim, junk[4:], 4096)


def test_big_blocks_and_long_chains(sim):
    """4 MiB blocks (config 4's block size), 255-chains of both kinds (the exact path), incompressible data (one literal run)"""
    cases = [corpus.lcg_bytes(4 << 20, 21, 4, 50), bytes(1 << 20), corpus.lcg_bytes(1 << 20, 22, 256, 1),
             b"ab" * 200000, corpus.lcg_bytes(300, 23, 256, 1) + bytes(70000) + corpus.lcg_bytes(70000, 24, 256, 1) + bytes(5000)]
    for p in cases:
        for comp in (O.compress(p), O.c_compress(p)):
            _same_as_oracle(sim, comp, len(p), 1)
            _same_as_oracle(sim, comp, len(p), 16)
            _same_as_oracle(sim, comp, len(p) + 17, 14)
            _same_as_oracle(sim, comp, len(p) - 1, 3)
            _same_as_oracle(sim, comp[:len(comp) - 1], len(p), 2)


def test_seeded_mutations(sim):
    """2 000 seeded multi-byte mutations of a real block: same error variant (or same bytes) as the oracle every time"""
    blk = bytearray(O.golden_block("compression_34k"))
    n = O.manifest()["compression_34k"]["plain_len"]
    x = 0x9E3779B97F4A7C15
    for it in range(2000):
        bad = bytearray(blk)
        for _ in range(1 + it % 3):
            x = (x * 6364136223846793005 + 1442695040888963407) & 0xFFFFFFFFFFFFFFFF
            pos = (x >> 20) % len(bad)
            bad[pos] = (x >> 50) & 0xFF
        cut = len(bad) if it % 5 else (x >> 7) % len(bad)
        _same_as_oracle(sim, bytes(bad[:cut]), n if it % 7 else n // 2, it % 32)


def test_hypothesis_random_inputs(sim):
    """random byte strings (two entropy regimes) and random corruptions, decoded from both encoders' blocks"""
    hyp = pytest.importorskip("hypothesis")
    st = hyp.strategies

    @hyp.settings(max_examples=150, deadline=None, database=None)
    @hyp.given(st.binary(min_size=0, max_size=3000), st.integers(0, 31), st.integers(0, 2 ** 32 - 1))
    def run(data, mis, seed):
        low = bytes(b & 3 for b in data) * 3   # long matches, 255-chains, periodic offsets
        for p in (data, low):
            for comp in (O.compress(p), O.c_compress(p) if p else b"\x00"):
                _same_as_oracle(sim, comp, len(p), mis)
                if comp:
                    bad = bytearray(comp)
                    bad[seed % len(bad)] ^= 1 << (seed >> 8) % 8
                    _same_as_oracle(sim, bytes(bad), len(p), mis)
                    _same_as_oracle(sim, comp[:seed % (len(comp) + 1)], len(p) + (seed >> 12) % 5, mis)

    run()
