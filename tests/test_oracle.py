"""CPU tests (-m "not gpu"): pin the oracle against every known-answer test, golden vector and fixture
the reference holds for the block/frame path, and against C liblz4 1.9.3 (the reference tests' own
cross-implementation anchor, tests/tests.rs:25-56)."""
import ctypes as C
import hashlib
import os

import pytest

import corpus
import oracle_api as O


# ---------------------------------------------------------------- decoder KATs (decompress.rs:534-622)
@pytest.mark.parametrize("kat", corpus.DECODER_KATS)
def test_decoder_kats(kat):
    data, cap, d, (exp, payload) = kat
    st, got = O.decompress(data, cap, dict_data=d)
    assert st == exp
    if exp == "ok":
        assert got == payload
    elif payload is not None:
        assert got == payload   # OutputTooSmall {expected, actual}


def test_does_token_fit():   # decompress.rs:179-186
    for tok, fit in corpus.TOKEN_FIT:
        assert bool(O.lib().lz4o_does_token_fit(tok)) == fit


def _count(first, second, cur=0, cand=0):
    c = C.c_size_t(cur)
    return int(O.lib().lz4o_count_same_bytes(bytes(first), len(first), C.byref(c), bytes(second), len(second), cand))


def test_count_same_bytes():   # compress.rs:807-881
    base = [1, 2, 3, 4] * 4
    assert _count(base + [0] * 12, base + [1] * 12) == 16
    assert _count(base + [1, 2, 3, 4] + [0] * 12, base + [1, 2, 3, 4] + [1] * 12) == 20
    assert _count(base + [1, 2, 3, 4, 3, 4] + [0] * 12, base + [1, 2, 3, 4, 3, 4] + [1] * 12) == 22
    assert _count(base + [1, 2, 3, 4, 3, 4, 5] + [0] * 12, base + [1, 2, 3, 4, 3, 4, 5] + [1] * 12) == 23
    assert _count(base + [1, 2, 3, 4, 3, 4, 5] + [0] * 12, base + [1, 2, 3, 4, 3, 4, 6] + [1] * 12) == 22
    assert _count(base + [1, 2, 3, 4, 3, 9, 5] + [0] * 12, base + [1, 2, 3, 4, 3, 4, 6] + [1] * 12) == 21
    first = bytes((i % 255) for i in range(112))
    for diff_idx in range(8, 100):
        second = bytearray(first)
        second[diff_idx] = 255
        for start in range(0, diff_idx + 1):
            assert _count(first, second, start, start) == diff_idx - start


# ---------------------------------------------------------------- encoder: fixtures, goldens, ratios
def test_get_maximum_output_size():   # compress.rs:588-590, SURVEY 8(a) a2
    assert [O.max_out(n) for n in (65536, 66675, 4194304, 0)] == [72109, 73362, 4613754, 20]


def test_tiny_encodings():   # SURVEY 8(c) derived KATs
    assert O.compress(b"") == bytes([0x00])
    assert O.compress(b"a") == bytes([0x10, 0x61])
    assert O.compress(b"a" * 12) == bytes([0xC0]) + b"a" * 12
    assert O.compress(b"a" * 13) == bytes([0x12, 0x61, 0x01, 0x00, 0x60]) + b"a" * 6
    z = O.compress(bytes(30000))
    assert len(z) == 129 and z[:5] == bytes([0x1F, 0x00, 0x01, 0x00, 0xFF])


def test_output_too_small_up_front():   # compress.rs:338-340
    with pytest.raises(ValueError, match="OutputTooSmall"):
        O.compress(b"hello world", cap=O.max_out(11) - 1)


@pytest.mark.parametrize("stem", corpus.FIXTURES)
def test_fixture_goldens(stem):
    m = O.manifest()[stem]
    plain = O.fixture_plain(stem)
    assert len(plain) == m["plain_len"]
    blk = O.compress(plain)
    assert blk == O.golden_block(stem)
    assert hashlib.md5(blk).hexdigest() == m["block_md5"] and len(blk) == m["block_len"]
    # C liblz4 decodes the oracle's block to the fixture, and the oracle decodes C liblz4's block
    assert O.c_decompress(blk, len(plain)) == plain
    assert O.decompress(O.c_compress(plain), len(plain)) == ("ok", plain)


def test_survey_sizes():   # 558 / 19 888 / 37 150 / 15 268
    got = [O.manifest()[s]["block_len"] for s in corpus.FIXTURES]
    assert got == [558, 19888, 37150, 15268]


def test_ratio_ceilings_block():   # tests/tests.rs:159-171
    for stem, ceil in corpus.RATIO_BLOCK.items():
        plain = O.fixture_plain(stem)
        assert len(O.compress(plain)) / len(plain) < ceil


def test_ratio_ceilings_frame():   # tests/tests.rs:174-192
    for stem, ceil in corpus.RATIO_FRAME.items():
        plain = O.fixture_plain(stem)
        rc, fr = O.frame_compress(plain)
        assert rc == 0 and len(fr) / len(plain) < ceil


def test_conformant_last_block():   # compress.rs:952-988
    a = b"a" * 15
    assert len(O.compress(a[:12])) > 12
    for n in (13, 14, 15):
        assert len(O.compress(a[:n])) <= n
    assert len(O.compress_with_dict(a[:11], a)) > 11
    assert len(O.compress_with_dict(a[:12], a)) > 12
    for n in (13, 14, 15):
        assert len(O.compress_with_dict(a[:n], a)) <= n


def test_dict():   # compress.rs:884-949, :991-998
    inp = bytes([10, 12, 14, 16, 18] * 4)
    comp = O.compress_with_dict(inp, inp)
    assert len(comp) < len(O.compress(inp))
    assert O.decompress(comp, len(inp), dict_data=inp) == ("ok", inp)
    O.compress_with_dict(inp, bytes([10, 12, 14]))   # no panic
    big = b"a" * (1024 * 1024)
    small = b"a" * 29
    c = O.compress_with_dict(small, big)
    assert O.decompress(c, len(small), dict_data=big[-65536:]) == ("ok", small)


# ---------------------------------------------------------------- round trips (tests/tests.rs:78-147)
def _roundtrip(data):
    c = O.compress(data)
    assert O.decompress(c, len(data)) == ("ok", data)
    assert O.c_decompress(c, len(data)) == data               # flex -> C
    if data:
        assert O.decompress(O.c_compress(data), len(data)) == ("ok", data)   # C -> flex
    for mode in (0, 1):
        rc, fr = O.frame_compress(data, block_mode=mode)
        assert rc == 0
        rc, back, used = O.frame_decompress(fr, len(data) + 16)
        assert rc == 0 and back == data and used == len(fr)
        assert O.c_frame_decompress(fr, len(data)) == data    # flex frame -> C
        rc, back, _ = O.frame_decompress(O.c_frame_compress(data, independent=(mode == 0)), len(data) + 16)
        assert rc == 0 and back == data                       # C frame -> flex


@pytest.mark.parametrize("i", range(len(corpus.roundtrip_inputs())))
def test_roundtrip_corpus(i):
    _roundtrip(corpus.roundtrip_inputs()[i])


@pytest.mark.parametrize("stem", corpus.FIXTURES)
def test_roundtrip_fixtures(stem):
    _roundtrip(O.fixture_plain(stem))


def test_roundtrip_generated():
    for seed, (alpha, run) in enumerate([(2, 1), (4, 8), (16, 3), (256, 1), (256, 64), (3, 300)]):
        for n in (1, 12, 13, 64, 1000, 65534, 65535, 65536, 65537, 200000):
            _roundtrip(corpus.lcg_bytes(n, seed * 1000 + n, alpha, run))


def test_no_panic_inputs():   # tests/tests.rs:321-351, :497-526
    for data in corpus.NO_PANIC_SIZE_PREPENDED:
        size = int.from_bytes(data[:4], "little")
        if size > 20_000_000:
            continue
        O.decompress(data[4:], size)
        O.decompress(data[4:], size, dict_data=data)


def test_no_output_leak():   # fuzz/fuzz_targets/fuzz_decomp_no_output_leak.rs:16-44
    for data in corpus.NO_PANIC_SIZE_PREPENDED + [O.golden_block("compression_1k")]:
        a = O.decompress(data, 4096, prefill=0)
        b = O.decompress(data, 4096, prefill=1)
        assert a == b


# ---------------------------------------------------------------- frame layer
def test_frame_header_goldens():   # fuzz_decomp_corrupt_frame.rs:26-27
    for kw, golden in corpus.FRAME_HEADER_GOLDENS:
        fi = O.frame_info(**kw)
        buf = C.create_string_buffer(19)
        n = O.lib().lz4o_frame_info_write(C.byref(fi), buf, 19)
        assert buf.raw[:n] == golden


def test_xxh32_matches_python_xxhash():
    xxhash = pytest.importorskip("xxhash")
    for n in (0, 1, 3, 4, 15, 16, 17, 31, 32, 100, 1000, 65536):
        data = corpus.lcg_bytes(n, n + 7)
        for seed in (0, 1, 0xDEADBEEF):
            assert O.xxh32(data, seed) == xxhash.xxh32(data, seed=seed).intdigest()


def test_frame_empty_input():   # frame/compress.rs:173-187 (header still emitted)
    rc, fr = O.frame_compress(b"")
    assert rc == 0 and fr == bytes([0x04, 0x22, 0x4D, 0x18, 0x60, 0x40, 0x82, 0, 0, 0, 0])
    assert O.c_frame_decompress(fr, 0) == b""


def test_frame_checksums():   # tests/tests.rs:650-684
    for stem in ("compression_34k", "compression_66k_JSON"):
        plain = O.fixture_plain(stem)
        rc, fr = O.frame_compress(plain, block_checksums=True)
        assert O.frame_decompress(fr, len(plain))[:2] == (0, plain)
        assert O.c_frame_decompress(fr, len(plain)) == plain      # pins block checksums against C lz4
        bad = bytearray(fr); bad[-5] ^= 0xFF
        assert O.frame_decompress(bytes(bad), len(plain))[0] == 26   # BlockChecksumError
        rc, fr = O.frame_compress(plain, content_checksum=True)
        assert O.frame_decompress(fr, len(plain))[:2] == (0, plain)
        assert O.c_frame_decompress(fr, len(plain)) == plain      # pins the content checksum against C lz4
        bad = bytearray(fr); bad[-1] ^= 0xFF
        assert O.frame_decompress(bytes(bad), len(plain))[0] == 27   # ContentChecksumError


def test_frame_content_size():   # tests/tests.rs:712-737
    plain = O.fixture_plain("compression_1k")
    rc, fr = O.frame_compress(plain, content_size=len(plain))
    assert O.frame_decompress(fr, len(plain))[:2] == (0, plain)
    rc, dummy = O.frame_compress(b"123", content_size=3)
    bad = dummy[:15] + fr[15:]
    rc, det, _ = O.frame_decompress(bad, len(plain))
    assert rc == 30 and det[:2] == (3, 725)     # ContentLengthError { expected: 3, actual: 725 }
    rc, det = O.frame_compress(plain, content_size=3)
    assert rc == 30 and det == (3, 725)


def test_frame_block_sizes():   # tests/tests.rs:687-709 (dickens fixture missing: 34k text tiled to 10 MB)
    plain = (O.fixture_plain("compression_65k") * 160)[:10 << 20]
    last = 1 << 62
    for bs in (4, 5, 6, 7):
        rc, fr = O.frame_compress(plain, block_size=bs)
        assert rc == 0 and O.frame_decompress(fr, len(plain))[:2] == (0, plain)
        assert len(fr) < last
        last = len(fr)
    assert O.c_frame_decompress(fr, len(plain)) == plain


def test_frame_concatenated():   # tests/tests.rs:633-647
    a, b = O.fixture_plain("compression_1k"), O.fixture_plain("compression_34k")
    fa, fb = O.frame_compress(a)[1], O.frame_compress(b)[1]
    cat = fa + fb
    rc, out, used = O.frame_decompress(cat, 1 << 20)
    assert (rc, out, used) == (0, a, len(fa))
    rc, out, used2 = O.frame_decompress(cat[used:], 1 << 20)
    assert (rc, out) == (0, b)


def test_frame_chunked_writes_equal_single_write():   # fuzz_roundtrip_frame.rs:14-80
    plain = O.fixture_plain("compression_66k_JSON") * 5
    for mode in (0, 1):
        for bs in (4, 5):
            rc, whole = O.frame_compress(plain, block_size=bs, block_mode=mode)
            chunks = [1, 7, 65535, 1, 65536, 100000, 13] + [4096] * 64
            rc2, parts = O.frame_compress(plain, chunks=chunks, block_size=bs, block_mode=mode)
            assert rc == 0 and rc2 == 0 and whole == parts
            assert O.c_frame_decompress(whole, len(plain)) == plain


def test_frame_independent_blocks_are_local():
    """block k of an Independent frame is a pure function of its bytes + 'first block' bit."""
    plain = O.fixture_plain("compression_66k_JSON") * 3
    rc, fr = O.frame_compress(plain, block_size=4)
    pos, k, off = 7, 0, 0
    while True:
        size = int.from_bytes(fr[pos:pos + 4], "little"); pos += 4
        if size == 0:
            break
        blk = plain[off:off + 65536]
        assert size & 0x80000000 == 0
        assert fr[pos:pos + size] == O.compress_frame_block(blk, first_block=(k == 0))
        pos += size; off += len(blk); k += 1
    assert off == len(plain)


def test_frame_errors():
    assert O.frame_decompress(b"\x00\x01\x02\x03\x04\x05\x06", 10)[0] == 21                     # WrongMagicNumber
    good = O.frame_compress(b"hello")[1]
    bad = bytearray(good); bad[6] ^= 1
    assert O.frame_decompress(bytes(bad), 10)[0] == 25                                           # HeaderChecksumError
    skippable = (0x184D2A50).to_bytes(4, "little") + (5).to_bytes(4, "little") + b"abcde"
    rc, det, _ = O.frame_decompress(skippable, 10)
    assert rc == 28 and det[0] == 5                                                              # SkippableFrame(5)
    big = bytes([0x04, 0x22, 0x4D, 0x18, 0x60, 0x40, 0x82]) + (70000).to_bytes(4, "little") + bytes(70000)
    assert O.frame_decompress(big, 100000)[0] == 24                                              # BlockTooBig
    legacy = (0x184C2102).to_bytes(4, "little")
    blk = O.compress(b"legacy frame payload " * 10)
    assert O.frame_decompress(legacy + len(blk).to_bytes(4, "little") + blk, 1000)[:2] == (0, b"legacy frame payload " * 10)
