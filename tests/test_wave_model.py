"""CPU: the scalar model of the throughput encoder (tests/sim/wave_encoder_model.c) emits valid LZ4 blocks that
the oracle's restatement of lz4_flex's decoder AND C liblz4 decode to the input, honours the end-of-block rules
of src/block/mod.rs:37-61, and compresses the reference's fixtures about as well as the reference encoder."""
import random

import pytest

import corpus
import oracle_api as O
import wave_model as W


def inputs():
    rnd = random.Random(1234)
    out = [b"", b"a", b"abcd" * 3, bytes(13), bytes(12), bytes(11), bytes(64), bytes(65), bytes(4096), bytes(100000)]
    out += list(corpus.ROUNDTRIP_STRINGS) + list(corpus.BUG_FUZZ)
    for stem in ("compression_1k", "compression_34k", "compression_65k", "compression_66k_JSON"):
        out.append(O.fixture_plain(stem))
    j = O.fixture_plain("compression_66k_JSON")
    for n in (8191, 8192, 8193, 16384 + 5, 65535, 65536, 65537, 131072 + 77, 200000):
        out.append((j * 4)[:n])
    out.append(bytes(rnd.getrandbits(8) for _ in range(70000)))                      # incompressible
    out.append(bytes(rnd.choice(b"ab") for _ in range(30000)))                       # tiny alphabet: long overlapping matches
    out.append(b"".join(bytes([rnd.getrandbits(8)]) * rnd.randint(1, 700) for _ in range(300)))   # runs
    out.append((bytes(range(256)) * 300)[:70001])                                    # period 256
    return out


@pytest.mark.parametrize("i", range(len(inputs())))
def test_model_round_trip(i):
    data = inputs()[i]
    comp = W.compress(data)
    assert len(comp) <= O.max_out(len(data))
    assert O.decompress(comp, len(data)) == ("ok", bytes(data))
    if data:
        assert O.c_decompress(comp, len(data)) == bytes(data)      # C liblz4 1.9.3 enforces the end-of-block rules too


def test_model_end_of_block_rules():
    """src/block/mod.rs:37-61: the last 5 bytes are literals, the last match starts >= 12 bytes before the end"""
    for n in (13, 14, 20, 64, 100, 1000, 70000):
        data = b"a" * n
        comp = W.compress(data)
        # walk the sequences
        i, o, last_match_start = 0, 0, None
        while True:
            t = comp[i]; i += 1
            lit = t >> 4
            if lit == 15:
                while True:
                    b = comp[i]; i += 1; lit += b
                    if b != 255:
                        break
            i += lit; o += lit
            if i >= len(comp):
                final_lit = lit
                break
            i += 2
            ml = (t & 15) + 4
            if (t & 15) == 15:
                while True:
                    b = comp[i]; i += 1; ml += b
                    if b != 255:
                        break
            last_match_start = o
            o += ml
        assert o == n and final_lit >= 5
        if last_match_start is not None:
            assert last_match_start <= n - 12


def test_model_ratio_close_to_reference_encoder():
    """the throughput parse is not the reference's, but it must not give away compression: within 3 % of the oracle
    (= lz4_flex's bytes) on every reference fixture, better on most"""
    for stem in ("compression_34k", "compression_65k", "compression_66k_JSON"):
        data = O.fixture_plain(stem)
        assert len(W.compress(data)) <= 1.03 * len(O.compress(data)), stem
    j = O.fixture_plain("compression_66k_JSON")
    tiles = (j * 3)[1000:1000 + 65536]
    assert len(W.compress(tiles)) <= len(O.compress(tiles))


def test_model_on_the_window_end_fixture():
    """the input tools/gpu_fuzz.py found (heads at the last positions of a window that is not the block's last): the model's block is
    a valid LZ4 block for the reference's decoder and for liblz4 (the GPU suite pins kernel == model on it)"""
    import os
    import zlib
    here = os.path.dirname(os.path.abspath(__file__))
    with open(os.path.join(here, "golden", "fuzz_window_end_heads.zlib"), "rb") as f:
        d = zlib.decompress(f.read())
    # windows that advance by 64 KiB (where the fuzzer found it) / by 32 KiB / by 48 KiB (the default for long blocks); eleven segments
    # per window (round 6) and the eight of rounds 2 - 5
    for nseg, sizes in ((11, (41659, 40896, 41124)), (8, (41631, 40898, 41108))):
        for slide, size in zip((0, 1, 2), sizes):
            c = W.compress(d, slide=slide, nseg=nseg)
            assert O.decompress(c, len(d)) == ("ok", d)
            assert O.c_decompress(c, len(d)) == d
            assert len(c) == size     # (41 666 before adjacent matches of one distance were merged, round 4)


def test_history_in_front_of_a_block():
    """hist = HIST (a Linked frame's block, LZ4FLEX_BLOCK_HISTORY): the block alone is emitted, its matches reach into the 32 KiB
    in front of it, the oracle decodes it behind those bytes as dictionary; every length class around the window arithmetic;
    and the ratio of a stream cut into 64 KiB blocks beats the reference's own Linked frame"""
    j = O.fixture_plain("compression_66k_JSON")
    H = W.HIST
    for L in (1, 5, 11, 12, 13, 100, 4096, 32767, 32768, 32769, 65535, 65536, 65537, 100000, 262144, 300001):
        s = (j * 6)[:H + L]
        c = W.compress(s, hist=H)
        assert O.decompress(c, L, dict_data=s[:H]) == ("ok", s[H:]), L
    stream = (j * 8)[:6 * 65536]
    with_h = sum(len(W.compress(stream[max(0, b - H):b + 65536], hist=H if b else 0)) for b in range(0, len(stream), 65536))
    alone = sum(len(W.compress(stream[b:b + 65536])) for b in range(0, len(stream), 65536))
    rc, ref = O.frame_compress(stream, block_size=4, block_mode=1)
    assert rc == 0
    assert with_h < alone and with_h < len(ref) - 50, (with_h, alone, len(ref))


def test_sub_windows_of_the_model():
    """sub = 2 / 3 / 4 (what the library does to the blocks of small batches): lengths around every boundary of every setting; valid
    blocks for the reference's decoder and for liblz4; blocks too short for a second sub-window are the one-window bytes"""
    j, t = O.fixture_plain("compression_66k_JSON"), O.fixture_plain("compression_65k")
    for sub, q in ((2, 32768), (3, 22016), (4, 16384)):
        for n in (q - 1, q, q + 1, 2 * q, 2 * q + 1, min(3 * q + 1, 65536), 65535, 65536, 12, 0):
            for src in (j, t):
                d = (src * 2)[7:7 + n]
                c = W.compress(d, sub=sub)
                assert O.decompress(c, len(d)) == ("ok", d), (sub, n)
                if d:
                    assert O.c_decompress(c, len(d)) == d
                if n <= q:
                    assert c == W.compress(d, sub=1)
    d = (j * 2)[:65537]
    assert W.compress(d, sub=3) == W.compress(d, sub=1)          # longer than a window: windows, not sub-windows


def test_model_run_windows():
    """round 6, run windows (lz4_compress_wave.hip index_window / run_geom; src/block/compress.rs:156-216: the reference's count_same_bytes is
    unbounded): a window of >= 8 KiB that is one byte repeated is ONE sequence.  The model's blocks are valid for the reference's decoder and
    for liblz4 on lengths around the threshold and the window ends, on runs inside other data, with every window stride, sub-windows and
    history; 4 MiB of zeros are 0.40 % (rounds 4 - 5: 0.64 %), one sequence per window"""
    j = O.fixture_plain("compression_66k_JSON")
    for n in (8191, 8192, 8193, 8203, 8204, 20000, 65535, 65536, 65537, 65536 + 8192, 131072 + 5, 300001):
        for slide in (0, 1, 2):
            for sub in (1, 4):
                d = bytes(n)
                c = W.compress(d, slide=slide if n > 65536 else 0, sub=sub)
                assert O.decompress(c, n) == ("ok", d), (n, slide, sub)
                assert O.c_decompress(c, n) == d
    z = bytes(4 << 20)
    c = W.compress(z)
    assert O.decompress(c, len(z)) == ("ok", z)
    assert len(c) <= 0.0041 * len(z)
    assert len(c) < len(O.compress(z)) * 1.03                       # the reference: ONE sequence, 0.39 %
    for d in (bytes(70000) + j + bytes(200000), j[:5000] + bytes(300000) + j[:77], bytes(65535) + b"\x01" + bytes(65536),
              bytes(1024) + b"\x01" + bytes(65536), b"\x07" * 140000):
        for slide in (0, 2):
            c = W.compress(d, slide=slide)
            assert O.decompress(c, len(d)) == ("ok", d)
            assert O.c_decompress(c, len(d)) == d
    H = W.HIST
    s = bytes(H + 200000)
    c = W.compress(s, hist=H)
    assert O.decompress(c, 200000, dict_data=s[:H]) == ("ok", s[H:])
