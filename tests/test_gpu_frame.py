"""GPU parity tests (-m gpu) for the frame layer: byte-identical frames vs the oracle's restatement of
FrameEncoder, decode parity, C liblz4 (LZ4F) cross-compatibility, error variants."""
import io

import pytest

import corpus
import oracle_api as O

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("exact_encoder")]   # encoder bytes are compared with lz4_flex's: reference-exact mode


@pytest.fixture(scope="module")
def fr():
    from lz4_flex_amd import _lib, frame
    assert _lib.load().lz4flex_device_count() >= 1
    return frame


def _enc(fr, data, chunks=None, **kw):
    buf = io.BytesIO()
    e = fr.FrameEncoder.with_frame_info(fr.FrameInfo(**kw), buf)
    if chunks is None:
        e.write_all(data)
    else:
        pos = 0
        for c in chunks:
            e.write(data[pos:pos + c]); pos += c
        e.write(data[pos:])
    e.finish()
    return buf.getvalue()


def _dec(fr, data):
    return fr.FrameDecoder.new(io.BytesIO(data)).read_to_end()


def test_header_goldens(fr):   # fuzz_decomp_corrupt_frame.rs:26-27
    assert fr.FrameInfo(block_size=fr.BlockSize.Max64KB).write() == corpus.FRAME_HEADER_GOLDENS[0][1]
    assert fr.FrameInfo(block_size=fr.BlockSize.Max64KB, block_mode=fr.BlockMode.Linked).write() == corpus.FRAME_HEADER_GOLDENS[1][1]


@pytest.mark.parametrize("i", range(len(corpus.roundtrip_inputs())))
def test_roundtrip_corpus_independent(fr, i):   # tests/tests.rs:96-104, :126-145
    data = corpus.roundtrip_inputs()[i]
    f = _enc(fr, data)
    rc, exp = O.frame_compress(data)
    assert rc == 0 and f == exp                       # byte-identical to the reference encoder's frame
    assert _dec(fr, f) == data
    assert O.c_frame_decompress(f, len(data)) == data               # flex(GPU) frame -> C
    assert _dec(fr, O.c_frame_compress(data, independent=True)) == data   # C frame -> flex(GPU)


@pytest.mark.parametrize("i", range(len(corpus.roundtrip_inputs())))
def test_decode_linked_frames(fr, i):   # Linked DECODE on GPU (prefix + ext-dict window); Linked encode is a later row
    data = corpus.roundtrip_inputs()[i]
    assert _dec(fr, O.frame_compress(data, block_mode=1)[1]) == data
    assert _dec(fr, O.c_frame_compress(data, independent=False)) == data


@pytest.mark.parametrize("giveup", [1, 2, 7, 11])
def test_linked_frame_with_a_block_that_gives_up(fr, giveup, exact_encoder):
    """A block of a chained batch that gives up WITHOUT an error of its own (a bounded wait that ran out: a time-sliced GPU) leaves
    itself and every block behind it to the second pass, which decodes a chain's blocks one after the other in chain order -- side
    by side, block k + 1 would read block k's last 64 KiB while it is being written again.  Linked frames of the
    oracle's encoder (blocks that do refer to their predecessors), 64 KiB blocks, give-up forced through the library's test hook;
    and a frame of short blocks (a flush() after every small write: the run length of the launches adapts)."""
    from lz4_flex_amd import _lib
    lib = _lib.load()
    data = (O.fixture_plain("compression_66k_JSON") * 12)[:780000]
    f = O.frame_compress(data, block_mode=1, block_size=4)[1]
    assert lib.lz4flex_set_tuning(None, b"debug_chain_giveup", giveup) == 0
    try:
        assert _dec(fr, f) == data
    finally:
        assert lib.lz4flex_set_tuning(None, b"debug_chain_giveup", 0) == 0
    chunks = [3000 + 977 * (i % 13) for i in range(90)]
    small = data[:sum(chunks)]
    buf = io.BytesIO()
    e = fr.FrameEncoder.with_frame_info(fr.FrameInfo(block_mode=fr.BlockMode.Linked, block_size=fr.BlockSize.Max64KB), buf)
    at = 0
    for n in chunks:
        e.write(small[at:at + n]); e.flush()                       # 90 short blocks, each referring to its predecessors
        at += n
    e.finish()
    f2 = buf.getvalue()
    rc, back, _used = O.frame_decompress(f2, len(small))
    assert rc == 0 and back == small
    assert _dec(fr, f2) == small


@pytest.mark.parametrize("stem", corpus.FIXTURES)
def test_fixtures(fr, stem):
    data = O.fixture_plain(stem)
    f = _enc(fr, data)
    assert f == O.frame_compress(data)[1]
    assert _dec(fr, f) == data
    if stem in corpus.RATIO_FRAME:                    # tests/tests.rs:174-192
        assert len(f) / len(data) < corpus.RATIO_FRAME[stem]


def test_multi_block_options(fr):   # fuzz_roundtrip_frame.rs:14-80
    data = O.fixture_plain("compression_66k_JSON") * 9 + O.fixture_plain("compression_65k") * 3
    for bs in (fr.BlockSize.Max64KB, fr.BlockSize.Max256KB, fr.BlockSize.Max1MB, fr.BlockSize.Max4MB, fr.BlockSize.Auto):
        for bc in (False, True):
            for cc in (False, True):
                f = _enc(fr, data, block_size=bs, block_checksums=bc, content_checksum=cc)
                rc, exp = O.frame_compress(data, block_size=int(bs), block_checksums=bc, content_checksum=cc)
                assert rc == 0 and f == exp, (bs, bc, cc)
                assert _dec(fr, f) == data
                assert O.c_frame_decompress(f, len(data)) == data
    # chunked writes and small launch batches do not change the bytes
    whole = _enc(fr, data, block_size=fr.BlockSize.Max64KB)
    assert _enc(fr, data, chunks=[1, 7, 65535, 1, 65536, 100000, 13], block_size=fr.BlockSize.Max64KB) == whole
    buf = io.BytesIO()
    e = fr.FrameEncoder.with_frame_info(fr.FrameInfo(block_size=fr.BlockSize.Max64KB), buf)
    e.set_batch_bytes(3 * 65536)
    e.write_all(data); e.finish()
    assert buf.getvalue() == whole
    d = fr.FrameDecoder.new(io.BytesIO(whole)); d.set_batch_bytes(2 * 65536)
    assert d.read_to_end() == data


@pytest.mark.parametrize("i", range(len(corpus.roundtrip_inputs())))
def test_roundtrip_corpus_linked(fr, i):   # tests/tests.rs:96-104 with BlockMode::Linked
    data = corpus.roundtrip_inputs()[i]
    f = _enc(fr, data, block_mode=fr.BlockMode.Linked)
    rc, exp = O.frame_compress(data, block_mode=1)
    assert rc == 0 and f == exp
    assert _dec(fr, f) == data
    assert O.c_frame_decompress(f, len(data)) == data


def test_linked_encode_multi_block_bit_exact(fr):
    """Linked frames: blocks depend on each other (prefix + ext-dict window, persistent table): byte-identical to
    the oracle's FrameEncoder restatement, incl. the ring wrap (several multiples of block_size + 64 KiB)"""
    data = O.fixture_plain("compression_66k_JSON") * 7 + corpus.lcg_bytes(200000, 3, 8, 5) + O.fixture_plain("compression_65k") * 3
    for bs in (fr.BlockSize.Max64KB, fr.BlockSize.Max256KB):
        f = _enc(fr, data, block_mode=fr.BlockMode.Linked, block_size=bs)
        rc, exp = O.frame_compress(data, block_mode=1, block_size=int(bs))
        assert rc == 0 and f == exp, bs
        assert _dec(fr, f) == data
        assert O.c_frame_decompress(f, len(data)) == data
    # chunked writes, flush points and small launch batches
    whole = _enc(fr, data, block_mode=fr.BlockMode.Linked, block_size=fr.BlockSize.Max64KB)
    assert _enc(fr, data, chunks=[1, 7, 65535, 1, 65536, 100000, 13], block_mode=fr.BlockMode.Linked,
                block_size=fr.BlockSize.Max64KB) == whole
    buf = io.BytesIO()
    e = fr.FrameEncoder.with_frame_info(fr.FrameInfo(block_mode=fr.BlockMode.Linked, block_size=fr.BlockSize.Max64KB), buf)
    e.set_batch_bytes(3 * 65536)
    e.write_all(data); e.finish()
    assert buf.getvalue() == whole
    # a linked frame is smaller than the independent one on repetitive data (cross-block matches found)
    assert len(whole) < len(_enc(fr, data, block_size=fr.BlockSize.Max64KB))


def test_linked_frame_throughput_mode(fr):
    """compress_mode fast: a Linked frame holds independently parsed blocks (one launch per batch instead of one dependency
    chain).  Its contract is the throughput encoder's: every LZ4 frame decoder -- the oracle's restatement of lz4_flex's,
    C liblz4's, this library's -- returns the input; header and framing stay the reference's."""
    from lz4_flex_amd import block
    data = O.fixture_plain("compression_66k_JSON") * 7 + corpus.lcg_bytes(200000, 3, 8, 5) + O.fixture_plain("compression_65k") * 3
    block.set_compress_mode("fast")
    try:
        for bs in (fr.BlockSize.Max64KB, fr.BlockSize.Max256KB):
            for kw in ({}, {"block_checksums": True, "content_checksum": True}):
                f = _enc(fr, data, block_mode=fr.BlockMode.Linked, block_size=bs, **kw)
                assert f[:6] == fr.FrameInfo(block_mode=fr.BlockMode.Linked, block_size=bs, **kw).write()[:6]     # a Linked frame's header
                r = O.frame_decompress(f, len(data))
                assert r[0] == 0 and r[1] == data
                assert O.c_frame_decompress(f, len(data)) == data
                assert _dec(fr, f) == data
                assert len(f) < 0.5 * len(data)
        whole = _enc(fr, data, block_mode=fr.BlockMode.Linked, block_size=fr.BlockSize.Max64KB)
        assert _enc(fr, data, chunks=[1, 7, 65535, 1, 65536, 100000, 13], block_mode=fr.BlockMode.Linked,
                    block_size=fr.BlockSize.Max64KB) == whole
        # two frames through one encoder, empty input, a short tail
        for d in (b"", b"x", data[:70000]):
            f = _enc(fr, d, block_mode=fr.BlockMode.Linked, block_size=fr.BlockSize.Max64KB)
            assert _dec(fr, f) == d and O.c_frame_decompress(f, len(d)) == d
    finally:
        block.set_compress_mode("exact")       # the module's fixture state


def test_linked_multi_block_decode(fr):
    data = O.fixture_plain("compression_66k_JSON") * 9 + corpus.lcg_bytes(300000, 3, 8, 5)
    for bs in (4, 5):
        f = O.frame_compress(data, block_mode=1, block_size=bs)[1]
        assert _dec(fr, f) == data
    assert _dec(fr, O.c_frame_compress(data, independent=False)) == data


def test_flush_makes_block_boundary(fr):
    data = O.fixture_plain("compression_34k")
    buf = io.BytesIO()
    e = fr.FrameEncoder.new(buf)
    e.write(data[:1000]); e.flush(); e.write(data[1000:]); e.finish()
    assert _dec(fr, buf.getvalue()) == data
    assert O.c_frame_decompress(buf.getvalue(), len(data)) == data


def test_concatenated(fr):   # tests/tests.rs:633-647
    a, b = O.fixture_plain("compression_1k"), O.fixture_plain("compression_34k")
    buf = io.BytesIO()
    e = fr.FrameEncoder.new(buf)
    e.write_all(a); e.try_finish(); e.write_all(b); e.finish()
    d = fr.FrameDecoder.new(io.BytesIO(buf.getvalue()))
    assert d.read_to_end() == a
    assert d.read_to_end() == b
    assert d.read_to_end() == b""


def test_checksums(fr):   # tests/tests.rs:650-684
    for stem in ("compression_34k", "compression_66k_JSON"):
        data = O.fixture_plain(stem)
        f = bytearray(_enc(fr, data, block_checksums=True))
        assert _dec(fr, bytes(f)) == data
        f[-5] ^= 0xFF
        with pytest.raises(fr.BlockChecksumError):
            _dec(fr, bytes(f))
        f = bytearray(_enc(fr, data, content_checksum=True))
        assert _dec(fr, bytes(f)) == data
        f[-1] ^= 0xFF
        with pytest.raises(fr.ContentChecksumError):
            _dec(fr, bytes(f))


def test_content_size(fr):   # tests/tests.rs:712-737
    data = O.fixture_plain("compression_1k")
    f = bytearray(_enc(fr, data, content_size=len(data)))
    assert _dec(fr, bytes(f)) == data
    dummy = _enc(fr, b"123", content_size=3)
    f[:15] = dummy[:15]
    with pytest.raises(fr.ContentLengthError) as ei:
        _dec(fr, bytes(f))
    assert (ei.value.expected, ei.value.actual) == (3, 725)
    with pytest.raises(fr.ContentLengthError):
        _enc(fr, data, content_size=3)


def test_errors(fr):
    with pytest.raises(fr.WrongMagicNumber):
        _dec(fr, b"\x00\x01\x02\x03\x04\x05\x06")
    good = bytearray(_enc(fr, b"hello")); good[6] ^= 1
    with pytest.raises(fr.HeaderChecksumError):
        _dec(fr, bytes(good))
    with pytest.raises(fr.SkippableFrame) as ei:
        _dec(fr, (0x184D2A50).to_bytes(4, "little") + (5).to_bytes(4, "little") + b"abcde")
    assert ei.value.length == 5
    big = corpus.FRAME_HEADER_GOLDENS[0][1] + (70000).to_bytes(4, "little") + bytes(70000)
    with pytest.raises(fr.BlockTooBig):
        _dec(fr, big)
    bad_block = corpus.FRAME_HEADER_GOLDENS[0][1] + (11).to_bytes(4, "little") + bytes([0x0E, 0, 0, 0x70, 0, 0, 0, 0, 0, 0, 0])
    with pytest.raises(fr.DecompressionError) as ei:
        _dec(fr, bad_block + bytes(4))
    assert ei.value.inner == "OffsetZero"
    legacy = (0x184C2102).to_bytes(4, "little")
    blkb = O.compress(b"legacy frame payload " * 10)
    assert _dec(fr, legacy + len(blkb).to_bytes(4, "little") + blkb) == b"legacy frame payload " * 10


def test_one_shot_helpers(fr):
    data = O.fixture_plain("compression_66k_JSON") * 3
    f = fr.compress_frame(data)
    assert f == O.frame_compress(data)[1]
    out, used = fr.decompress_frame(f, len(data))
    assert out == data and used == len(f)


def test_frame_decoder_bufread(fr):
    """io::BufRead for FrameDecoder (frame/decompress.rs:410-422): fill_buf / consume walk the same bytes read() returns"""
    data = (O.fixture_plain("compression_66k_JSON") * 5)[:300000]
    for mode in (0, 1):
        rc, f = O.frame_compress(data, block_mode=mode, block_size=4)
        assert rc == 0
        dec = fr.FrameDecoder.new(io.BytesIO(f))
        got = bytearray()
        while True:
            b = dec.fill_buf()
            if not b:
                break
            k = max(1, len(b) // 3)                          # consume less than offered: the rest is offered again
            got += b[:k]
            dec.consume(k)
        assert bytes(got) == data
        with pytest.raises(ValueError):
            dec.consume(1)
