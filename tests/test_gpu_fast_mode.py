"""GPU (-m gpu): the DEFAULT configuration of the library (compress_mode fast = the throughput encoder) through every layer
that the byte-exact modules (test_gpu_frame.py, test_gpu_configs.py, test_gpu_cli_props.py) only run in reference-exact mode.

The parity bar of this mode (BASELINE.json north_star, compress side) is "a valid LZ4 stream that the reference decodes to
the identical input (ratio reported)", so the checker of every test here is a FOREIGN decoder: the oracle's restatement of
lz4_flex's block / frame decoder AND C liblz4 (LZ4_decompress_safe / LZ4F), never only this library's own decoder.  The
reference's numeric pins apply to this mode as well: the ratio ceilings of tests/tests.rs:159-192."""
import io
import os
import subprocess
import sys

import numpy as np
import pytest

import corpus
import oracle_api as O
import wave_model as W

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def mods():
    from lz4_flex_amd import _lib, block, frame, sharded, workloads
    lib = _lib.load()
    assert lib.lz4flex_device_count() >= 1
    block.set_compress_mode("fast")
    assert lib.lz4flex_get_tuning(None, b"compress_mode") == 0
    return block, frame, sharded, workloads


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


def _foreign_frame_decoders_return(f, data):
    r = O.frame_decompress(f, len(data))
    assert r[0] == 0 and r[1] == data, "the oracle's FrameDecoder (lz4_flex restated) does not return the input"
    assert O.c_frame_decompress(f, len(data)) == data, "C liblz4 (LZ4F) does not return the input"


@pytest.mark.parametrize("stem", corpus.FIXTURES)
def test_block_ratio_pins_default_mode(mods, stem):
    """tests/tests.rs:159-171: the block ratio ceilings hold for the encoder a user gets by default (the 66k JSON fixture is
    66 675 bytes = two windows: the second one is anchored at the block's end so that its 1 139 bytes see their history)"""
    block = mods[0]
    data = O.fixture_plain(stem)
    comp = block.compress(data)
    assert O.decompress(comp, len(data)) == ("ok", data)
    assert O.c_decompress(comp, len(data)) == data
    assert comp == W.compress(data)
    if stem in corpus.RATIO_BLOCK:
        assert len(comp) / len(data) < corpus.RATIO_BLOCK[stem], (stem, len(comp) / len(data))


@pytest.mark.parametrize("stem", corpus.FIXTURES)
def test_frame_ratio_pins_default_mode(mods, stem):
    """tests/tests.rs:174-192 with FrameEncoder::new defaults"""
    block, frame = mods[0], mods[1]
    data = O.fixture_plain(stem)
    f = _enc(frame, data)
    _foreign_frame_decoders_return(f, data)
    assert frame.FrameDecoder.new(io.BytesIO(f)).read_to_end() == data
    if stem in corpus.RATIO_FRAME:
        assert len(f) / len(data) < corpus.RATIO_FRAME[stem], (stem, len(f) / len(data))


def test_anchored_last_window_lengths(mods):
    """blocks around the window size and with every kind of tail: == scalar model, decoded by both foreign decoders; a tail
    behind a full window compresses (it has history)"""
    block = mods[0]
    j = O.fixture_plain("compression_66k_JSON") * 6
    for n in (65536, 65537, 65536 + 11, 65536 + 12, 65536 + 13, 65536 + 64, 65536 + 255, 65536 + 256, 65536 + 257, 65536 + 511,
              65536 + 512, 65536 + 513, 66675, 2 * 65536 - 1, 2 * 65536, 2 * 65536 + 1, 131072 + 77, 3 * 65536 + 40000, 300000):
        data = j[:n]
        comp = block.compress(data)
        assert O.decompress(comp, n) == ("ok", data), n
        assert O.c_decompress(comp, n) == data, n
        assert comp == W.compress(data), n
    full, tail = block.compress(j[:65536]), block.compress(j[:65536 + 4000])
    assert len(tail) - len(full) < 0.4 * 4000


def test_multi_block_options_default_mode(mods):   # fuzz_roundtrip_frame.rs:14-80
    block, frame = mods[0], mods[1]
    data = O.fixture_plain("compression_66k_JSON") * 9 + O.fixture_plain("compression_65k") * 3
    for bs in (frame.BlockSize.Max64KB, frame.BlockSize.Max256KB, frame.BlockSize.Max1MB, frame.BlockSize.Max4MB, frame.BlockSize.Auto):
        for bc in (False, True):
            for cc in (False, True):
                for mode in (frame.BlockMode.Independent, frame.BlockMode.Linked):
                    f = _enc(frame, data, block_size=bs, block_checksums=bc, content_checksum=cc, block_mode=mode)
                    _foreign_frame_decoders_return(f, data)
                    assert frame.FrameDecoder.new(io.BytesIO(f)).read_to_end() == data
                    assert len(f) < 0.45 * len(data)
    whole = _enc(frame, data, block_size=frame.BlockSize.Max64KB)
    assert _enc(frame, data, chunks=[1, 7, 65535, 1, 65536, 100000, 13], block_size=frame.BlockSize.Max64KB) == whole
    buf = io.BytesIO()
    e = frame.FrameEncoder.with_frame_info(frame.FrameInfo(block_size=frame.BlockSize.Max64KB), buf)
    e.set_batch_bytes(3 * 65536)
    e.write_all(data); e.finish()
    assert buf.getvalue() == whole


@pytest.mark.parametrize("i", range(len(corpus.roundtrip_inputs())))
def test_roundtrip_corpus_default_mode(mods, i):   # tests/tests.rs:96-104, :126-145
    block, frame = mods[0], mods[1]
    data = corpus.roundtrip_inputs()[i]
    comp = block.compress(data)
    assert O.decompress(comp, len(data)) == ("ok", data)
    if data:
        assert O.c_decompress(comp, len(data)) == data
    for mode in (frame.BlockMode.Independent, frame.BlockMode.Linked):
        _foreign_frame_decoders_return(_enc(frame, data, block_mode=mode), data)


def test_config4_sharded_frame_default_mode(mods):
    """configs[3] at 48 MiB + a partial block through the sharded path (world 1): the gathered frame is decoded by the oracle's
    FrameDecoder and by liblz4's LZ4F, not only by this library"""
    import torch
    block, frame, sharded, Wl = mods
    n = 12 * (4 << 20) + 128 * 1000
    src = Wl.log_stream(0, n, device="cuda")
    fi = frame.FrameInfo(block_size=frame.BlockSize.Max4MB)
    fr = sharded.compress_frame_sharded(src, 0, fi)
    torch.cuda.synchronize()
    host = src.cpu().numpy().tobytes()
    got = fr.cpu().numpy().tobytes()
    _foreign_frame_decoders_return(got, host)
    assert 0.25 < len(got) / n < 0.33
    out, (lo, hi), _ = sharded.decompress_frame_sharded(fr)
    assert (lo, hi) == (0, 13) and torch.equal(out, src)
    # a device batch of 64 KiB JSON blocks (configs[1] shape): every block by both foreign block decoders
    plain = O.fixture_plain("compression_66k_JSON")
    tiles = Wl.json_tiles(plain, 96 * 65536, device="cuda")
    comp, comp_off, comp_len, in_len = sharded.compress_blocks_device(tiles, 65536, np.zeros(96, dtype=np.uint32))
    torch.cuda.synchronize()
    h_src, h_comp = tiles.cpu().numpy().tobytes(), comp.cpu().numpy()
    h_off, h_len = comp_off.cpu().tolist(), comp_len.cpu().tolist()
    for i in range(96):
        blk = bytes(h_comp[h_off[i]:h_off[i] + h_len[i]])
        want = h_src[i * 65536:(i + 1) * 65536]
        assert O.decompress(blk, 65536) == ("ok", want), i
        assert O.c_decompress(blk, 65536) == want, i
    assert sum(h_len) / len(h_src) <= 0.2321          # the reference's ratio on these tiles


def test_cli_roundtrip_default_mode(mods, tmp_path):
    data = O.fixture_plain("compression_66k_JSON") * 3
    src = tmp_path / "data.json"
    src.write_bytes(data)
    env = {k: v for k, v in os.environ.items() if k != "LZ4FLEX_COMPRESS_MODE"}
    env["PYTHONPATH"] = ROOT
    r = subprocess.run([sys.executable, "-m", "lz4_flex_amd.cli", str(src)], capture_output=True, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr
    lz = tmp_path / "data.json.lz4"
    _foreign_frame_decoders_return(lz.read_bytes(), data)
    src.unlink()
    r = subprocess.run([sys.executable, "-m", "lz4_flex_amd.cli", str(lz)], capture_output=True, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr
    assert src.read_bytes() == data
    r = subprocess.run([sys.executable, "-m", "lz4_flex_amd.cli"], input=data, capture_output=True, env=env, cwd=ROOT)
    assert r.returncode == 0
    _foreign_frame_decoders_return(r.stdout, data)


def test_property_roundtrip_default_mode(mods):   # tests/tests.rs:591-623 proptest_roundtrip
    pytest.importorskip("hypothesis")
    from hypothesis import given, settings, strategies as st
    block, frame = mods[0], mods[1]

    @st.composite
    def vec_of_vec(draw):
        length = draw(st.integers(0, 40))
        parts = [draw(st.lists(st.integers(0, max(index - 1, 0)), min_size=0, max_size=255)) for index in range(1, length)]
        return bytes(b for p in parts for b in p)

    @settings(max_examples=40, deadline=None)
    @given(vec_of_vec())
    def prop(data):
        c = block.compress(data)
        assert c == W.compress(data)
        assert O.decompress(c, len(data)) == ("ok", data)
        if data:
            assert O.c_decompress(c, len(data)) == data
        assert block.decompress(c, len(data)) == data
        for mode in (frame.BlockMode.Independent, frame.BlockMode.Linked):
            f = _enc(frame, data, block_mode=mode)
            _foreign_frame_decoders_return(f, data)
            assert frame.FrameDecoder.new(io.BytesIO(f)).read_to_end() == data
    prop()


def test_compress_deterministic_makes_bytes_a_function_of_the_block(mods):
    """src/block/compress.rs:599-601 is a pure function of its input.  The default throughput encoder is not: small batches cut their blocks
    into sub-windows ("compress_subwindows" 0), so the same block compresses to different bytes alone, among 160 and among 600 blocks.
    "compress_deterministic" 1 is what a caller that hashes / dedupes compressed blocks sets: the same bytes in every batch -- the scalar
    model's bytes without sub-windows -- and every result decodes (oracle) to the block."""
    import ctypes as C
    from lz4_flex_amd import _lib
    blk = mods[0]
    lib = _lib.load()
    plain = O.fixture_plain("compression_66k_JSON")
    tile = (plain * 3)[1234:1234 + 65536]
    other = (O.fixture_plain("compression_65k") * 3)[77:77 + 65536]
    ctx = C.c_void_p()
    assert lib.lz4flex_ctx_create(C.byref(ctx), -1) == 0
    try:
        assert lib.lz4flex_set_tuning(ctx, b"compress_mode", 0) == 0
        got = {}
        for det in (0, 1):
            assert lib.lz4flex_set_tuning(ctx, b"compress_deterministic", det) == 0
            assert lib.lz4flex_get_tuning(ctx, b"compress_deterministic") == det
            for n in (1, 160, 600):
                where = n // 2                                             # the block under test sits in the middle of the batch
                src = np.frombuffer(other * where + tile + other * (n - 1 - where), dtype=np.uint8)
                cap = O.max_out(65536)
                outb = np.zeros(n * cap, dtype=np.uint8)
                ol, st = blk.compress_batch(src, [65536 * i for i in range(n)], [65536] * n, outb, [cap * i for i in range(n)], [cap] * n, ctx=ctx)
                assert not st.any()
                b = bytes(outb[cap * where:cap * where + int(ol[where])])
                assert O.decompress(b, 65536) == ("ok", tile)
                got[(det, n)] = b
        assert got[(1, 1)] == got[(1, 160)] == got[(1, 600)] == W.compress(tile, sub=1)
        assert len({got[(0, 1)], got[(0, 160)], got[(0, 600)]}) > 1       # (the default really does depend on the batch: else this test shows nothing)
        assert lib.lz4flex_set_tuning(ctx, b"compress_deterministic", 2) < 0
    finally:
        lib.lz4flex_ctx_destroy(ctx)
