"""GPU tests (-m gpu): the lz4-like CLI (reference lz4_bin) and hypothesis-driven properties restating the
reference's proptest / fuzz targets (tests/tests.rs:591-623, fuzz/fuzz_targets/*.rs) over the C ABI."""
import io
import os
import subprocess
import sys

import pytest

import corpus
import oracle_api as O

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("exact_encoder")]   # encoder bytes are compared with lz4_flex's: reference-exact mode
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def mods():
    from lz4_flex_amd import _lib, block, frame
    assert _lib.load().lz4flex_device_count() >= 1
    return block, frame


def test_cli_roundtrip(mods, tmp_path):
    data = O.fixture_plain("compression_66k_JSON") * 3
    src = tmp_path / "data.json"
    src.write_bytes(data)
    env = dict(os.environ, PYTHONPATH=ROOT, LZ4FLEX_COMPRESS_MODE="exact")   # compared with the oracle's frame bytes
    r = subprocess.run([sys.executable, "-m", "lz4_flex_amd.cli", str(src)], capture_output=True, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr
    lz = tmp_path / "data.json.lz4"
    assert lz.read_bytes() == O.frame_compress(data)[1]           # FrameEncoder::new defaults, same bytes
    assert O.c_frame_decompress(lz.read_bytes(), len(data)) == data
    src.unlink()
    r = subprocess.run([sys.executable, "-m", "lz4_flex_amd.cli", str(lz)], capture_output=True, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr
    assert src.read_bytes() == data
    # stdin -> stdout, both directions
    r = subprocess.run([sys.executable, "-m", "lz4_flex_amd.cli"], input=data, capture_output=True, env=env, cwd=ROOT)
    assert r.returncode == 0 and O.c_frame_decompress(r.stdout, len(data)) == data
    r2 = subprocess.run([sys.executable, "-m", "lz4_flex_amd.cli", "-d"], input=r.stdout, capture_output=True, env=env, cwd=ROOT)
    assert r2.returncode == 0 and r2.stdout == data


def test_property_roundtrip_low_entropy(mods):   # tests/tests.rs:591-623 proptest_roundtrip
    hyp = pytest.importorskip("hypothesis")
    from hypothesis import given, settings, strategies as st
    block, frame = mods

    @st.composite
    def vec_of_vec(draw):
        length = draw(st.integers(0, 40))
        parts = [draw(st.lists(st.integers(0, max(index - 1, 0)), min_size=0, max_size=255)) for index in range(1, length)]
        return bytes(b for p in parts for b in p)

    @settings(max_examples=40, deadline=None)
    @given(vec_of_vec())
    def prop(data):
        c = block.compress(data)
        assert c == O.compress(data)
        assert block.decompress(c, len(data)) == data
        assert O.c_decompress(c, len(data)) == data
        for mode in (frame.BlockMode.Independent, frame.BlockMode.Linked):
            buf = io.BytesIO()
            e = frame.FrameEncoder.with_frame_info(frame.FrameInfo(block_mode=mode), buf)
            e.write_all(data); e.finish()
            assert buf.getvalue() == O.frame_compress(data, block_mode=int(mode))[1]
            assert frame.FrameDecoder.new(io.BytesIO(buf.getvalue())).read_to_end() == data
    prop()


def test_property_corrupt_blocks_match_oracle(mods):   # fuzz_decomp_corrupt_block.rs:18-40
    hyp = pytest.importorskip("hypothesis")
    from hypothesis import given, settings, strategies as st
    block, frame = mods
    good = O.golden_block("compression_1k")

    @settings(max_examples=60, deadline=None)
    @given(st.binary(min_size=0, max_size=300), st.integers(0, 2000), st.booleans())
    def prop(junk, cap, splice):
        data = (good[:len(junk)] + junk + good[len(junk) * 2:]) if splice else junk
        exp = O.decompress(data, cap)
        out = bytearray(cap)
        try:
            n = block.decompress_into(data, out)
            got = ("ok", bytes(out[:n]))
        except block.OutputTooSmall as e:
            got = ("OutputTooSmall", (e.expected, e.actual))
        except block.DecompressError as e:
            got = (type(e).__name__, exp[1])
        assert got[0] == exp[0]
        if exp[0] in ("ok", "OutputTooSmall"):
            assert got[1] == exp[1]
    prop()


def test_tuning_reads_back():
    """lz4flex_get_tuning returns what lz4flex_set_tuning stored; unknown keys and values are refused"""
    import ctypes as C
    from lz4_flex_amd import _lib
    lib = _lib.load()
    ctx = C.c_void_p()
    assert lib.lz4flex_ctx_create(C.byref(ctx), -1) == 0
    try:
        for key, vals in ((b"compress_mode", (1, 0)), (b"decompress_variant", (4, 13, 7, 1, 0)), (b"decompress_blocks_per_wg", (32, 64, 0)),
                          (b"decompress_lanes", (16,)), (b"compress_lanes", (16, 8)), (b"compress_variant", (1,))):
            for v in vals:
                assert lib.lz4flex_set_tuning(ctx, key, v) == 0, (key, v)
                assert lib.lz4flex_get_tuning(ctx, key) == v, (key, v)
        assert lib.lz4flex_get_tuning(ctx, b"no_such_key") == -_lib.E_INVALID_ARG
        assert lib.lz4flex_set_tuning(ctx, b"decompress_variant", 3) == -_lib.E_INVALID_ARG     # round 1's pipelined decoder is gone
        for gone in (5, 6, 9, 12):                                                              # the wave decoder (round 6: replaced by 13); tools-only kernels
            assert lib.lz4flex_set_tuning(ctx, b"decompress_variant", gone) == -_lib.E_INVALID_ARG
        assert lib.lz4flex_set_tuning(ctx, b"decompress_lanes", 8) == -_lib.E_INVALID_ARG       # variant builds only
        assert lib.lz4flex_set_tuning(ctx, b"compress_variant", 3) == -_lib.E_INVALID_ARG
        assert lib.lz4flex_get_tuning(ctx, b"decompress_variant") == 0
    finally:
        lib.lz4flex_ctx_destroy(ctx)
