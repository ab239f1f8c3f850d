"""GPU parity tests (-m gpu): the HIP block codec, called through the C ABI, against the oracle.
Bit-exact: decoder output/byte count/error variant == oracle; encoder output bytes == oracle
(== lz4_flex's own block bytes)."""
import hashlib

import numpy as np
import pytest

import corpus
import oracle_api as O

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("exact_encoder")]   # encoder bytes are compared with lz4_flex's: reference-exact mode


@pytest.fixture(scope="module")
def blk():
    from lz4_flex_amd import _lib, block
    lib = _lib.load()
    assert lib.lz4flex_device_count() >= 1, "GPU tests need a HIP device: " + _lib.last_error()
    return block


def _gpu_decode(blk, data, cap, dict_data=None):
    out = bytearray(cap)
    try:
        n = blk.decompress_into(data, out) if dict_data is None else blk.decompress_into_with_dict(data, out, dict_data)
    except blk.OutputTooSmall as e:
        return "OutputTooSmall", (e.expected, e.actual)
    except blk.DecompressError as e:
        return type(e).__name__, (0, 0)
    return "ok", bytes(out[:n])


# every decoder kernel behind lz4flex_decompress_batch: lanes > 0 = lz4_decompress.hip (variant 1; the default build keeps the
# 16-lane group width, 8 / 32 / 64 exist with -DLZ4FLEX_ALL_VARIANTS only),
# -408 / -432 / -464 = parser / copier split decoder with 8 / 32 / 64 blocks per workgroup (8 copier lanes x 4 bytes per block;
# 64: 4 lanes x 16 bytes), -7 = one block per workgroup (parallel-chain decoder, lz4_decompress_pcd.hip), -8 = the same kernel with
# its small test geometry (2 KiB tiles, 64-byte parts, 128 sequences per batch, 0.5 + 1 KiB window: boundaries everywhere),
# -9 = the plan / replay decoder (lz4_decompress_plan.hip + lz4_decompress_replay.hip: a copy plan per block, then four lanes per block replay it)
# -12 = the fused decoder (lz4_decompress_fused.hip: parser -> emitter -> quads in one workgroup of 64 blocks)
# -13 = one block per wavefront, one lane per sequence (lz4_decompress_seq.hip, round 6; it replaced -5 / -6, the wave decoder and its two-wavefront form)
# (-9, the plan / replay decoder, left the product library in round 5, -12 in round 6: -DLZ4FLEX_TOOLS builds only)
def _decoders_of_the_library():
    """the decoder matrix comes from the library (lz4flex_get_tuning "decoder_config_<i>": variant * 1000 + parameter; no device needed),
    so a decoder the library can be pinned to cannot go untested; encoded as above"""
    from lz4_flex_amd import _lib
    try:
        lib = _lib.load()
    except ImportError:                                   # (collection on a box without the built library: the GPU tests cannot run there anyway)
        return [16, -408, -432, -464, -7, -8, -10, -11, -13]
    out, i = [], 0
    while True:
        v = lib.lz4flex_get_tuning(None, b"decoder_config_%d" % i)
        if v < 0:
            break
        variant, par = divmod(v, 1000)
        out.append(par if variant == 1 else (-400 - par if variant == 4 else -variant))
        i += 1
    return out


DECODERS = _decoders_of_the_library()
assert set(DECODERS) >= {16, -408, -432, -464, -7, -8, -10, -11, -13}, DECODERS


def _select_decoder(lib, ctx, lanes):
    if lanes > 0:
        assert lib.lz4flex_set_tuning(ctx, b"decompress_variant", 1) == 0
        assert lib.lz4flex_set_tuning(ctx, b"decompress_lanes", lanes) == 0
    elif lanes <= -400:
        assert lib.lz4flex_set_tuning(ctx, b"decompress_variant", 4) == 0
        assert lib.lz4flex_set_tuning(ctx, b"decompress_blocks_per_wg", -lanes - 400) == 0
    elif lanes <= -7:
        assert lib.lz4flex_set_tuning(ctx, b"decompress_variant", -lanes) == 0
    else:
        raise AssertionError(lanes)


# ---------------------------------------------------------------- KATs: decompress.rs:534-622
@pytest.mark.parametrize("kat", corpus.DECODER_KATS)
def test_decoder_kats(blk, kat):
    data, cap, d, (exp, payload) = kat
    st, got = _gpu_decode(blk, data, cap, d)
    assert st == exp
    if exp == "ok":
        assert got == payload
    elif payload is not None:
        assert got == payload


@pytest.mark.parametrize("lanes", DECODERS)
def test_decoder_kats_all_group_widths(blk, lanes):
    from lz4_flex_amd import _lib
    import ctypes as C
    lib = _lib.load()
    ctx = C.c_void_p()
    assert lib.lz4flex_ctx_create(C.byref(ctx), -1) == 0
    _select_decoder(lib, ctx, lanes)
    try:
        for data, cap, d, (exp, payload) in corpus.DECODER_KATS:
            if d is not None or len(data) == 0:
                continue
            inb = np.frombuffer(data, dtype=np.uint8)
            out = np.zeros(max(cap, 1), dtype=np.uint8)
            ol, st, det = blk.decompress_batch(inb, [0], [len(data)], out, [0], [cap], ctx=ctx)
            name = "ok" if st[0] == 0 else O.ERR_NAMES[int(st[0])]
            assert name == exp
            if exp == "ok":
                assert bytes(out[:ol[0]]) == payload
            elif payload is not None:
                assert (int(det[0][0]), int(det[0][1])) == payload
    finally:
        lib.lz4flex_ctx_destroy(ctx)


@pytest.mark.parametrize("lanes", DECODERS)
def test_adversarial_batch_all_decoders(blk, lanes):
    """2 490 blocks in ONE batch (prefixes, corruptions, short sinks, runs, short periods, random data, both encoders): bytes,
    length, error variant and OutputTooSmall{expected, actual} of every block equal the oracle's, with every decoder kernel;
    nothing is written behind a sink"""
    from lz4_flex_amd import _lib
    import ctypes as C
    lib = _lib.load()
    cases = corpus.adversarial_blocks()
    want = [O.decompress(c, k) for c, k in cases]
    inb = np.frombuffer(b"".join(c for c, _ in cases) + bytes(64), dtype=np.uint8)
    in_off = np.cumsum([0] + [len(c) for c, _ in cases[:-1]])
    out_off = np.cumsum([0] + [k + 64 for _, k in cases[:-1]])
    caps = [k for _, k in cases]
    out = np.full(int(out_off[-1]) + caps[-1] + 64, 0xA5, dtype=np.uint8)
    ctx = C.c_void_p()
    assert lib.lz4flex_ctx_create(C.byref(ctx), -1) == 0
    _select_decoder(lib, ctx, lanes)
    try:
        ol, st, det = blk.decompress_batch(inb, list(in_off), [len(c) for c, _ in cases], out, list(out_off), caps, ctx=ctx)
    finally:
        lib.lz4flex_ctx_destroy(ctx)
    for i, ((c, k), w) in enumerate(zip(cases, want)):
        o = int(out_off[i])
        if w[0] == "ok":
            assert st[i] == 0 and ol[i] == len(w[1]) and out[o:o + len(w[1])].tobytes() == w[1], (i, len(c), k, int(st[i]), int(ol[i]))
        else:
            assert O.ERR_NAMES.get(int(st[i])) == w[0], (i, len(c), k, int(st[i]), w[0])
            if w[0] == "OutputTooSmall":
                assert (int(det[i][0]), int(det[i][1])) == tuple(w[1]), (i, det[i], w[1])
        assert out[o + k:o + k + 64].tobytes() == b"\xA5" * 64, "block %d wrote behind its sink" % i


@pytest.mark.parametrize("lanes", DECODERS)
def test_synthetic_copy_chain_blocks_all_decoders(blk, lanes):
    """50 hand-written blocks (corpus.synthetic_blocks: matches that aim INTO earlier matches -- copies of copies, what the
    workgroup decoder relinks --, at the bytes straddling two sequences, at their own output, at the block's first byte; every
    length class) and the same blocks cut short / with a sink a few bytes short: every decoder kernel == the oracle"""
    from lz4_flex_amd import _lib
    import ctypes as C
    lib = _lib.load()
    cases = []
    for comp, plain in corpus.synthetic_blocks():
        cases.append((comp, len(plain)))
        if len(plain) < 200000:
            cases.append((comp[:len(comp) * 2 // 3], len(plain)))
            cases.append((comp, len(plain) - 3))
    want = [O.decompress(c, k) for c, k in cases]
    assert sum(w[0] == "ok" for w in want) >= 50
    inb = np.frombuffer(b"".join(c for c, _ in cases) + bytes(64), dtype=np.uint8)
    in_off = np.cumsum([0] + [len(c) for c, _ in cases[:-1]])
    out_off = np.cumsum([0] + [k + 64 for _, k in cases[:-1]])
    caps = [k for _, k in cases]
    out = np.full(int(out_off[-1]) + caps[-1] + 64, 0xA5, dtype=np.uint8)
    ctx = C.c_void_p()
    assert lib.lz4flex_ctx_create(C.byref(ctx), -1) == 0
    _select_decoder(lib, ctx, lanes)
    try:
        ol, st, det = blk.decompress_batch(inb, list(in_off), [len(c) for c, _ in cases], out, list(out_off), caps, ctx=ctx)
    finally:
        lib.lz4flex_ctx_destroy(ctx)
    for i, ((c, k), w) in enumerate(zip(cases, want)):
        o = int(out_off[i])
        if w[0] == "ok":
            assert st[i] == 0 and ol[i] == len(w[1]) and out[o:o + len(w[1])].tobytes() == w[1], (i, len(c), k, int(st[i]), int(ol[i]))
        else:
            assert O.ERR_NAMES.get(int(st[i])) == w[0], (i, len(c), k, int(st[i]), w[0])
            if w[0] == "OutputTooSmall":
                assert (int(det[i][0]), int(det[i][1])) == tuple(w[1]), (i, det[i], w[1])
        assert out[o + k:o + k + 64].tobytes() == b"\xA5" * 64, "block %d wrote behind its sink" % i


@pytest.mark.parametrize("lanes", DECODERS)
def test_long_literal_runs_all_decoders(blk, lanes):
    """literal runs around the length from which the split decoder moves them memory to memory (lz4_decompress_split.hip,
    LONG_LIT = 1024, in 64-byte steps up to 256 bytes before the run's end) -- in the middle of a block, as its first bytes,
    as its last literals -- each followed by matches that aim into the run: at its last bytes (the rebuilt LDS window), a
    little more than 512 bytes back (the first bytes memory holds alone) and at its first byte; plus the same blocks cut
    short and with a sink a few bytes short: every decoder kernel == the oracle"""
    from lz4_flex_amd import _lib
    import ctypes as C
    lib = _lib.load()
    rng = np.random.default_rng(77)
    text = O.fixture_plain("compression_65k")
    cases = []
    for L in (960, 1023, 1024, 1025, 1087, 1088, 1089, 1279, 1280, 1343, 2048, 5000, 40000, 65000):
        noise = rng.integers(0, 256, L, dtype=np.uint8).tobytes()
        tail = noise[-100:] + text[:40] + noise[-700:-560] + text[40:60] + noise[:90] + text[100:400]
        for plain in (text[:3000] + noise + tail, noise + tail, text[:500] + noise + tail + noise[: L // 2] + bytes(rng.integers(0, 256, L, dtype=np.uint8)),
                      text[:777] + noise):
            plain = plain[:65536 * 2]
            comp = O.compress(plain)
            cases.append((comp, len(plain)))
            cases.append((comp[:len(comp) - 300], len(plain)))
            cases.append((comp, len(plain) - 3))
    want = [O.decompress(c, k) for c, k in cases]
    inb = np.frombuffer(b"".join(c for c, _ in cases) + bytes(64), dtype=np.uint8)
    in_off = np.cumsum([0] + [len(c) for c, _ in cases[:-1]])
    out_off = np.cumsum([0] + [k + 64 for _, k in cases[:-1]])
    caps = [k for _, k in cases]
    out = np.full(int(out_off[-1]) + caps[-1] + 64, 0xA5, dtype=np.uint8)
    ctx = C.c_void_p()
    assert lib.lz4flex_ctx_create(C.byref(ctx), -1) == 0
    _select_decoder(lib, ctx, lanes)
    try:
        ol, st, det = blk.decompress_batch(inb, list(in_off), [len(c) for c, _ in cases], out, list(out_off), caps, ctx=ctx)
    finally:
        lib.lz4flex_ctx_destroy(ctx)
    for i, ((c, k), w) in enumerate(zip(cases, want)):
        o = int(out_off[i])
        if w[0] == "ok":
            assert st[i] == 0 and ol[i] == len(w[1]) and out[o:o + len(w[1])].tobytes() == w[1], (i, len(c), k, int(st[i]), int(ol[i]))
        else:
            assert O.ERR_NAMES.get(int(st[i])) == w[0], (i, len(c), k, int(st[i]), w[0])
            if w[0] == "OutputTooSmall":
                assert (int(det[i][0]), int(det[i][1])) == tuple(w[1]), (i, det[i], w[1])
        assert out[o + k:o + k + 64].tobytes() == b"\xA5" * 64, "block %d wrote behind its sink" % i


# ---------------------------------------------------------------- fixtures
@pytest.mark.parametrize("stem", corpus.FIXTURES)
def test_fixture_decode_bit_exact(blk, stem):
    m = O.manifest()[stem]
    golden = O.golden_block(stem)
    plain = blk.decompress(golden, m["plain_len"])
    assert hashlib.md5(plain).hexdigest() == m["plain_md5"]
    assert ("ok", plain) == O.decompress(golden, m["plain_len"])
    # larger capacity is allowed (CHANGELOG.md:67-70) and yields the same bytes
    assert blk.decompress(golden, m["plain_len"] + 1000) == plain
    # C liblz4's encoding of the same fixture decodes to the same bytes
    assert blk.decompress(O.c_compress(plain), len(plain)) == plain


@pytest.mark.parametrize("stem", corpus.FIXTURES)
def test_fixture_encode_bit_exact(blk, stem):
    plain = O.fixture_plain(stem)
    comp = blk.compress(plain)
    assert comp == O.golden_block(stem) == O.compress(plain)
    assert O.c_decompress(comp, len(plain)) == plain


def test_config1_66k_json_single_block(blk):
    """BASELINE configs[0]: the 66 675-byte JSON fixture as one block, compress + decompress, bit-exact"""
    plain = O.fixture_plain("compression_66k_JSON")
    comp = blk.compress(plain)
    assert len(comp) == 15268
    assert blk.decompress(comp, len(plain)) == plain
    assert blk.decompress_size_prepended(blk.compress_prepend_size(plain)) == plain


# ---------------------------------------------------------------- round trips: tests/tests.rs:78-147
def _roundtrip(blk, data):
    comp = blk.compress(data)
    assert comp == O.compress(data)                                   # == reference encoder bytes
    assert blk.decompress(comp, len(data)) == data
    assert blk.decompress_size_prepended(blk.compress_prepend_size(data)) == data
    assert O.c_decompress(comp, len(data)) == data                    # flex(GPU) -> C
    if data:
        assert blk.decompress(O.c_compress(data), len(data)) == data   # C -> flex(GPU)


@pytest.mark.parametrize("i", range(len(corpus.roundtrip_inputs())))
def test_roundtrip_corpus(blk, i):
    _roundtrip(blk, corpus.roundtrip_inputs()[i])


def test_roundtrip_generated(blk):
    for seed, (alpha, run) in enumerate([(2, 1), (4, 8), (16, 3), (256, 1), (256, 64), (3, 300)]):
        for n in (1, 12, 13, 14, 64, 1000, 65534, 65535, 65536):
            _roundtrip(blk, corpus.lcg_bytes(n, seed * 1000 + n, alpha, run))


def test_output_too_small_up_front(blk):   # compress.rs:338-340
    out = bytearray(blk.get_maximum_output_size(11) - 1)
    with pytest.raises(blk.CompressOutputTooSmall):
        blk.compress_into(b"hello world", out)
    assert out == bytearray(len(out))      # nothing written


def test_conformant_last_block(blk):       # compress.rs:952-968
    a = b"a" * 15
    assert len(blk.compress(a[:12])) > 12
    for n in (13, 14, 15):
        assert len(blk.compress(a[:n])) <= n


def test_no_panic_inputs_match_oracle(blk):   # tests/tests.rs:321-351, :497-526
    for data in corpus.NO_PANIC_SIZE_PREPENDED:
        size = int.from_bytes(data[:4], "little")
        if size > 20_000_000:
            continue
        assert _gpu_decode(blk, data[4:], size) == _norm(O.decompress(data[4:], size))
        assert _gpu_decode(blk, data[4:], size, data) == _norm(O.decompress(data[4:], size, dict_data=data))


def _norm(r):
    st, payload = r
    if st not in ("ok", "OutputTooSmall"):
        return st, (0, 0)
    return st, payload


def test_dict(blk):   # compress.rs:884-949, :991-998
    inp = bytes([10, 12, 14, 16, 18] * 4)
    comp = blk.compress_with_dict(inp, inp)
    assert comp == O.compress_with_dict(inp, inp)
    assert len(comp) < len(blk.compress(inp))
    assert blk.decompress_with_dict(comp, len(inp), inp) == inp
    blk.compress_with_dict(inp, bytes([10, 12, 14]))          # test_dict_no_panic
    big = b"a" * (1 << 20)
    small = b"a" * 29
    c = blk.compress_with_dict(small, big)                      # test_dict_size: 1 MiB dict -> last 64 KiB
    assert c == O.compress_with_dict(small, big)
    assert blk.decompress_with_dict(c, len(small), big[-65536:]) == small


def test_dict_conformant_last_block(blk):   # compress.rs:970-987
    a = b"a" * 15
    assert len(blk.compress_with_dict(a[:11], a)) > 11
    assert len(blk.compress_with_dict(a[:12], a)) > 12
    for n in (13, 14, 15):
        assert len(blk.compress_with_dict(a[:n], a)) <= n


def test_dict_encode_bit_exact_generated(blk):
    """compress_into_with_dict (init_dict + ext-dict matches + both table kinds) == oracle, and round trips"""
    json = O.fixture_plain("compression_66k_JSON")
    text = O.fixture_plain("compression_65k")
    cases = [(json[:3000], json[3000:9000]), (json[:70000], json[100:30100]), (text[:100], text[50:20050]),
             (corpus.lcg_bytes(5000, 1, 4, 6), corpus.lcg_bytes(20000, 2, 4, 6)), (json[:9], json[:5000]),
             (json[:40000], json[20000:66675]), (text, text[:64000])]
    for d, data in cases:
        out = bytearray(blk.get_maximum_output_size(len(data)))
        n = blk.compress_into_with_dict(data, out, d)
        assert bytes(out[:n]) == O.compress_with_dict(data, d), (len(d), len(data))
        assert blk.decompress_with_dict(bytes(out[:n]), len(data), d[-65536:]) == data


def test_truncated_and_corrupted_blocks_match_oracle(blk):
    """every prefix / single-byte corruption of a real block: same outcome (bytes or error variant) as the oracle"""
    plain = O.fixture_plain("compression_1k")
    good = O.golden_block("compression_1k")
    cases = [good[:k] for k in range(0, len(good), 7)]
    for k in range(0, len(good), 11):
        bad = bytearray(good); bad[k] ^= 0x5A
        cases.append(bytes(bad))
    n = len(cases)
    inb = np.frombuffer(b"".join(cases), dtype=np.uint8)
    in_len = np.array([len(c) for c in cases], dtype=np.uint32)
    in_off = np.concatenate([[0], np.cumsum(in_len[:-1], dtype=np.uint64)]).astype(np.uint64)
    cap = len(plain) + 64
    out = np.full(n * cap, 0xEE, dtype=np.uint8)
    ol, st, det = blk.decompress_batch(inb, in_off, in_len, out, np.arange(n, dtype=np.uint64) * cap, np.full(n, cap, np.uint32))
    for i, c in enumerate(cases):
        est, epay = O.decompress(c, cap)
        if est == "ok":
            assert st[i] == 0 and bytes(out[i * cap:i * cap + ol[i]]) == epay
        else:
            assert O.ERR_NAMES[int(st[i])] == est
            if est == "OutputTooSmall":
                assert (int(det[i][0]), int(det[i][1])) == epay


def test_no_output_leak(blk):   # fuzz_decomp_no_output_leak.rs:16-44
    for data in corpus.NO_PANIC_SIZE_PREPENDED + [O.golden_block("compression_1k")]:
        res = []
        for fill in (0, 1):
            out = bytearray([fill]) * 4096
            try:
                n = blk.decompress_into(data, out)
                res.append(("ok", bytes(out[:n])))
            except blk.DecompressError as e:
                res.append((type(e).__name__, None))
        assert res[0] == res[1]


# ---------------------------------------------------------------- batches
def _tile(src, total, block):
    reps = (total + len(src) - 1) // len(src) + 1
    buf = (src * reps)[:total]
    return buf


@pytest.mark.parametrize("lanes,variant", [(8, 1), (16, 1)])
def test_compress_batch_bit_exact_vs_oracle(blk, lanes, variant):
    """the reference-exact encoder (compress_mode 1): both group widths produce the reference's bytes (variant 3, the encoder
    without its emitter wavefront, exists with -DLZ4FLEX_ALL_VARIANTS only)"""
    from lz4_flex_amd import _lib
    import ctypes as C
    lib = _lib.load()
    ctx = C.c_void_p()
    assert lib.lz4flex_ctx_create(C.byref(ctx), -1) == 0
    assert lib.lz4flex_set_tuning(ctx, b"compress_mode", 1) == 0
    assert lib.lz4flex_set_tuning(ctx, b"compress_lanes", lanes) == 0
    assert lib.lz4flex_set_tuning(ctx, b"compress_variant", variant) == 0
    try:
        srcs = [O.fixture_plain("compression_66k_JSON"), O.fixture_plain("compression_65k"), corpus.lcg_bytes(70000, 5, 4, 9),
                bytes(70000), corpus.lcg_bytes(70000, 6, 256, 1)]
        blocks = []
        for s in srcs:
            t = _tile(s, 6 * 65536 + 1234, 65536)
            blocks += [t[i:i + 65536] for i in range(0, len(t), 65536)]
        blocks += [b"", b"a", b"a" * 12, b"a" * 13, srcs[0][:65535], srcs[0][:65534], srcs[1][:100]]
        n = len(blocks)
        inb = np.frombuffer(b"".join(blocks), dtype=np.uint8)
        in_len = np.array([len(b) for b in blocks], dtype=np.uint32)
        in_off = np.concatenate([[0], np.cumsum(in_len[:-1], dtype=np.uint64)]).astype(np.uint64)
        stride = 72128
        out = np.zeros(n * stride, dtype=np.uint8)
        for flags in (None, np.full(n, 2, np.uint32), np.full(n, 3, np.uint32)):
            ol, st = blk.compress_batch(inb, in_off, in_len, out, np.arange(n, dtype=np.uint64) * stride,
                                        np.full(n, stride, np.uint32), flags=flags, ctx=ctx)
            assert (st == 0).all()
            for i, b in enumerate(blocks):
                got = bytes(out[i * stride:i * stride + ol[i]])
                if flags is None:
                    exp = O.compress(b)
                else:
                    exp = O.compress_frame_block(b, first_block=(flags[i] == 2))
                assert got == exp, (i, len(b), lanes, variant, None if flags is None else int(flags[i]))
    finally:
        lib.lz4flex_ctx_destroy(ctx)


def test_compress_big_blocks_bit_exact(blk):
    """blocks above 64 KiB take the u32-table kernel (config 4 uses 4 MiB blocks)"""
    src = O.fixture_plain("compression_66k_JSON") + O.fixture_plain("compression_65k")
    for n in (65537, 100000, 262144, 1 << 20):
        data = _tile(src, n, n)
        comp = blk.compress(data)
        assert comp == O.compress(data)
        assert blk.decompress(comp, n) == data


@pytest.mark.parametrize("lanes", DECODERS)
def test_decompress_batch_bit_exact_vs_oracle(blk, lanes):
    """every decoder kernel (DECODERS) on a mixed batch, byte-exact against the oracle"""
    from lz4_flex_amd import _lib
    import ctypes as C
    lib = _lib.load()
    ctx = C.c_void_p()
    assert lib.lz4flex_ctx_create(C.byref(ctx), -1) == 0
    _select_decoder(lib, ctx, lanes)
    try:
        srcs = [O.fixture_plain("compression_66k_JSON"), O.fixture_plain("compression_65k"), corpus.lcg_bytes(70000, 5, 4, 9),
                bytes(70000), corpus.lcg_bytes(70000, 6, 256, 1), corpus.lcg_bytes(70000, 7, 3, 40)]
        plains = []
        for s in srcs:
            t = _tile(s, 5 * 65536 + 777, 65536)
            plains += [t[i:i + 65536] for i in range(0, len(t), 65536)]
        plains += [b"", b"q", b"abc" * 5]
        comps = [O.compress(p) for p in plains] + [O.c_compress(p) for p in plains if p]
        plains = plains + [p for p in plains if p]
        n = len(comps)
        inb = np.frombuffer(b"".join(comps), dtype=np.uint8)
        in_len = np.array([len(c) for c in comps], dtype=np.uint32)
        in_off = np.concatenate([[0], np.cumsum(in_len[:-1], dtype=np.uint64)]).astype(np.uint64)
        out_cap = np.array([len(p) for p in plains], dtype=np.uint32)
        out_off = np.concatenate([[0], np.cumsum(out_cap[:-1], dtype=np.uint64)]).astype(np.uint64)
        out = np.zeros(int(out_cap.sum()) + 1, dtype=np.uint8)
        ol, st, det = blk.decompress_batch(inb, in_off, in_len, out, out_off, out_cap, ctx=ctx)
        assert (st == 0).all(), st
        assert (ol == out_cap).all()
        assert bytes(out[:-1]) == b"".join(plains)
    finally:
        lib.lz4flex_ctx_destroy(ctx)


def test_compress_into_with_table_variants(blk):
    """compress.rs:742-766: Small == compress_into below 65 535 bytes; an input >= 65 535 upgrades the table to Large for
    good, after which small inputs compress with the u32 table + 5-byte hash (== the frame encoder's first block)"""
    small = O.fixture_plain("compression_34k")
    big = (O.fixture_plain("compression_66k_JSON") * 2)[:100000]
    t = blk.CompressTable.small()
    out = bytearray(O.max_out(len(big)))
    for _ in range(2):                                       # reused across calls
        n = blk.compress_into_with_table(small, out, t)
        assert bytes(out[:n]) == O.compress(small) and not t.is_large
    n = blk.compress_into_with_table(big, out, t)
    assert bytes(out[:n]) == O.compress(big) and t.is_large
    n = blk.compress_into_with_table(small, out, t)
    assert bytes(out[:n]) == O.compress_frame_block(small, first_block=True)     # Large table, never downgraded
    assert O.decompress(bytes(out[:n]), len(small)) == ("ok", small)
    t2 = blk.CompressTable.large()
    n = blk.compress_into_with_table(small, out, t2)
    assert bytes(out[:n]) == O.compress_frame_block(small, first_block=True)
    with pytest.raises(blk.CompressOutputTooSmall):
        blk.compress_into_with_table(small, bytearray(10), t2)


def test_size_prepended_with_dict_entry_points(blk):
    """compress.rs:692-694, decompress.rs:521-527 through their own C entry points"""
    d = O.fixture_plain("compression_1k")
    data = (d * 3)[100:1500]
    c = blk.compress_prepend_size_with_dict(data, d)
    assert c[:4] == len(data).to_bytes(4, "little") and c[4:] == O.compress_with_dict(data, d)
    assert blk.decompress_size_prepended_with_dict(c, d) == data
    assert blk.compress_prepend_size_with_dict(data, b"ab") == len(data).to_bytes(4, "little") + O.compress(data)   # dict <= 3 bytes ignored
    with pytest.raises(blk.ExpectedAnotherByte):
        blk.decompress_size_prepended_with_dict(b"\x01\x00", d)
