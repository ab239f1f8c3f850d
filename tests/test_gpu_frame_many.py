"""GPU parity tests (-m gpu) for lz4flex_frame_compress_many / lz4flex_frame_decompress_many: N streams, a frame each, all blocks of
all streams in one batch (Linked frames: N chains side by side).  Every frame written is decoded by the oracle's FrameDecoder
(lz4_flex's, restated) and, in reference-exact mode, compared byte for byte with the oracle's FrameEncoder; every frame decoded is
compared with the stream it was made from; whatever the batch path leaves to the streaming decoder (flush boundaries, concatenated
frames, stored blocks, errors) must come back exactly as lz4flex_frame_decompress returns it for that stream alone."""
import io
import random

import pytest

import corpus
import oracle_api as O

pytestmark = [pytest.mark.gpu]


@pytest.fixture(scope="module")
def fr():
    from lz4_flex_amd import _lib, frame
    assert _lib.load().lz4flex_device_count() >= 1
    return frame


def _streams(seed, n, lo, hi):
    rnd = random.Random(seed)
    json, text = O.fixture_plain("compression_66k_JSON"), O.fixture_plain("compression_65k")
    out = []
    for i in range(n):
        size = rnd.randrange(lo, hi)
        src = (json, text)[i % 2]
        at = rnd.randrange(0, len(src))
        s = (src[at:] + src * (size // len(src) + 1))[:size]
        if i % 5 == 4:                                           # incompressible stretches: blocks stored raw
            s = s[:size // 3] + bytes(rnd.getrandbits(8) for _ in range(min(70000, size // 3))) + s[size // 3:]
            s = s[:size]
        out.append(s)
    return out


MODES = [(0, 4), (1, 4), (1, 5), (0, 0), (1, 7)]


@pytest.mark.parametrize("mode,bs", MODES)
@pytest.mark.parametrize("compress_mode", ["fast", "exact"])
def test_round_trip_many_streams(fr, mode, bs, compress_mode):
    from lz4_flex_amd import block
    streams = _streams(100 * mode + bs, 37, 1, 700000) + [b"", b"a", bytes(65536), bytes(65537), bytes(131072)]
    info = fr.FrameInfo(block_mode=fr.BlockMode(mode), block_size=fr.BlockSize(bs))
    block.set_compress_mode(compress_mode)
    try:
        frames = fr.compress_frames(streams, info)
    finally:
        block.set_compress_mode("fast")
    assert len(frames) == len(streams)
    for s, f in zip(streams, frames):
        rc, back, used = O.frame_decompress(f, len(s))
        assert rc == 0 and back == s and used == len(f)           # the reference's FrameDecoder returns the stream
        if compress_mode == "exact":
            rc, exp = O.frame_compress(s, block_mode=mode, block_size=bs)
            assert rc == 0 and f == exp                           # ... and the frame is the reference encoder's, byte for byte
    back = fr.decompress_frames(frames, [len(s) for s in streams])
    assert back == streams
    # generous and exact capacities give the same result
    assert fr.decompress_frames(frames, [len(s) + 100000 for s in streams]) == streams


@pytest.mark.parametrize("bc,cc,size", [(True, False, False), (False, True, False), (True, True, True)])
@pytest.mark.parametrize("mode", [0, 1])
def test_checksums_and_content_size(fr, mode, bc, cc, size):
    streams = _streams(7 + mode, 19, 1, 400000) + [b""]
    info = fr.FrameInfo(block_mode=fr.BlockMode(mode), block_size=fr.BlockSize.Max64KB, block_checksums=bc, content_checksum=cc,
                        content_size=0 if size else None)
    frames = fr.compress_frames(streams, info)
    for s, f in zip(streams, frames):
        rc, back, used = O.frame_decompress(f, len(s))
        assert rc == 0 and back == s and used == len(f)
        if size:
            assert fr.FrameInfo.read(f[:19]).content_size == len(s)
    assert fr.decompress_frames(frames, [len(s) for s in streams]) == streams
    # one flipped bit in a payload / in the content checksum: that stream fails as it does alone, the others are untouched
    bad = list(frames)
    victim = 3
    b = bytearray(bad[victim]); b[len(b) // 2] ^= 0x10; bad[victim] = bytes(b)
    res = fr.decompress_frames(bad, [len(s) for s in streams], return_errors=True)
    for i, (s, r) in enumerate(zip(streams, res)):
        if i != victim:
            assert r == s
    try:
        alone = fr.decompress_frame(bad[victim], len(streams[victim]))[0]
    except Exception as e:
        alone = e
    if isinstance(alone, Exception):
        assert type(res[victim]) is type(alone)
    else:
        assert res[victim] == alone


def test_frames_of_the_reference_encoder(fr):
    """frames written by the oracle's FrameEncoder (lz4_flex's bytes: Linked blocks that do refer to their predecessors), by the C
    library, with flush boundaries, with a second frame behind -- every shape goes through decompress_frames"""
    streams = _streams(3, 24, 1, 500000)
    frames = []
    for i, s in enumerate(streams):
        kind = i % 6
        if kind == 0:
            frames.append(O.frame_compress(s, block_mode=1, block_size=4)[1])
        elif kind == 1:
            frames.append(O.frame_compress(s, block_mode=0, block_size=4)[1])
        elif kind == 2:
            frames.append(O.c_frame_compress(s, independent=False))
        elif kind == 3:                                           # flush() boundaries: short blocks inside the frame
            buf = io.BytesIO()
            e = fr.FrameEncoder.with_frame_info(fr.FrameInfo(block_mode=fr.BlockMode.Linked, block_size=fr.BlockSize.Max64KB), buf)
            e.write(s[:1000]); e.flush(); e.write(s[1000:71000]); e.flush(); e.write(s[71000:]); e.finish()
            rc, b, _ = O.frame_decompress(buf.getvalue(), len(s))
            assert rc == 0 and b == s
            frames.append(buf.getvalue())
        elif kind == 4:                                           # two frames back to back: the first one is the stream's
            half = len(s) // 2
            frames.append(O.frame_compress(s[:half], block_mode=1, block_size=4)[1] + O.frame_compress(s[half:], block_mode=0, block_size=5)[1])
        else:
            frames.append(O.frame_compress(s, block_mode=1, block_size=5, content_checksum=True, block_checksums=True)[1])
    want = [s[:len(s) // 2] if i % 6 == 4 else s for i, s in enumerate(streams)]   # (read_to_end ends with the first frame, tests/tests.rs:633-647)
    assert fr.decompress_frames(frames, [len(s) for s in streams]) == want


def test_errors_are_the_single_stream_errors(fr):
    good = _streams(5, 6, 100000, 300000)
    frames = [O.frame_compress(s, block_mode=1, block_size=4)[1] for s in good]
    cases = list(frames)
    cases[0] = b"\x00\x01\x02\x03\x04\x05\x06"                                               # WrongMagicNumber
    h = bytearray(frames[1]); h[6] ^= 1; cases[1] = bytes(h)                                  # HeaderChecksumError
    cases[2] = frames[2][:len(frames[2]) // 2]                                                # truncated
    cases[3] = corpus.FRAME_HEADER_GOLDENS[0][1] + (70000).to_bytes(4, "little") + bytes(70000)   # BlockTooBig
    cases[4] = corpus.FRAME_HEADER_GOLDENS[0][1] + (11).to_bytes(4, "little") + bytes([0x0E, 0, 0, 0x70, 0, 0, 0, 0, 0, 0, 0]) + bytes(4)   # OffsetZero
    res = fr.decompress_frames(cases, [len(s) for s in good], return_errors=True)
    for i, c in enumerate(cases):
        try:
            alone = fr.decompress_frame(c, len(good[i]))[0]
        except Exception as e:
            alone = e
        if isinstance(alone, Exception):
            assert type(res[i]) is type(alone), (i, res[i], alone)
            assert getattr(res[i], "inner", None) == getattr(alone, "inner", None)
        else:
            assert res[i] == alone, i
    assert res[5] == good[5]
    # an output buffer too small: that stream fails, the others do not
    caps = [len(s) for s in good]
    caps[2] -= 1
    res = fr.decompress_frames(frames, caps, return_errors=True)
    assert isinstance(res[2], Exception)
    assert [r for i, r in enumerate(res) if i != 2] == [s for i, s in enumerate(good) if i != 2]


@pytest.mark.parametrize("giveup", [1, 5, 40])
def test_a_chain_block_that_gives_up(fr, giveup):
    """a block of the chained batch that gives up without an error (a bounded wait that ran out) leaves itself and its chain's later
    blocks to the ordered second pass: same bytes"""
    from lz4_flex_amd import _lib
    lib = _lib.load()
    streams = _streams(11, 9, 200000, 500000)
    frames = [O.frame_compress(s, block_mode=1, block_size=4)[1] for s in streams]
    assert lib.lz4flex_set_tuning(None, b"debug_chain_giveup", giveup) == 0
    try:
        assert fr.decompress_frames(frames, [len(s) for s in streams]) == streams
    finally:
        assert lib.lz4flex_set_tuning(None, b"debug_chain_giveup", 0) == 0


@pytest.mark.parametrize("variant", [0, 7, 8, 10, 11])
def test_device_resident_many_linked_streams(fr, variant):
    """256 streams of 512 KiB in device memory, Linked 64 KiB blocks: compress_many -> decompress_many on the device, every workgroup
    geometry of the chained decoder; three frames are also read back and decoded by the oracle"""
    import torch
    from lz4_flex_amd import _lib, workloads
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    n, size = 256, 512 * 1024
    plain = O.fixture_plain("compression_66k_JSON")
    src = workloads.json_tiles(plain, n * size, device=dev)
    info = fr.FrameInfo(block_mode=fr.BlockMode.Linked, block_size=fr.BlockSize.Max64KB)
    cap = int(lib.lz4flex_frame_compress_bound(size, info._c()))
    frames = torch.zeros(n * cap, dtype=torch.uint8, device=dev)
    back = torch.zeros(n * size, dtype=torch.uint8, device=dev)
    in_off = [i * size for i in range(n)]
    f_off = [i * cap for i in range(n)]
    flen, st = fr.compress_frames_device(src, in_off, [size] * n, info, frames, f_off, [cap] * n)
    assert st == [0] * n
    assert lib.lz4flex_set_tuning(None, b"decompress_variant", variant) == 0
    try:
        olen, st = fr.decompress_frames_device(frames, f_off, flen, back, in_off, [size] * n)
    finally:
        assert lib.lz4flex_set_tuning(None, b"decompress_variant", 0) == 0
    assert st == [0] * n and olen == [size] * n
    assert torch.equal(back, src)
    host = src.cpu().numpy().tobytes()
    for i in (0, 100, 255):
        f = frames[f_off[i]:f_off[i] + flen[i]].cpu().numpy().tobytes()
        rc, b, used = O.frame_decompress(f, size)
        assert rc == 0 and b == host[i * size:(i + 1) * size] and used == len(f)


def test_more_blocks_than_one_chained_call(fr):
    """35 000 Linked frames of two blocks each: 70 000 blocks, more than the 65 536 of one chained call -- the streams are dealt to
    several calls"""
    import torch
    from lz4_flex_amd import _lib, workloads
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    n, size = 35000, 65536 + 4096
    src = workloads.json_tiles(O.fixture_plain("compression_66k_JSON"), n * size, device=dev)
    info = fr.FrameInfo(block_mode=fr.BlockMode.Linked, block_size=fr.BlockSize.Max64KB)
    cap = int(lib.lz4flex_frame_compress_bound(size, info._c()))
    frames = torch.zeros(n * cap, dtype=torch.uint8, device=dev)
    back = torch.zeros(n * size, dtype=torch.uint8, device=dev)
    in_off = [i * size for i in range(n)]
    f_off = [i * cap for i in range(n)]
    flen, st = fr.compress_frames_device(src, in_off, [size] * n, info, frames, f_off, [cap] * n)
    assert st == [0] * n
    olen, st = fr.decompress_frames_device(frames, f_off, flen, back, in_off, [size] * n)
    assert st == [0] * n and olen == [size] * n
    assert torch.equal(back, src)


def test_many_short_linked_streams_level_by_level(fr):
    """round 6: from 1 024 Linked streams on, a call decodes block k of every stream in ONE launch (a plain batch with prefixes: the sequence
    decoder from 641 blocks on, src/frame/decompress.rs:195-222,280-305) instead of one workgroup per block polling its predecessor.  1 300
    frames WRITTEN BY THE REFERENCE ENCODER (the oracle's FrameEncoder: real back-references into the previous block), ragged lengths (1 byte
    ... 5 blocks: the levels thin out), incompressible stretches (stored blocks between compressed ones), some with checksums; then the same
    streams through this library's encoder; a few frames corrupted: the batch returns what the single-stream call returns"""
    rnd = random.Random(77)
    streams = _streams(77, 1300, 1, 330000)
    frames = []
    for i, s in enumerate(streams):
        rc, f = O.frame_compress(s, block_mode=1, block_size=4, block_checksums=i % 7 == 0, content_checksum=i % 11 == 0)
        assert rc == 0
        frames.append(f)
    back = fr.decompress_frames(frames, [len(s) for s in streams])
    assert back == streams
    info = fr.FrameInfo(block_mode=fr.BlockMode.Linked, block_size=fr.BlockSize.Max64KB)
    ours = fr.compress_frames(streams, info)
    assert fr.decompress_frames(ours, [len(s) + 7 for s in streams]) == streams
    for i in (3, 500, 1299):
        rc, b, used = O.frame_decompress(ours[i], len(streams[i]))
        assert rc == 0 and b == streams[i]
    # corrupted frames among the good ones: every stream's verdict is the single-stream call's
    bad = list(frames)
    hit = [5, 640, 641, 1200]
    for i in hit:
        f = bytearray(bad[i])
        at = rnd.randrange(len(f) // 2, len(f) - 1)
        f[at] ^= 0x5A
        bad[i] = bytes(f)
    res = fr.decompress_frames(bad, [len(s) for s in streams], return_errors=True)
    for i, r in enumerate(res):
        if i in hit:
            try:
                alone = fr.decompress_frames([bad[i]], [len(streams[i])])[0]
            except Exception as e:              # noqa: BLE001 -- the single-stream verdict, whatever its class
                assert isinstance(r, Exception) and type(r) is type(e) and str(r) == str(e), (i, r, e)
            else:
                assert r == alone
        else:
            assert r == streams[i], i


@pytest.mark.parametrize("n,size", [(4096, 256 * 1024), (1100, 3 * 65536 + 17)])
def test_device_resident_many_short_linked_streams(fr, n, size):
    """device-resident: thousands of Linked streams of a few blocks (the level-by-level path), compress_many -> decompress_many == the source;
    frames read back are decoded by the oracle"""
    import torch
    from lz4_flex_amd import _lib, workloads
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    plain = O.fixture_plain("compression_66k_JSON")
    src = workloads.json_tiles(plain, n * size, device=dev)
    info = fr.FrameInfo(block_mode=fr.BlockMode.Linked, block_size=fr.BlockSize.Max64KB)
    cap = int(lib.lz4flex_frame_compress_bound(size, info._c()))
    frames = torch.zeros(n * cap, dtype=torch.uint8, device=dev)
    back = torch.zeros(n * size, dtype=torch.uint8, device=dev)
    in_off = [i * size for i in range(n)]
    f_off = [i * cap for i in range(n)]
    flen, st = fr.compress_frames_device(src, in_off, [size] * n, info, frames, f_off, [cap] * n)
    assert st == [0] * n
    olen, st = fr.decompress_frames_device(frames, f_off, flen, back, in_off, [size] * n)
    assert st == [0] * n and olen == [size] * n
    assert torch.equal(back, src)
    host = src[:3 * size].cpu().numpy().tobytes()
    for i in (0, 2):
        f = frames[f_off[i]:f_off[i] + flen[i]].cpu().numpy().tobytes()
        rc, b, used = O.frame_decompress(f, size)
        assert rc == 0 and b == host[i * size:(i + 1) * size] and used == len(f)


@pytest.mark.parametrize("mode,bs", [(1, 4), (1, 5), (1, 7)])
def test_level_by_level_on_small_calls(fr, mode, bs):
    """"decompress_level_chains" 1: the level path on the ragged batches of test_round_trip_many_streams (37 streams of 1 ... 700 000 bytes,
    stored blocks, every block size: levels of a handful of blocks go to the workgroup decoder's prefix mode, larger ones to the sequence
    decoder's), frames of this library's encoder and of the reference's"""
    from lz4_flex_amd import _lib
    lib = _lib.load()
    streams = _streams(100 * mode + bs, 37, 1, 700000) + [b"", b"a", bytes(65536), bytes(65537), bytes(131072)]
    info = fr.FrameInfo(block_mode=fr.BlockMode(mode), block_size=fr.BlockSize(bs))
    frames = fr.compress_frames(streams, info)
    ref = [O.frame_compress(s, block_mode=mode, block_size=bs)[1] for s in streams]
    assert lib.lz4flex_set_tuning(None, b"decompress_level_chains", 1) == 0
    try:
        assert lib.lz4flex_get_tuning(None, b"decompress_level_chains") == 1
        assert fr.decompress_frames(frames, [len(s) for s in streams]) == streams
        assert fr.decompress_frames(ref, [len(s) + 1000 for s in streams]) == streams
    finally:
        assert lib.lz4flex_set_tuning(None, b"decompress_level_chains", 1024) == 0
