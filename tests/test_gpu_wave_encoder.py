"""GPU (-m gpu): the throughput ("wave") encoder, lz4_compress_wave.hip, through the C ABI.

Parity bar for this mode (BASELINE.json north_star, compress side): a valid LZ4 block that the reference's decoder
(oracle restatement) and C liblz4 decode to the identical input.  On top of that the kernel must reproduce its
scalar model (tests/sim/wave_encoder_model.c) byte for byte: same candidates, same parse, same bytes."""
import ctypes as C
import random

import numpy as np
import pytest

import corpus
import oracle_api as O
import wave_model as W

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def blk():
    from lz4_flex_amd import _lib, block
    lib = _lib.load()
    assert lib.lz4flex_device_count() >= 1
    assert lib.lz4flex_set_tuning(None, b"compress_mode", 0) == 0     # throughput mode on the default context
    return block


def SUB(n_blocks):
    """the sub-windows the library's default gives the blocks (of <= 64 KiB) of a batch of n_blocks (lz4_compress_wave.hip Item::sub)"""
    from lz4_flex_amd import _lib
    return W.auto_sub(n_blocks, _lib.load().lz4flex_get_tuning(None, b"compress_workgroups"))


def cases():
    rnd = random.Random(99)
    out = [b"", b"a", b"abcd" * 3, bytes(11), bytes(12), bytes(13), bytes(64), bytes(65), bytes(4096), bytes(100000)]
    out += list(corpus.ROUNDTRIP_STRINGS) + list(corpus.BUG_FUZZ)
    for stem in ("compression_1k", "compression_34k", "compression_65k", "compression_66k_JSON"):
        out.append(O.fixture_plain(stem))
    j = O.fixture_plain("compression_66k_JSON")
    for n in (63, 64, 127, 8191, 8192, 8193, 16384 + 5, 65535, 65536, 65537, 65536 + 11, 65536 + 12, 131072 + 77, 300000):
        out.append((j * 6)[:n])
    out.append(bytes(rnd.getrandbits(8) for _ in range(70000)))
    out.append(bytes(rnd.choice(b"ab") for _ in range(30000)))
    out.append(b"".join(bytes([rnd.getrandbits(8)]) * rnd.randint(1, 700) for _ in range(300)))
    out.append((bytes(range(256)) * 300)[:70001])
    out.append(bytes(rnd.getrandbits(8) for _ in range(20)) * 4000)       # period 20: every step full of same-bucket lanes
    # adjacent matches of one distance are merged inside an encode_seqs call (round 4): runs of every length around the 1 024-byte cap
    # and around 64 sequences per call, runs separated by single literals, long runs in a block of several windows
    out.append(b"".join(bytes([65 + k % 20]) * n for k, n in enumerate([1023, 1024, 1025, 2047, 2048, 2049, 5000, 8192, 8193, 20000])))
    out.append(b"".join(bytes([rnd.getrandbits(8)]) * rnd.choice([1100, 2100, 3000]) + bytes([rnd.getrandbits(8)]) for _ in range(40)))
    out.append(bytes(70000) + b"xyz" * 30000 + bytes(200000))
    out.append((b"0123456789abcdef" * 8192)[:131072 + 5])
    return out


@pytest.mark.parametrize("i", range(len(cases())))
def test_wave_encoder_scalar_call(blk, i):
    data = cases()[i]
    comp = blk.compress(data)                         # lz4flex_compress_into: one block through the persistent kernel
    assert O.decompress(comp, len(data)) == ("ok", bytes(data)), "not a valid LZ4 block for lz4_flex's decoder"
    if data:
        assert O.c_decompress(comp, len(data)) == bytes(data), "C liblz4 rejects the block"
    assert comp == W.compress(data), "kernel and scalar model disagree (len %d vs %d)" % (len(comp), len(W.compress(data)))


def test_wave_encoder_device_batch_many_blocks(blk):
    """more blocks than persistent workgroups, mixed lengths (multi-window blocks included), device-resident:
    every block == model, decodes with the oracle; the GPU decoder round-trips the batch"""
    import torch
    from lz4_flex_amd import _lib as L
    lib = L.load()
    rnd = random.Random(5)
    j = O.fixture_plain("compression_66k_JSON")
    t = O.fixture_plain("compression_65k")
    blocks = []
    for k in range(1500):
        src = j if k % 3 else t
        n = rnd.choice([0, 1, 100, 5000, 65536, 65536, 65536, 40000, 70000, 140000]) if k % 7 else 65536
        ph = rnd.randrange(len(src))
        blocks.append((src * 4)[ph:ph + n])
    in_len = np.array([len(b) for b in blocks], dtype=np.uint32)
    in_off = np.zeros(len(blocks), dtype=np.uint64)
    in_off[1:] = np.cumsum(in_len.astype(np.uint64) + 3)[:-1]          # unaligned block starts
    buf = np.zeros(int(in_off[-1] + in_len[-1]) + 64, dtype=np.uint8)
    for b, o in zip(blocks, in_off):
        buf[int(o):int(o) + len(b)] = np.frombuffer(b, dtype=np.uint8)
    cap = np.array([O.max_out(len(b)) for b in blocks], dtype=np.uint32)
    out_off = np.zeros(len(blocks), dtype=np.uint64)
    out_off[1:] = np.cumsum(cap.astype(np.uint64))[:-1]
    dev = torch.device("cuda", 0)
    d_in = torch.from_numpy(buf).to(dev)
    d_out = torch.zeros(int(out_off[-1] + cap[-1]), dtype=torch.uint8, device=dev)
    tt = lambda a, dt: torch.from_numpy(a.view(dt)).to(dev)
    d_in_off, d_in_len = tt(in_off, np.int64), tt(in_len, np.int32)
    d_out_off, d_cap = tt(out_off, np.int64), tt(cap, np.int32)
    d_len = torch.zeros(len(blocks), dtype=torch.int32, device=dev)
    d_st = torch.full((len(blocks),), -1, dtype=torch.int32, device=dev)
    ctx = C.c_void_p()
    assert lib.lz4flex_ctx_create(C.byref(ctx), 0) == 0
    assert lib.lz4flex_set_tuning(ctx, b"compress_mode", 0) == 0
    p = lambda x: C.c_void_p(x.data_ptr())
    for rep in range(2):                              # twice: the second launch reuses the workspace
        rc = lib.lz4flex_compress_batch(ctx, p(d_in), p(d_in_off), p(d_in_len), None, len(blocks), p(d_out), p(d_out_off),
                                        p(d_cap), p(d_len), p(d_st), L.MEM_DEVICE, None)
        assert rc == 0, L.last_error()
        torch.cuda.synchronize()
        assert int((d_st != 0).sum().item()) == 0
    h_out, h_len = d_out.cpu().numpy(), d_len.cpu().numpy()
    for k in range(0, len(blocks), 1):
        got = bytes(h_out[int(out_off[k]):int(out_off[k]) + int(h_len[k])])
        if k % 10 == 0 or len(blocks[k]) > 65536:
            assert got == W.compress(blocks[k], sub=SUB(len(blocks))), (k, len(blocks[k]))
        assert O.decompress(got, len(blocks[k])) == ("ok", blocks[k]), k
    # GPU decoder on the GPU encoder's output
    d_back = torch.zeros_like(d_in)
    d_blen = torch.zeros(len(blocks), dtype=torch.int32, device=dev)
    d_bst = torch.full((len(blocks),), -1, dtype=torch.int32, device=dev)
    rc = lib.lz4flex_decompress_batch(ctx, p(d_out), p(d_out_off), p(d_len), len(blocks), p(d_back), p(d_in_off), p(d_in_len),
                                      p(d_blen), p(d_bst), None, L.MEM_DEVICE, None)
    assert rc == 0
    torch.cuda.synchronize()
    assert int((d_bst != 0).sum().item()) == 0 and torch.equal(d_blen, d_in_len)
    hb = d_back.cpu().numpy()
    for k in range(len(blocks)):
        assert bytes(hb[int(in_off[k]):int(in_off[k]) + len(blocks[k])]) == blocks[k], k
    lib.lz4flex_ctx_destroy(ctx)


def test_wave_encoder_output_too_small(blk):
    """compress.rs:338-340: OutputTooSmall is decided up front from get_maximum_output_size, nothing is written"""
    from lz4_flex_amd import _lib as L
    data = O.fixture_plain("compression_1k")
    out = bytearray(b"\xEE" * (O.max_out(len(data)) - 1))
    with pytest.raises(blk.CompressOutputTooSmall):
        blk.compress_into(data, out)
    assert bytes(out) == b"\xEE" * len(out)
    # per-block status of a batch
    lib = L.load()
    src = np.frombuffer(data * 2, dtype=np.uint8).copy()
    outb = np.full(4096, 0xEE, dtype=np.uint8)
    ol, st = blk.compress_batch(src, [0, len(data)], [len(data), len(data)], outb, [0, 2048], [10, 2048])
    assert st.tolist() == [L.E_OUTPUT_TOO_SMALL, 0] and bytes(outb[:10]) == b"\xEE" * 10
    assert O.decompress(bytes(outb[2048:2048 + int(ol[1])]), len(data)) == ("ok", data)


@pytest.mark.parametrize("carry_wait", [1, 0])
def test_wave_encoder_few_large_blocks_window_mode(blk, carry_wait):
    """fewer blocks than persistent workgroups: the windows of a block are dealt to different workgroups and the output position
    travels between them through the workspace.  Ragged sizes (one window ... 90 windows), an empty block, a block whose sink
    is too small in the middle; every block == model, decodes with the oracle; twice (the carry ring and counters are reset).
    carry_wait 0: a window that has to wait for its predecessor gives up at once (what a time-sliced GPU does to the bounded
    wait) -- the blocks it poisons are encoded again by the second launch: same bytes, same statuses, no device error"""
    from lz4_flex_amd import _lib as L
    lib = L.load()
    assert lib.lz4flex_set_tuning(None, b"compress_carry_wait", carry_wait) == 0
    try:
        _window_mode_batch(blk, L)
    finally:
        assert lib.lz4flex_set_tuning(None, b"compress_carry_wait", 1) == 0


def _window_mode_batch(blk, L):
    rnd = random.Random(17)
    j = O.fixture_plain("compression_66k_JSON")
    t = O.fixture_plain("compression_65k")
    sizes = [5 * 1048576 + 123, 70000, 0, 3 * 65536, 1500000, 65536, 12, 90 * 65536 + 7, 200001]
    blocks = []
    for k, n in enumerate(sizes):
        src = (j if k % 2 else t)
        ph = rnd.randrange(len(src))
        reps = n // len(src) + 3
        blocks.append((src * reps)[ph:ph + n])
    src_buf = np.frombuffer(b"".join(blocks), dtype=np.uint8).copy()
    in_len = [len(b) for b in blocks]
    in_off = [int(x) for x in np.concatenate([[0], np.cumsum(in_len)[:-1]])]
    cap = [O.max_out(n) for n in in_len]
    cap[4] = 1000                                                      # OutputTooSmall for block 4, decided up front
    out_off = [int(x) for x in np.concatenate([[0], np.cumsum(cap)[:-1]])]
    for rep in range(2):
        outb = np.full(sum(cap) + 64, 0xEE, dtype=np.uint8)
        ol, st = blk.compress_batch(src_buf, in_off, in_len, outb, out_off, cap)
        assert st.tolist() == [0, 0, 0, 0, L.E_OUTPUT_TOO_SMALL, 0, 0, 0, 0]
        for k, b in enumerate(blocks):
            if k == 4:
                assert bytes(outb[out_off[k]:out_off[k] + cap[k]]) == b"\xEE" * cap[k]
                continue
            got = bytes(outb[out_off[k]:out_off[k] + int(ol[k])])
            assert got == W.compress(b, sub=SUB(len(blocks))), (k, len(b))
            assert O.decompress(got, len(b)) == ("ok", b), k


def test_wave_encoder_heads_at_a_window_end(blk):
    """found by tools/gpu_fuzz.py (seed 31): a 139 612-byte block whose second window ends in positions that are heads -- their
    4 bytes reach past the window -- and holds 129 heads in its last 256 positions: the model halved that superstep, the kernel
    (zeros behind the window in LDS: three heads fewer) did not and cut a long match at 84 bytes.  Valid either way; the
    kernel must equal its model.  The window's slack now holds the block's next bytes."""
    import os
    import zlib
    here = os.path.dirname(os.path.abspath(__file__))
    with open(os.path.join(here, "golden", "fuzz_window_end_heads.zlib"), "rb") as f:
        d = zlib.decompress(f.read())
    assert len(d) == 139612
    from lz4_flex_amd import _lib as L
    lib = L.load()
    try:
        for slide in (0, 1, 2):       # windows advancing by 64 KiB (where the fuzzer found it), by 32 KiB and by 48 KiB (the default for long blocks)
            assert lib.lz4flex_set_tuning(None, b"compress_sliding_window", slide) == 0
            c = blk.compress(d)
            assert O.decompress(c, len(d)) == ("ok", d)
            assert c == W.compress(d, slide=slide)
    finally:
        assert lib.lz4flex_set_tuning(None, b"compress_sliding_window", W.SLIDE_DEFAULT) == 0


def test_wave_encoder_sliding_windows_off(blk):
    """"compress_sliding_window" 0: the windows of a long block advance by 64 KiB (round 3's bytes); blocks of every length class
    == model with slide 0, decoded by the oracle"""
    from lz4_flex_amd import _lib as L
    lib = L.load()
    j = O.fixture_plain("compression_66k_JSON")
    assert lib.lz4flex_set_tuning(None, b"compress_sliding_window", 0) == 0
    try:
        for n in (65536, 65537, 66675, 98304, 131072 + 77, 300000, 1048576 + 5):
            d = (j * 17)[:n]
            c = blk.compress(d)
            assert O.decompress(c, n) == ("ok", d)
            assert c == W.compress(d, slide=0), n
    finally:
        assert lib.lz4flex_set_tuning(None, b"compress_sliding_window", W.SLIDE_DEFAULT) == 0


def test_wave_encoder_window_strides(blk):
    """"compress_sliding_window" 1 / 2: the windows of a long block advance by 32 / 48 KiB; blocks of every length class == model with
    that setting, decoded by the oracle (0: the test above)"""
    from lz4_flex_amd import _lib as L
    lib = L.load()
    j = O.fixture_plain("compression_66k_JSON")
    t = O.fixture_plain("compression_65k")
    try:
        for slide in (1, 2):
            assert lib.lz4flex_set_tuning(None, b"compress_sliding_window", slide) == 0
            for n in (65537, 66675, 98304, 114688, 114689, 131072 + 77, 163840, 300000, 1048576 + 5, 4 << 20):
                for src in (j, t):
                    d = (src * (n // len(src) + 2))[:n]
                    c = blk.compress(d)
                    assert O.decompress(c, n) == ("ok", d)
                    assert c == W.compress(d, slide=slide), (slide, n)
    finally:
        assert lib.lz4flex_set_tuning(None, b"compress_sliding_window", W.SLIDE_DEFAULT) == 0


def _history_batch(blk, L, sizes, seed, stream=None):
    """blocks cut from ONE stream; flags promise the bytes in front of a block as history (LZ4FLEX_BLOCK_HISTORY): 0, fewer than
    the encoder uses (ignored), exactly HIST, more.  Every block == model (which gets the same 32 KiB in front), and the oracle
    decodes it behind the previous 64 KiB of the stream as dictionary.  stream: given (then cut into 64 KiB blocks), else drawn"""
    rnd = random.Random(seed)
    j = O.fixture_plain("compression_66k_JSON")
    t = O.fixture_plain("compression_65k")
    if stream is not None:
        sizes = [min(65536, len(stream) - o) for o in range(0, len(stream), 65536)]
    total = sum(sizes)
    if stream is None:
        stream = b""
        while len(stream) < total:
            src = j if rnd.random() < 0.6 else t
            ph = rnd.randrange(len(src))
            stream += (src * 2)[ph:ph + rnd.choice([3000, 20000, 66000, 100000])]
    stream = stream[:total]
    in_len = list(sizes)
    in_off = [int(x) for x in np.concatenate([[0], np.cumsum(in_len)[:-1]])]
    hist = []
    for k, o in enumerate(in_off):
        h = [0, 1000, 32767, 32768, 32768, 40000, 65536, 65536][k % 8]
        hist.append(min(h, o))
    flags = [(h << 8) | (k % 4) for k, h in enumerate(hist)]              # (the low bits mean nothing to the throughput encoder)
    cap = [O.max_out(n) for n in in_len]
    out_off = [int(x) for x in np.concatenate([[0], np.cumsum(cap)[:-1]])]
    src_buf = np.frombuffer(stream, dtype=np.uint8).copy()
    outb = np.full(sum(cap) + 64, 0xEE, dtype=np.uint8)
    ol, st = blk.compress_batch(src_buf, in_off, in_len, outb, out_off, cap, flags=flags)
    assert not st.any()
    used = 0
    step = max(1, len(sizes) // 150)
    for k, (o, n) in enumerate(zip(in_off, in_len)):
        got = bytes(outb[out_off[k]:out_off[k] + int(ol[k])])
        b = stream[o:o + n]
        with_h = hist[k] >= W.HIST and n > 0
        used += with_h
        if k % step == 0 or n > 65536:
            want = W.compress(stream[o - W.HIST:o + n], hist=W.HIST) if with_h else W.compress(b, sub=SUB(len(sizes)))
            assert got == want, (k, o, n, hist[k])
        assert O.decompress(got, n, dict_data=stream[max(0, o - 65536):o] if with_h else None) == ("ok", b), (k, o, n)
    assert used >= len(sizes) // 3
    return stream, in_off, in_len, ol, outb, out_off


@pytest.mark.parametrize("carry_wait", [1, 0])
def test_wave_encoder_history_few_blocks(blk, carry_wait):
    """fewer blocks than workgroups (windows dealt out, carries through the workspace; carry_wait 0: through the second launch)"""
    from lz4_flex_amd import _lib as L
    lib = L.load()
    sizes = [65536, 65536, 65536, 1, 11, 12, 13, 100, 65536, 40000, 70000, 262144, 65535, 65537, 98304, 32768, 32769, 1048576 + 5, 65536, 5, 65536,
             4 * 1048576, 65536, 131072, 65536, 1000]
    assert lib.lz4flex_set_tuning(None, b"compress_carry_wait", carry_wait) == 0
    try:
        _history_batch(blk, L, sizes, 31)
    finally:
        assert lib.lz4flex_set_tuning(None, b"compress_carry_wait", 1) == 0


def test_wave_encoder_history_many_blocks(blk):
    """more blocks than workgroups (a workgroup walks the windows of its blocks in order, the carry stays in LDS)"""
    from lz4_flex_amd import _lib as L
    rnd = random.Random(8)
    sizes = [rnd.choice([65536, 65536, 65536, 16384, 50000, 70000, 131072]) for _ in range(1400)]
    _history_batch(blk, L, sizes, 32)


def test_history_must_lie_inside_the_input(blk):
    """a host batch whose flags promise more bytes than lie in front of the block is refused (-E_INVALID_ARG)"""
    from lz4_flex_amd import _lib as L, block
    src = np.zeros(70000, dtype=np.uint8)
    out = np.zeros(O.max_out(65536) + 64, dtype=np.uint8)
    with pytest.raises(block.DeviceError):
        blk.compress_batch(src, [1000], [65536], out, [0], [O.max_out(65536)], flags=[40000 << 8])


@pytest.mark.parametrize("carry_wait", [1, 0])
@pytest.mark.parametrize("setting,n_blocks", [(0, 1), (0, 100), (0, 160), (0, 200), (0, 300), (1, 40), (2, 40), (3, 40), (4, 40), (4, 700), (3, 600)])
def test_wave_encoder_subwindows(blk, setting, n_blocks, carry_wait):
    """small batches (round 5): a block of at most 64 KiB is cut into 2 or 4 sub-windows that different workgroups encode side by side, the
    output position travelling between them like between the windows of a long block.  "compress_subwindows" 0 (by batch size: 1 block and
    100 blocks -> 4, 160 -> 3, 200 -> 2, 300 -> 1 with 512 workgroups), 1 / 2 / 3 / 4 forced (700 blocks: more blocks than workgroups, a workgroup walks
    its blocks' sub-windows itself).  Ragged lengths around every quarter, text / JSON / noise / runs; every block == the scalar model with
    that many sub-windows, decodes with the oracle and with liblz4; carry_wait 0: every waiting sub-window gives up, the second launch
    encodes the block again to the same bytes"""
    from lz4_flex_amd import _lib as L
    lib = L.load()
    rnd = random.Random(1000 * setting + n_blocks)
    j, t = O.fixture_plain("compression_66k_JSON"), O.fixture_plain("compression_65k")
    noise = bytes(rnd.getrandbits(8) for _ in range(70000))
    lens = [65536, 65536, 65535, 49152, 49153, 32768, 32769, 32767, 16384, 16385, 16383, 40000, 20000, 1000, 12, 0, 65536, 60001, 33000, 70000, 65537, 22016, 22017, 44032, 44033]
    blocks = []
    for k in range(n_blocks):
        src = (j, t, j, noise, bytes(70000), j)[k % 6]
        n = lens[k % len(lens)] if k % 5 else 65536
        ph = rnd.randrange(len(src))
        blocks.append((src * 3)[ph:ph + n])
    sub = setting if setting else SUB(n_blocks)
    assert lib.lz4flex_set_tuning(None, b"compress_subwindows", setting) == 0
    assert lib.lz4flex_set_tuning(None, b"compress_carry_wait", carry_wait) == 0
    try:
        src_buf = np.frombuffer(b"".join(blocks) + b"\0", dtype=np.uint8).copy()
        in_len = [len(b) for b in blocks]
        in_off = [int(x) for x in np.concatenate([[0], np.cumsum(in_len)[:-1]])]
        cap = [O.max_out(n) for n in in_len]
        out_off = [int(x) for x in np.concatenate([[0], np.cumsum(cap)[:-1]])]
        for rep in range(2):
            outb = np.full(sum(cap) + 64, 0xEE, dtype=np.uint8)
            ol, st = blk.compress_batch(src_buf, in_off, in_len, outb, out_off, cap)
            assert not st.any()
            for k, b in enumerate(blocks):
                got = bytes(outb[out_off[k]:out_off[k] + int(ol[k])])
                if k < 64 or k % 9 == 0:
                    assert got == W.compress(b, sub=sub), (k, len(b), sub)
                assert O.decompress(got, len(b)) == ("ok", b), k
                if b and k % 4 == 0:
                    assert O.c_decompress(got, len(b)) == b, k
    finally:
        assert lib.lz4flex_set_tuning(None, b"compress_subwindows", 0) == 0
        assert lib.lz4flex_set_tuning(None, b"compress_carry_wait", 1) == 0


def _run_window_inputs():
    rnd = random.Random(23)
    j = O.fixture_plain("compression_66k_JSON")
    out = []
    for n in (8191, 8192, 8193, 8192 + 11, 8192 + 12, 8192 + 16, 20000, 65535, 65536, 65537, 65536 + 8191, 65536 + 8192, 65536 + 8203, 131072, 131072 + 5,
              300001, 1048576, 4194304):
        out.append(bytes(n))
    out.append(b"\x07" * 70000)
    out.append(bytes(65536) + b"\x01" * 65536 + bytes(65536))                 # three run windows of different bytes (64 KiB stride), mixed windows with the default stride
    out.append(bytes(70000) + j + bytes(200000))                              # runs, data, runs
    out.append(j[:5000] + bytes(300000) + j[:77])                             # a run that starts and ends inside windows
    out.append(bytes(65536 * 2) + b"\x01")                                    # the run ends with the block's last byte
    out.append(b"\x01" + bytes(65536 * 2))
    out.append(bytes(1024) + b"\x01" + bytes(65536))                          # the first KiB is a run, the window is not
    out.append(bytes(65535) + b"\x01" + bytes(65536))                         # ... the window's last byte differs
    out.append(bytes(40000) + bytes([rnd.getrandbits(8) for _ in range(3)]) + bytes(100000))
    return out


@pytest.mark.parametrize("slide", [2, 0, 1])
def test_wave_encoder_run_windows(blk, slide):
    """RUN WINDOWS (round 6; src/block/compress.rs:156-216: the reference's count_same_bytes is unbounded, 30 000 zeros are one match): a window
    of >= 8 KiB that is one byte repeated becomes ONE sequence.  Scalar calls (window mode, sub-windows for blocks of <= 64 KiB) and one
    batch of them with every window stride: kernel == model, decoded by the oracle and liblz4; 4 MiB of zeros compress to <= 0.45 %"""
    from lz4_flex_amd import _lib as L
    lib = L.load()
    inputs = _run_window_inputs()
    assert lib.lz4flex_set_tuning(None, b"compress_sliding_window", slide) == 0
    try:
        for d in inputs:
            c = blk.compress(d)
            assert O.decompress(c, len(d)) == ("ok", d), len(d)
            assert O.c_decompress(c, len(d)) == d
            assert c == W.compress(d, slide=slide if len(d) > 65536 else 0), (len(d), len(c))
            if d == bytes(4194304):
                assert len(c) <= 0.0045 * len(d)
        src_buf = np.frombuffer(b"".join(inputs), dtype=np.uint8).copy()
        in_len = [len(b) for b in inputs]
        in_off = [int(x) for x in np.concatenate([[0], np.cumsum(in_len)[:-1]])]
        cap = [O.max_out(n) for n in in_len]
        out_off = [int(x) for x in np.concatenate([[0], np.cumsum(cap)[:-1]])]
        outb = np.zeros(sum(cap) + 64, dtype=np.uint8)
        ol, st = blk.compress_batch(src_buf, in_off, in_len, outb, out_off, cap)
        assert not st.any()
        for k, d in enumerate(inputs):
            got = bytes(outb[out_off[k]:out_off[k] + int(ol[k])])
            assert got == W.compress(d, slide=slide if len(d) > 65536 else 0, sub=SUB(len(inputs))), (k, len(d))
        back = np.zeros(len(src_buf), dtype=np.uint8)
        dl, dst, _ = blk.decompress_batch(outb, out_off, ol, back, in_off, in_len)
        assert not dst.any() and (back == src_buf).all()
    finally:
        assert lib.lz4flex_set_tuning(None, b"compress_sliding_window", W.SLIDE_DEFAULT) == 0


def test_wave_encoder_run_windows_many_blocks_and_history(blk):
    """run windows in block mode (more blocks than workgroups: 64 KiB blocks of zeros, of one other byte, of JSON, alternating) and
    behind history (a Linked frame's blocks, LZ4FLEX_BLOCK_HISTORY): kernel == model, the oracle decodes every block"""
    j = O.fixture_plain("compression_66k_JSON")
    kinds = [bytes(65536), (j * 2)[100:100 + 65536], b"\xAA" * 65536, bytes(30000) + j[:35536], bytes(65536 - 9) + b"tail bytes"[:9]]
    blocks = [kinds[k % len(kinds)] for k in range(1100)]
    src_buf = np.frombuffer(b"".join(blocks), dtype=np.uint8).copy()
    in_len = [len(b) for b in blocks]
    in_off = [int(x) for x in np.concatenate([[0], np.cumsum(in_len)[:-1]])]
    cap = [O.max_out(n) for n in in_len]
    out_off = [int(x) for x in np.concatenate([[0], np.cumsum(cap)[:-1]])]
    outb = np.zeros(sum(cap) + 64, dtype=np.uint8)
    ol, st = blk.compress_batch(src_buf, in_off, in_len, outb, out_off, cap)
    assert not st.any()
    models = [W.compress(d, sub=SUB(len(blocks))) for d in kinds]
    for k, d in enumerate(blocks):
        got = bytes(outb[out_off[k]:out_off[k] + int(ol[k])])
        assert got == models[k % len(kinds)], k
        if k < 10:
            assert O.decompress(got, len(d)) == ("ok", d)
    assert len(models[0]) <= 0.0045 * 65536
    # history: one stream of zeros / JSON / zeros cut into 64 KiB blocks, every block with the 32 KiB in front of it as history
    from lz4_flex_amd import _lib as L
    stream = bytes(3 * 65536 + 100) + (j * 3)[:2 * 65536] + b"\x33" * (4 * 65536)
    _history_batch(blk, L, None, 0, stream=stream)
