"""GPU (-m gpu): the sequence decoder (lz4_flex_amd/csrc/lz4_decompress_seq.hip, decompress_variant 13: one block per wavefront, one lane
per sequence) on ITS OWN boundaries -- the generic decoder matrix of test_gpu_block.py (KATs, adversarial batch, every prefix / corruption,
synthetic dependency blocks), test_gpu_pcd.py (large blocks) and test_gpu_dispatch_matrix.py (the thresholds +- 1) runs it as DECODERS -13;
here are blocks WRITTEN sequence by sequence to sit on the kernel's geometry:
  a lane copies literal runs <= 64 bytes, matches <= 273 bytes that do not overlap their source, far matches <= 64 bytes; anything else is
  executed alone by the wavefront, runs of >= 1 KiB memory to memory (periodic form for offsets < 1 KiB); tiles of 3 840 compressed bytes
  in 64 parts of 60; chunks of <= 64 sequences and <= 1 120 output bytes; a 3 584-byte window that slides by keeping 1 280 bytes.
Checker: the oracle (lz4_flex's decoder restated), which also says what an invalid variant of each block is (status, OutputTooSmall detail).
With the second pass off, the kernel must have decoded every valid block ITSELF (a silent fall-back to the reference-order kernel would pass
every other test)."""
import ctypes as C
import random

import numpy as np
import pytest

import oracle_api as O

pytestmark = pytest.mark.gpu
REDO = 0x7F000001


@pytest.fixture(scope="module")
def env():
    from lz4_flex_amd import _lib, block
    lib = _lib.load()
    assert lib.lz4flex_device_count() >= 1
    return lib, block


def _ctx(lib, second_pass=1):
    ctx = C.c_void_p()
    assert lib.lz4flex_ctx_create(C.byref(ctx), -1) == 0
    assert lib.lz4flex_set_tuning(ctx, b"decompress_variant", 13) == 0
    assert lib.lz4flex_set_tuning(ctx, b"decompress_second_pass", second_pass) == 0
    return ctx


def _batch(block, ctx, comps, caps, slack=64, misalign=0):
    inb = np.frombuffer(bytes(misalign) + b"".join(comps) + bytes(64), dtype=np.uint8)
    in_len = [len(c) for c in comps]
    in_off = (np.concatenate([[0], np.cumsum(in_len[:-1], dtype=np.uint64)]) + misalign).astype(np.uint64)
    out_off = (np.concatenate([[0], np.cumsum([k + slack for k in caps[:-1]], dtype=np.uint64)]) + misalign).astype(np.uint64)
    out = np.full(int(out_off[-1]) + caps[-1] + slack, 0xA5, dtype=np.uint8)
    ol, st, det = block.decompress_batch(inb, in_off, in_len, out, out_off, caps, ctx=ctx)
    return out, out_off, ol, st, det


class Writer:
    """an LZ4 block, sequence by sequence (src/block/compress.rs:463-487 is the layout), with the plain text the format's byte-wise
    semantics give (decompress_safe.rs:93-247)"""

    def __init__(self, seed=1):
        self.comp, self.out, self.rnd = bytearray(), bytearray(), random.Random(seed)

    def _len(self, v):
        while v >= 255:
            self.comp.append(255)
            v -= 255
        self.comp.append(v)

    def seq(self, lit, off, ml):
        lits = bytes(self.rnd.getrandbits(8) for _ in range(lit)) if isinstance(lit, int) else bytes(lit)
        self.comp.append((min(len(lits), 15) << 4) | min(ml - 4, 15))
        if len(lits) >= 15:
            self._len(len(lits) - 15)
        self.comp += lits
        self.out += lits
        assert 1 <= off <= len(self.out) and off <= 65535 and ml >= 4, (off, len(self.out), ml)
        self.comp += bytes((off & 0xFF, off >> 8))
        if ml - 4 >= 15:
            self._len(ml - 19)
        start = len(self.out) - off
        for k in range(ml):
            self.out.append(self.out[start + k])
        return self

    def end(self, lit=5):
        lits = bytes(self.rnd.getrandbits(8) for _ in range(lit))
        self.comp.append(min(lit, 15) << 4)
        if lit >= 15:
            self._len(lit - 15)
        self.comp += lits
        self.out += lits
        return bytes(self.comp), bytes(self.out)


def _blocks():
    rnd = random.Random(606)
    out = []

    def add(name, w, tail=5):
        c, p = w.end(tail)
        st, got = O.decompress(c, len(p))
        assert st == "ok" and got == p, name            # the writer and the oracle agree on what the block says
        out.append((name, c, p))

    # ---- a lane's limits: literal runs 63 .. 66, matches 272 .. 275 (273 = 19 + 254: one length byte), 16 / 17 / 32 / 33 / 48 / 49 bytes (the pieces)
    for lit in (0, 1, 2, 3, 4, 7, 8, 15, 16, 17, 31, 32, 33, 47, 48, 49, 63, 64, 65, 66, 200, 214, 215, 216, 269, 270, 271, 300):
        w = Writer(lit)
        w.seq(20, 7, 9)
        for _ in range(70):
            w.seq(lit, rnd.randint(24, len(w.out)), rnd.randint(4, 30))
        add("literal runs of %d" % lit, w)
    for ml in (4, 5, 7, 8, 9, 15, 16, 17, 18, 19, 20, 31, 32, 33, 34, 47, 48, 49, 50, 63, 64, 65, 128, 272, 273, 274, 275, 528, 529, 1023, 1024, 1025, 3000):
        w = Writer(ml)
        w.seq(4000, 1000, 50)
        for _ in range(70):
            w.seq(rnd.randint(0, 3), rnd.randint(ml, min(len(w.out), 3000)), ml)       # never its own output: a lane's match
        add("matches of %d" % ml, w)
    # ---- far and near: offsets around what the window holds (1 280 .. 3 584 bytes back), far matches of 63 .. 66 bytes and longer
    for ml in (4, 15, 16, 17, 33, 49, 63, 64, 65, 66, 100, 273, 274):
        w = Writer(1000 + ml)
        w.seq(9000, 5000, 40)
        for k in range(140):
            off = rnd.choice((1100, 1279, 1280, 1281, 1296, 2000, 2303, 2304, 2305, 3500, 3583, 3584, 3585, 3600, 5000, 8000, len(w.out)))
            w.seq(rnd.randint(0, 5), max(ml, min(off, len(w.out))), ml)
        add("far / near matches of %d" % ml, w)
    # ---- matches that read their own output (the wavefront's path, periodic form): every small offset, lengths short and long
    for off in (1, 2, 3, 4, 5, 7, 8, 15, 16, 17, 31, 32, 33, 63, 64, 65, 100, 255, 256, 1000, 1023, 1024, 1025, 2000):
        w = Writer(2000 + off)
        w.seq(max(off, 30), off, 4)
        for ml in (5, 17, 40, off + 1, 2 * off + 3, 300, 1023, 1024, 1025, 4000, 20000):
            w.seq(rnd.randint(0, 20), off, max(ml, 4))
            w.seq(3, rnd.randint(1, min(len(w.out), 60000)), 6)
        add("own-output matches, offset %d" % off, w)
    # ---- long literal runs (memory to memory from 1 KiB on) followed by matches into them, at their ends, across them
    for run in (1000, 1023, 1024, 1025, 2048, 3583, 3584, 3585, 5000, 70000):
        w = Writer(3000 + run)
        w.seq(run, 1, 4)
        w.seq(0, run // 2, 40).seq(0, min(run + 44, 65535), 30).seq(2, 5, 4).seq(run, min(run, 65535), 50).seq(0, 51, 4)
        add("literal runs of %d" % run, w)
    # ---- tiles and parts: sequences of 3 bytes (1 280 per tile: the token list's capacity), tokens on the last byte of a tile, sequences that jump
    # over whole parts and whole tiles (literal runs inside the compressed stream), length bytes that straddle a tile's end
    w = Writer(41)
    w.seq(40, 20, 4)
    for _ in range(6000):
        w.seq(0, rnd.randint(4, 40), 4)
    add("6 000 three-byte sequences", w)
    for shift in range(0, 64, 3):
        w = Writer(500 + shift)
        w.seq(3800 + shift, 100, 12)
        for _ in range(300):
            w.seq(rnd.choice((0, 0, 1, 2, 60, 61, 120, 250, 300)), rnd.randint(4, 3000), rnd.choice((4, 19, 20, 273, 274)))
        add("tile boundary shifted by %d" % shift, w)
    w = Writer(43)
    w.seq(10, 3, 5)
    for _ in range(40):
        w.seq(rnd.choice((3839, 3840, 3841, 7680, 8000)), rnd.randint(16, 2000), rnd.randint(4, 40))
    add("literal runs that jump over tiles", w)
    # ---- chunks: 64 sequences of many bytes each (the 1 120-byte budget cuts), dense dependencies (every match reads the one before it)
    w = Writer(44)
    w.seq(600, 300, 100)
    for _ in range(500):
        w.seq(rnd.randint(0, 2), rnd.choice((100, 101, 150, 273)), rnd.choice((100, 150, 273)))
    add("long matches: chunks cut by bytes", w)
    w = Writer(45)
    w.seq(64, 30, 8)
    for _ in range(4000):
        ml = rnd.randint(4, 24)
        w.seq(rnd.choice((0, 0, 0, 1)), rnd.randint(ml, ml + 30), ml)                 # the source ends within ~ 30 bytes of the destination
    add("chains: every match reads its neighbours' output", w)
    # ---- the block's end: last literals of 0 .. 70 bytes (a lane's or the wavefront's), a block that is one literal run
    for tail in (0, 1, 5, 14, 15, 16, 63, 64, 65, 70, 300):
        w = Writer(600 + tail)
        w.seq(30, 9, 14).seq(1, 20, 5)
        add("last literals %d" % tail, w, tail)
    for n in (0, 1, 12, 13, 64, 65, 1023, 1024, 5000):
        w = Writer(700 + n)
        c, p = w.end(n)
        out.append(("literals only %d" % n, c, p))
    return out


def test_blocks_on_the_decoders_own_boundaries(env):
    """every block above: bytes == oracle with an exact sink, with a sink 777 bytes larger, nothing behind either; a sink 1 / 40 bytes short and a
    block cut short end like the oracle says (status, OutputTooSmall detail) -- through the second pass, which is the reference-order kernel"""
    lib, block = env
    blocks = _blocks()
    comps, caps, want = [], [], []
    for name, c, p in blocks:
        for cap in (len(p), len(p) + 777):
            comps.append(c); caps.append(cap); want.append((name, ("ok", p)))
        for cap in (max(len(p) - 1, 0), max(len(p) - 40, 0)):
            comps.append(c); caps.append(cap); want.append((name + " (short sink)", O.decompress(c, cap)))
        comps.append(c[:-3]); caps.append(len(p)); want.append((name + " (cut)", O.decompress(c[:-3], len(p))))
    for misalign in (0, 5):                         # the batch's buffers at an odd address: every block's input and output are unaligned
        ctx = _ctx(lib)
        try:
            out, out_off, ol, st, det = _batch(block, ctx, comps, caps, misalign=misalign)
        finally:
            lib.lz4flex_ctx_destroy(ctx)
        for i, (name, w) in enumerate(want):
            o = int(out_off[i])
            if w[0] == "ok":
                assert st[i] == 0 and ol[i] == len(w[1]), (name, i, int(st[i]), int(ol[i]), len(w[1]))
                assert out[o:o + len(w[1])].tobytes() == w[1], "%s: bytes differ" % name
                assert out[o + len(w[1]):o + caps[i] + 64].tobytes() == b"\xA5" * (caps[i] - len(w[1]) + 64), "%s: wrote behind its end" % name
            else:
                assert O.ERR_NAMES.get(int(st[i])) == w[0], (name, int(st[i]), w[0])
                if w[0] == "OutputTooSmall":
                    assert (int(det[i][0]), int(det[i][1])) == tuple(w[1]), (name, det[i], w[1])
                assert out[o + caps[i]:o + caps[i] + 64].tobytes() == b"\xA5" * 64, "%s: wrote behind its sink" % name


def test_the_kernel_decodes_every_valid_block_itself(env):
    """second pass off: a valid block must come out of THIS kernel (status 0, the oracle's bytes), an invalid one must be left marked
    (0x7F000001) -- nothing is quietly handed to the reference-order kernel"""
    lib, block = env
    blocks = _blocks()
    comps = [c for _, c, _ in blocks] + [c[:-2] for _, c, p in blocks if len(c) > 8]
    caps = [len(p) for _, _, p in blocks] + [len(p) for _, c, p in blocks if len(c) > 8]
    plains = [p for _, _, p in blocks]
    ctx = _ctx(lib, second_pass=0)
    try:
        out, out_off, ol, st, det = _batch(block, ctx, comps, caps)
    finally:
        lib.lz4flex_ctx_destroy(ctx)
    for i, p in enumerate(plains):
        o = int(out_off[i])
        assert st[i] == 0 and ol[i] == len(p) and out[o:o + len(p)].tobytes() == p, (blocks[i][0], int(st[i]), int(ol[i]), len(p))
    for i in range(len(plains), len(comps)):
        verdict = O.decompress(comps[i], caps[i])
        assert (st[i] == 0) == (verdict[0] == "ok"), (i, int(st[i]), verdict[0])
        assert st[i] in (0, REDO), (i, int(st[i]))


def test_a_medium_batch_through_the_default_dispatch(env):
    """1 500 blocks (the default dispatch's range for this decoder: 641 ... 14 336) of every kind above and of both encoders' JSON / text tiles, one
    launch, `decompress_variant` 0: == oracle"""
    import wave_model as W
    lib, block = env
    rnd = random.Random(9)
    j, t = O.fixture_plain("compression_66k_JSON"), O.fixture_plain("compression_65k")
    pool = [(c, p) for _, c, p in _blocks() if len(p) <= 200000]
    for k in range(40):
        src = (j if k % 2 else t) * 3
        ph = rnd.randrange(len(src) // 3)
        p = src[ph:ph + rnd.choice((65536, 65536, 30000, 1000))]
        pool.append(((O.compress if k % 3 else W.compress)(p), p))
    picks = [pool[rnd.randrange(len(pool))] for _ in range(1500)]
    ctx = C.c_void_p()
    assert lib.lz4flex_ctx_create(C.byref(ctx), -1) == 0
    try:
        assert 641 <= len(picks) <= lib.lz4flex_get_tuning(ctx, b"dispatch_threshold_4")
        out, out_off, ol, st, det = _batch(block, ctx, [c for c, _ in picks], [len(p) for _, p in picks])
    finally:
        lib.lz4flex_ctx_destroy(ctx)
    for i, (c, p) in enumerate(picks):
        o = int(out_off[i])
        assert st[i] == 0 and ol[i] == len(p) and out[o:o + len(p)].tobytes() == p, i
