"""Inputs and known-answer vectors the reference's own tests hold for the block/frame path.
Vectors are data, cited by reference file:line (paths relative to the lz4_flex tree)."""

# ---- decoder KATs: src/block/decompress.rs:534-622 (same list in decompress_safe.rs:396-484) ----
# (input bytes, output capacity, dict or None, expected) ; expected = ("ok", bytes) | (ErrorName, (expected, actual) | None)
DECODER_KATS = [
    (bytes([0x30, ord("a"), ord("4"), ord("9")]), 3, None, ("ok", b"a49")),                       # :536 all_literal
    (b"", 255, None, ("ExpectedAnotherByte", None)),                                              # :541-544
    (bytes([0xF0]), 255, None, ("ExpectedAnotherByte", None)),                                    # :545-549 incomplete literal len
    (bytes([0x0F, 0]), 255, None, ("ExpectedAnotherByte", None)),                                 # :550-554 incomplete match offset
    (bytes([0x0F, 1, 0]), 255, None, ("ExpectedAnotherByte", None)),                              # :555-559 incomplete match len
    (bytes([0x40, ord("a"), 1, 0]), 4, None, ("LiteralOutOfBounds", None)),                       # :566-569
    (bytes([0x20, ord("a"), ord("a"), 1, 0]), 1, None, ("OutputTooSmall", (2, 1))),               # :571-577
    (bytes([0x10, ord("a"), 1, 0]), 4, None, ("OutputTooSmall", (5, 4))),                         # :579-585
    (bytes([0x0E, 255] + [0] * 18), 256, None, ("OffsetOutOfBounds", None)),                      # :588-594 hot loop
    (bytes([0x0E, 255, 0, 0x70, 0, 0, 0, 0, 0, 0, 0]), 256, bytes(250), ("OffsetOutOfBounds", None)),   # :596-603 dict
    (bytes([0x0F, 1, 0, 1, 0x70, 0, 0, 0, 0, 0, 0, 0]), 256, None, ("OffsetOutOfBounds", None)),  # :605-608 overlapping
    (bytes([0x40, 0, 0, 0, 0, 255, 0, 0x70, 0, 0, 0, 0, 0, 0, 0]), 256, None, ("OffsetOutOfBounds", None)),  # :610-613
    (bytes([0x0E, 0, 0, 0x70, 0, 0, 0, 0, 0, 0, 0]), 256, None, ("OffsetZero", None)),            # :617-621
]

# does_token_fit, src/block/decompress.rs:179-186
TOKEN_FIT = [(15, False), (14, True), (114, True), (0b11110000, False), (0b10110000, True)]

# count_same_bytes exact counts, src/block/compress.rs:807-881: (first, second, expected)
COUNT_SAME = [
    (bytes([1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16] + [0] * 6), None, 16),
]

# ---- round-trip corpus: tests/tests.rs:353-588 ----
ROUNDTRIP_STRINGS = [
    b"AAAAAAAAAAAAAAAAAAAAAAAAaAAAAAAAAAAAAAAAAAAAAAAAA",                                           # :357
    b"AAAAAAAAAAAAAAAAAAAAAAAABBBBBBBBBaAAAAAAAAAAAAAAAAAAAAAAAA",                                  # :358
    b"AAAAAAAAAAAAAAAAAAAAAAAABBBBBBBBBaAAAAAAAAAAAAAAAAAAAAAAAABBBBBBBBBa",                        # :362
    b"AAAAAAAAAAAZZZZZZZZAAAAAAAA",                                                                 # :366
    b"to live or not to live", b"Love is a wonderful terrible thing",                               # :376,380
    b"There is nothing either good or bad, but thinking makes it so.", b"I burn, I pine, I perish.",  # :384,388
    b"Save water, it doesn't grow on trees.", b"The panda bear has an amazing black-and-white fur.",
    b"The average panda eats as much as 9 to 14 kg of bamboo shoots a day.",
    b"You are 60% water. Save 60% of yourself!", b"To cute to die! Save the red panda!",            # :393-397
    b"as6yhol.;jrew5tyuikbfewedfyjltre22459ba", b"jhflkdjshaf9p8u89ybkvjsdbfkhvg4ut08yfrr",         # :402-403
    b"ahhd", b"ahd", b"x-29", b"x", b"k", b".", b"ajsdh", b"aaaaaa",                                # :407-414
    b"aaaaaabcbcbcbc", b"", bytes(13),                                                              # :419,424,429
]

BUG_FUZZ = [   # tests/tests.rs:433-495
    bytes([8, 6] + [0] * 288 + [46, 0, 0, 8, 0, 138]),
    bytes([122] + [0] * 15 + [8] + [0] * 81 + [65, 0, 0, 128, 10, 1, 10, 1, 0, 122]),
    bytes([36, 16, 0, 0, 79, 177, 176, 176, 171, 1, 0, 255, 207, 79, 79, 79, 79, 79, 1, 1, 49, 0, 16,
           0, 79, 79, 79, 79, 79, 1, 0, 255, 36, 79, 79, 79, 79, 79, 1, 0, 255, 207, 79, 79, 79, 79,
           79, 1, 0, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 8, 207, 1, 207, 207, 79, 199,
           79, 79, 40, 79, 1, 1, 1, 1, 1, 1, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15,
           15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 79, 15, 15, 14, 15, 15, 15, 15, 15, 15,
           15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 61, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 0,
           48, 45, 0, 1, 0, 0, 1, 0]),
    bytes([147]),
    bytes([255, 255, 255, 255, 253, 235, 156, 140, 8, 0, 140, 45, 169, 0, 27, 128, 48, 0, 140, 0, 0,
           255, 255, 255, 253, 235, 156, 140, 8, 61, 255, 255, 255, 255, 65, 239, 254]),
    bytes([181, 181, 181, 181, 181, 147, 147, 147, 0, 0, 255, 218, 44, 0, 177, 44, 0, 233, 177, 74,
           85, 47, 95, 146, 189, 177, 1, 0, 255, 2, 109, 180, 255, 255, 0, 0, 0, 181, 181, 181, 147,
           147, 147, 0, 0, 255, 218, 146, 146, 181, 0, 0, 181]),
]

COMPRESSION_WORKS = (   # tests/tests.rs:536-544
    b"An iterator that knows its exact length.\n"
    b"        Many Iterators don't know how many times they will iterate, but some do. If an iterator knows how many "
    b"times it can iterate, providing access to that information can be useful. For example, if you want to iterate "
    b"backwards, a good start is to know where the end is.\n"
    b"        When implementing an ExactSizeIterator, you must also implement Iterator. When doing so, the "
    b"implementation of size_hint must return the exact size of the iterator.\n"
    b"        The len method has a default implementation, so you usually shouldn't implement it. However, you may be "
    b"able to provide a more performant implementation than the default, so overriding it in this case makes sense."
)

# corrupt inputs that must not crash (result ignored): tests/tests.rs:321-351 (size-prepended) and :497-526
NO_PANIC_SIZE_PREPENDED = [
    bytes([122, 1, 0, 1, 0, 10, 1, 0]),
    bytes([44, 251, 49, 0, 0, 0, 8, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 16, 0, 0, 0, 0, 0, 0, 0, 0]),
    bytes([7, 0, 0, 0, 0, 0, 0, 11, 0, 0, 7, 16, 0, 0, 0, 0, 0, 0, 0, 0, 0, 4, 1, 0, 0]),
    bytes([0, 61, 0, 0, 0, 7, 0]),
    bytes([8, 0, 0, 0, 4, 0, 0, 0]),
    bytes([39, 0, 0, 0, 0, 0, 0, 237, 0, 0, 0, 0, 0, 0, 16, 0, 0, 4, 0, 0, 0, 39, 32, 0, 2, 0, 162, 5,
           36, 0, 0, 0, 0, 7, 0]),                                                                  # bug_fuzz_7
    bytes([0] * 20 + [10, 0, 0, 10]),                                                               # bug_fuzz_8
]

# frame header goldens: fuzz/fuzz_targets/fuzz_decomp_corrupt_frame.rs:26-27
FRAME_HEADER_GOLDENS = [
    (dict(block_mode=0, block_size=4), bytes([0x04, 0x22, 0x4D, 0x18, 0x60, 0x40, 0x82])),
    (dict(block_mode=1, block_size=4), bytes([0x04, 0x22, 0x4D, 0x18, 0x40, 0x40, 0xC0])),
]

# ratio ceilings: tests/tests.rs:159-192
RATIO_BLOCK = {"compression_34k": 0.585, "compression_65k": 0.574, "compression_66k_JSON": 0.229}
RATIO_FRAME = {"compression_34k": 0.585, "compression_65k": 0.574, "compression_66k_JSON": 0.235}

FIXTURES = ["compression_1k", "compression_34k", "compression_65k", "compression_66k_JSON"]


def roundtrip_inputs():
    """every input tests/tests.rs pushes through test_roundtrip that exists in this repository"""
    out = list(ROUNDTRIP_STRINGS) + list(BUG_FUZZ) + [COMPRESSION_WORKS, bytes(30000)]   # :529-532 so_many_zeros
    return out


def lcg_bytes(n, seed, alphabet=256, run=1):
    """deterministic pseudo-random bytes with tunable entropy (alphabet size, run length)"""
    out = bytearray(n)
    x = seed & 0xFFFFFFFFFFFFFFFF
    i = 0
    while i < n:
        x = (x * 6364136223846793005 + 1442695040888963407) & 0xFFFFFFFFFFFFFFFF
        b = (x >> 33) % alphabet
        r = 1 + ((x >> 20) % run) if run > 1 else 1
        for _ in range(r):
            if i < n:
                out[i] = b
                i += 1
    return bytes(out)


def adversarial_blocks():
    """[(compressed bytes, sink capacity)]: every prefix of a small block, single-byte corruptions of a 34 KB one, short and
    long sinks, runs, short-period data (offsets 2..33: the copiers' single-lane pieces and periodic path), random data (long
    literal runs), blocks from both encoders, truncated and with a sink 7 bytes short, tiny inputs.  Shared by
    tests/test_gpu_block.py (every decoder kernel against the oracle) and tools/dec_variants.py (variant builds)."""
    import random
    import oracle_api as O
    rnd = random.Random(5)
    cases = []
    small = O.fixture_plain("compression_1k")
    blk = O.compress(small)
    for k in range(len(blk) + 1):
        cases.append((blk[:k], len(small)))
    mid = O.fixture_plain("compression_34k")
    cm = O.compress(mid)
    for k in range(0, len(cm), 11):
        bad = bytearray(cm)
        bad[k] ^= 0x5A
        cases.append((bytes(bad), len(mid)))
    for cap in (0, 1, 100, len(mid) - 1, len(mid) + 1000):
        cases.append((cm, cap))
    gen = [bytes(30000), b"ab" * 9000, b"abc" * 7000, bytes(range(5)) * 3000, bytes(range(7)) * 2000, bytes(range(13)) * 3000,
           bytes(range(17)) * 2000, bytes(range(33)) * 900, bytes(rnd.getrandbits(8) for _ in range(20000)),
           bytes(rnd.choice(b"ab") for _ in range(40000)), O.fixture_plain("compression_65k")[:65536],
           O.fixture_plain("compression_66k_JSON")[:65536], b"x" * 300 + bytes(rnd.getrandbits(8) for _ in range(300)) + b"y" * 70000]
    for d in gen:
        for enc in (O.compress, O.c_compress):
            c = enc(d)
            cases.append((c, len(d)))
            cases.append((c[:-1], len(d)))
            cases.append((c, len(d) - 7))
    for n in range(0, 40):
        d = bytes(rnd.getrandbits(2) for _ in range(n))
        cases.append((O.compress(d), n))
    return cases


def synthetic_block(rnd, target):
    """(compressed, plain): an LZ4 block written sequence by sequence, not by an encoder -- matches aim at what a greedy encoder
    rarely produces in bulk: the interior of earlier matches (copies of copies of copies, the chains the workgroup decoder
    relinks), the bytes straddling two sequences, their own output (offsets 1..8), the block's first bytes, literal runs and
    matches of every length class (one and several length bytes).  The plain text is produced with the format's byte-wise
    semantics right here; the CPU suite checks it against the oracle."""
    out, comp, seqs = bytearray(), bytearray(), []

    def put_len(v):
        while v >= 255:
            comp.append(255)
            v -= 255
        comp.append(v)

    while len(out) < target:
        r = rnd.random()
        lit = 0 if r < 0.3 else rnd.randint(1, 8) if r < 0.8 else rnd.randint(9, 40) if r < 0.95 else rnd.randint(100, 600)
        if not out and lit == 0:
            lit = 1
        r = rnd.random()
        ml = rnd.randint(4, 8) if r < 0.4 else rnd.randint(9, 30) if r < 0.8 else rnd.randint(31, 100) if r < 0.95 else rnd.randint(300, 3000)
        pos = len(out) + lit
        r = rnd.random()
        recent = seqs[-40:]
        if r < 0.15:
            off = rnd.randint(1, 8)                                   # its own output: periodic
        elif r < 0.65 and recent:
            ms_j, ml_j = rnd.choice(recent)                           # inside an earlier match: a copy of a copy
            if rnd.random() < 0.7 and ml_j > ml:
                start = ms_j + rnd.randint(0, ml_j - ml)              # entirely inside it
            else:
                start = ms_j + rnd.randint(0, ml_j - 1)
            off = pos - start
        elif r < 0.8 and recent:
            ms_j, ml_j = rnd.choice(recent)                           # straddling the end of an earlier match
            off = pos - (ms_j + ml_j - rnd.randint(1, 3))
        elif r < 0.85:
            off = pos                                                 # the block's first byte
        else:
            off = rnd.randint(1, 65535)
        off = max(1, min(off, pos, 65535))
        tok = (min(lit, 15) << 4) | min(ml - 4, 15)
        comp.append(tok)
        if lit >= 15:
            put_len(lit - 15)
        lits = bytes(rnd.getrandbits(8) for _ in range(lit))
        comp += lits
        out += lits
        comp += bytes((off & 0xFF, off >> 8))
        if ml - 4 >= 15:
            put_len(ml - 4 - 15)
        start = len(out) - off
        if off >= ml:
            out += out[start:start + ml]
        else:
            for k in range(ml):
                out.append(out[start + k])
        seqs.append((pos, ml))
    tail = bytes(rnd.getrandbits(8) for _ in range(rnd.randint(5, 30)))   # the last sequence: literals only
    comp.append(min(len(tail), 15) << 4)
    if len(tail) >= 15:
        put_len(len(tail) - 15)
    comp += tail
    out += tail
    return bytes(comp), bytes(out)


def synthetic_blocks(seed=77, sizes=(150000,) * 24 + (5000,) * 24 + (1 << 20,) * 2):
    import random
    rnd = random.Random(seed)
    return [synthetic_block(rnd, n) for n in sizes]
