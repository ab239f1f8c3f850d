"""The fused decoder (lz4_flex_amd/csrc/lz4_decompress_fused.hip) on the host: its parser and its emitter are the kernel's code
(lz4_split_parser.h, lz4_fused_common.h compiled with -DLZ4FLEX_HOST_SIM), its quad a lane-exact model (tests/sim/fused_model.cpp: 16-byte
moves, reads at once / writes in lane order, memory sources read LOOKAHEAD steps early, a 1 KiB ring, special steps at the end of a turn)
with guards for every read outside the block or of unwritten output.  Output bytes, length, error variant and
OutputTooSmall{expected, actual} == the oracle on the reference's KATs (src/block/decompress.rs:534-622), the fixtures, every prefix
and corruptions of real blocks, runs, short periods, long literal runs, all source alignments, with the three stages stepped in
several pseudo-random orders (every queue is seen full and empty).  CPU only."""
import pytest

import corpus
import fused_model as F
import oracle_api as O


@pytest.fixture(scope="module")
def model():
    if F.clangxx() is None:
        pytest.skip("no clang++ (ext_vector_type) to build the host model")
    return F.lib()


def _decode(data, cap, misalign=0, seed=0):
    st, out, det, _ = F.decode(data, cap, misalign, seed)
    assert st >= 0, "the model's guard %d fired (len %d cap %d misalign %d seed %d)" % (st, len(data), cap, misalign, seed)
    if st:
        return O.ERR_NAMES[st], (int(det[0]), int(det[1]))
    return "ok", out


def _same_as_oracle(data, cap, misalign=0, seed=0):
    want = O.decompress(data, cap)
    got = _decode(data, cap, misalign, seed)
    if want[0] == "ok":
        assert got == want
    else:
        assert got[0] == want[0]
        if want[0] == "OutputTooSmall":
            assert got[1] == want[1]


def test_kats(model):
    for data, cap, d, (exp, payload) in corpus.DECODER_KATS:
        if d is not None:
            continue
        for seed in (0, 3):
            st, got = _decode(data, cap, 0, seed)
            assert st == exp, (data, st, exp)
            if exp == "ok" or payload is not None:
                assert got == payload


@pytest.mark.parametrize("stem", corpus.FIXTURES)
def test_fixtures_alignments_and_schedules(model, stem):
    m = O.manifest()[stem]
    blk = O.golden_block(stem)
    for mis in list(range(4)) + [16, 17, 18, 19]:        # bits 0-1: source misalignment; bit 4: the parser's wave-level tests fire at random
        for seed in (0, 1, 2, 5, 11):
            _same_as_oracle(blk, m["plain_len"], mis, seed)
    _same_as_oracle(blk, m["plain_len"] + 1000, 1, 4)
    _same_as_oracle(blk, m["plain_len"] - 1, 2, 6)
    _same_as_oracle(O.c_compress(O.fixture_plain(stem)), m["plain_len"], 3, 9)


def test_roundtrip_corpus_and_entropies(model):
    inputs = corpus.roundtrip_inputs() + [corpus.lcg_bytes(70000, 5, 4, 9), corpus.lcg_bytes(70000, 6, 256, 1),
                                          corpus.lcg_bytes(70000, 7, 3, 40), bytes(70000), corpus.lcg_bytes(3000, 8, 2, 300)]
    inputs += [corpus.lcg_bytes(n, 11 + n, 5, 3) for n in list(range(0, 130)) + [255, 256, 257, 271, 272, 300, 1000]]
    for i, p in enumerate(inputs):
        if len(p) > F.MAX_FIELD:
            continue
        for comp in (O.compress(p), O.c_compress(p) if p else None):
            if comp is None:
                continue
            _same_as_oracle(comp, len(p), len(p) % 4 + 16 * (i & 1), i % 7)
            if len(p):
                _same_as_oracle(comp, len(p) - 1, 0, (i + 3) % 7)


def test_every_prefix_and_corruptions(model):
    blk = O.golden_block("compression_1k")
    n = O.manifest()["compression_1k"]["plain_len"]
    for cut in range(len(blk)):
        _same_as_oracle(blk[:cut], n, cut % 4, cut % 5)
        _same_as_oracle(blk[:cut], n, 16 + cut % 4, 1 + cut % 3)
    big = O.golden_block("compression_66k_JSON")
    nb = O.manifest()["compression_66k_JSON"]["plain_len"]
    for pos in list(range(0, 300)) + list(range(300, len(big), 97)) + list(range(len(big) - 60, len(big))):
        for val in (0x00, 0xFF, big[pos] ^ 0x10):
            bad = bytearray(big)
            bad[pos] = val
            _same_as_oracle(bytes(bad), nb, 16 * (pos & 1), pos % 6)
    for junk in corpus.NO_PANIC_SIZE_PREPENDED + corpus.BUG_FUZZ:
        _same_as_oracle(junk, 4096, 0, 2)
        _same_as_oracle(junk[4:], 4096, 1, 0)


def test_runs_periods_and_long_literals(model):
    """what the emitter and the special steps exist for: every short period (doubling pieces), matches of one distance longer than the ring,
    literal runs around the ring size and longer (special steps through the ring), incompressible blocks (ONE run), runs between matches
    that reach into them, far and near sources around NEAR_MAX"""
    cases = [bytes([(i % p) * 7 & 255 for i in range(9000)]) for p in list(range(1, 40)) + [63, 64, 65, 127, 128, 129, 959, 960, 961, 1023, 1024, 1025, 2000]]
    cases += [bytes(300000), b"ab" * 100000, corpus.lcg_bytes(200000, 22, 256, 1), corpus.lcg_bytes(500, 1, 256, 1) * 300]
    noise = corpus.lcg_bytes(70000, 24, 256, 1)
    for run in (270, 271, 800, 830, 832, 833, 900, 1023, 1024, 1025, 1100, 2047, 2048, 5000, 65000):
        cases.append(b"header header header " + noise[:run] + noise[run - 200:run - 100] + noise[:50] + b"x" * 40 + noise[run - 16:run] + b"tail tail tail tail")
    for i, p in enumerate(cases):
        for comp in (O.compress(p), O.c_compress(p)):
            _same_as_oracle(comp, len(p), i % 4, i % 5)
            _same_as_oracle(comp, len(p) + 17, 16 + (i + 1) % 4, (i + 2) % 5)
            _same_as_oracle(comp, len(p) - 1, 3, 1)
            _same_as_oracle(comp[:len(comp) - 1], len(p), 2, 3)


def test_seeded_mutations(model):
    """seeded multi-byte mutations of a real block: same error variant (or same bytes) as the oracle every time"""
    blk = bytearray(O.golden_block("compression_34k"))
    n = O.manifest()["compression_34k"]["plain_len"]
    x = 0x9E3779B97F4A7C15
    for it in range(600):
        bad = bytearray(blk)
        for _ in range(1 + it % 3):
            x = (x * 6364136223846793005 + 1442695040888963407) & 0xFFFFFFFFFFFFFFFF
            pos = (x >> 20) % len(bad)
            bad[pos] = (x >> 50) & 0xFF
        cut = len(bad) if it % 5 else (x >> 7) % len(bad)
        _same_as_oracle(bytes(bad[:cut]), n if it % 7 else n // 2, it % 4 + 16 * (it & 1), it % 9)


def test_oversized_blocks_are_left_alone(model):
    """records name positions below 512 KiB: the kernel marks larger blocks for the reference-order kernel (the model says -100)"""
    st, _, _, _ = F.decode(b"\x00", 1 << 19)
    assert st == -100


def test_hypothesis_random_inputs(model):
    hyp = pytest.importorskip("hypothesis")
    st = hyp.strategies

    @hyp.settings(max_examples=120, deadline=None, database=None)
    @hyp.given(st.binary(min_size=0, max_size=3000), st.integers(0, 31), st.integers(0, 2 ** 32 - 1))
    def run(data, mis, seed):
        low = bytes(b & 3 for b in data) * 3   # long matches, 255-chains, periodic offsets
        for p in (data, low):
            for comp in (O.compress(p), O.c_compress(p) if p else b"\x00"):
                _same_as_oracle(comp, len(p), mis & 19, seed % 13)
                if comp:
                    bad = bytearray(comp)
                    bad[seed % len(bad)] ^= 1 << (seed >> 8) % 8
                    _same_as_oracle(bytes(bad), len(p), mis & 19, seed % 11)
                    _same_as_oracle(comp[:seed % (len(comp) + 1)], len(p) + (seed >> 12) % 5, mis & 19, seed % 7)

    run()
