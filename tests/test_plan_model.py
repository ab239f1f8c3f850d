"""CPU: the host model of the plan / replay decoder (tests/sim/plan_model.cpp; kernels: lz4_flex_amd/csrc/lz4_decompress_plan.hip,
lz4_decompress_replay.hip) against the oracle (lz4_flex's decoder restated).  For every input the model either returns exactly the
oracle's bytes or calls the block irregular; it may do the latter only when the oracle fails.  The replay model moves 16 bytes per lane whatever a piece's length and fails a run when a
lane reads outside the compressed block or outside the final output, or writes outside the sink."""
import random

import corpus
import oracle_api as O
import plan_model as M
import wave_model as W


def check(comp, cap):
    exp = O.decompress(comp, cap)
    got = M.decode(comp, cap)
    if exp[0] == "ok":
        assert got is not None, "a valid block was handed to the reference-order kernel"
        assert got == exp[1]
        return "ok"
    assert got is None, "the model decoded a block the reference rejects (%s)" % exp[0]
    return "irregular"


def test_model_adversarial_blocks():
    kinds = {}
    for comp, cap in corpus.adversarial_blocks():
        k = check(comp, cap)
        kinds[k] = kinds.get(k, 0) + 1
    assert kinds.get("ok", 0) > 100 and kinds.get("irregular", 0) > 100, kinds


def test_model_fixtures_three_encoders_exact_and_short_sinks():
    for name in corpus.FIXTURES:
        data = O.fixture_plain(name)
        for comp in (O.compress(data), W.compress(data), O.c_compress(data)):
            assert check(comp, len(data)) == "ok"
            assert check(comp, len(data) + 77) == "ok"
            assert check(comp, len(data) - 1) == "irregular"


def test_model_json_tiles_every_ring_phase():
    """64 KiB JSON tiles (the benchmark's blocks) at phases that move every piece across the ring's seams"""
    from lz4_flex_amd import workloads
    plain = O.fixture_plain("compression_66k_JSON")
    for phase in (0, 1, 17, 2047, 30001):
        data = bytes(workloads.json_tiles(plain, 65536, phase=phase).numpy())
        for comp in (O.compress(data), W.compress(data)):
            assert check(comp, 65536) == "ok"


def test_model_runs_periods_and_random():
    rnd = random.Random(5)
    for n in (16, 17, 63, 64, 65, 100, 1000, 2047, 2048, 2049, 4096, 70000):
        for data in (bytes(n), bytes([7]) * n, (b"ab" * n)[:n], (b"abc" * n)[:n], (b"0123456789abcde" * n)[:n],
                     (b"0123456789abcdefg" * n)[:n], bytes(rnd.randrange(256) for _ in range(n)),
                     corpus.lcg_bytes(n, n, alphabet=2), corpus.lcg_bytes(n, n + 1, alphabet=4, run=5)):
            for comp in (O.compress(data), W.compress(data), O.c_compress(data)):
                assert check(comp, len(data)) == "ok"


def test_model_synthetic_blocks():
    for comp, plain in corpus.synthetic_blocks(sizes=(150000,) * 4 + (5000,) * 12):
        assert check(comp, len(plain)) == "ok"
