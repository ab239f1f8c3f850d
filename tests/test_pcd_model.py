"""CPU: the host model of the parallel-chain decoder (tests/sim/pcd_model.cpp; kernel: lz4_flex_amd/csrc/lz4_decompress_pcd.hip)
against the oracle (lz4_flex's decoder restated).  For every input: the model either returns exactly the oracle's bytes, or calls
the block irregular -- and it may only do that when the oracle fails (an error or a sink that is too small), or for the one
legitimate reason a valid block is handed to the reference-order kernel: a chain that does not settle within MAX_ITERS walks
(never seen on real data; counted here).  Tiny geometries put tile / part / batch / window boundaries everywhere."""
import random

import pytest

import corpus
import oracle_api as O
import pcd_model as M
import wave_model as W


def geometries():
    d = M.defaults()
    tiny = M.Params(ct=128, p=32, batch=4, hist=16, wnew=48, max_iters=200)
    small = M.Params(ct=1024, p=64, batch=16, hist=64, wnew=256, max_iters=200)
    mid = M.Params(ct=4096, p=128, batch=64, hist=1024, wnew=4096, max_iters=200)
    return [("default", d), ("tiny", tiny), ("small", small), ("mid", mid)]


def check(comp, cap, params, seed=1, must_be_regular=False):
    exp = O.decompress(comp, cap)
    got, st = M.decode(comp, cap, params, seed)
    if exp[0] == "ok":
        if got is None:
            assert not must_be_regular, "a valid block was handed to the reference-order kernel"
            return "irregular-valid", st
        assert got == exp[1]
        return "ok", st
    assert got is None, "the model decoded a block the reference rejects (%s)" % exp[0]
    return "irregular", st


@pytest.mark.parametrize("gname,params", geometries())
def test_model_adversarial_blocks(gname, params):
    n_irreg_valid = 0
    for i, (comp, cap) in enumerate(corpus.adversarial_blocks()):
        r, _ = check(comp, cap, params, seed=i)
        n_irreg_valid += r == "irregular-valid"
    assert n_irreg_valid == 0


@pytest.mark.parametrize("gname,params", geometries())
def test_model_fixtures_and_both_encoders(gname, params):
    rnd = random.Random(7)
    datas = [O.fixture_plain(s) for s in corpus.FIXTURES]
    j = O.fixture_plain("compression_66k_JSON")
    datas += [(j * 5)[:300000], bytes(100000), bytes(rnd.getrandbits(8) for _ in range(50000)),
              b"".join(bytes([rnd.getrandbits(8)]) * rnd.randint(1, 700) for _ in range(300)),
              bytes(rnd.getrandbits(8) for _ in range(3000)) + bytes(70000) + bytes(rnd.getrandbits(8) for _ in range(40000)) + b"ab" * 30000]
    datas += list(corpus.roundtrip_inputs())
    for d in datas:
        for enc in (O.compress, O.c_compress, W.compress):
            c = enc(d)
            for seed in (1, 2):
                check(c, len(d), params, seed, must_be_regular=True)
            check(c, len(d) + 1000, params, 3, must_be_regular=True)      # a larger sink is fine (decompress_into)
            if len(d) > 8:
                check(c, len(d) - 1, params, 4)                          # OutputTooSmall -> irregular
                check(c[:len(c) // 2], len(d), params, 5)                # truncated


def test_model_statistics_on_the_benchmark_data():
    """what the kernel's design counts on: a tile's chain settles after two walks, few parts need a third"""
    import numpy as np
    from lz4_flex_amd import workloads
    j = O.fixture_plain("compression_66k_JSON")
    log = bytes(workloads.log_stream(0, 4 << 20, device="cpu").numpy())
    text = (O.fixture_plain("compression_65k") * 20)[:1 << 20]
    for name, d in (("log 4 MiB", log), ("json 1 MiB", (j * 20)[:1 << 20]), ("text 1 MiB", text)):
        for enc in (W.compress, O.compress):
            c = enc(d)
            got, st = M.decode(c, len(d))
            assert got == d
            assert st.iters <= 8 * st.tiles, (name, st.iters, st.tiles)              # measured: 2.1 - 4.6 rounds per tile ...
            assert st.part_walks <= 2.2 * (len(c) / 128 + st.tiles)                 # ... but only 1.8 walks per part: later rounds re-walk a few parts
            assert st.giants <= 1
