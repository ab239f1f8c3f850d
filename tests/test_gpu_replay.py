"""GPU (-m gpu): the REPLAY kernel (lz4_flex_amd/csrc/lz4_decompress_replay.hip) on copy plans compiled by the host model
(tests/sim/plan_model.cpp, which feeds the reference's parse to the emitter the kernels share: lz4_plan_common.h).  The plan
kernel that will produce these plans on the device is not in the tree yet: until then the replay kernel is not on any product
path (no decoder dispatches to it) -- this test pins that what it does with a plan is what the model's lane-by-lane replay does,
i.e. the oracle's bytes, for every valid block of the adversarial batch, the fixtures under three encoders, runs / short
periods, the synthetic copy-chain blocks and 64 KiB JSON tiles, with nothing written outside a block's output."""
import ctypes as C
import random

import numpy as np
import pytest

import corpus
import oracle_api as O
import plan_model as M
import wave_model as W

pytestmark = pytest.mark.gpu


def _replay(blocks):
    """blocks: [(compressed bytes, expected plain bytes)] -> list of decoded bytes, through host plans + the GPU replay kernel"""
    import torch
    from lz4_flex_amd import _lib
    lib = _lib.load()
    lib.lz4flex_debug_replay.restype = C.c_int
    lib.lz4flex_debug_replay.argtypes = [C.c_void_p] * 4 + [C.c_uint, C.c_void_p]
    dev = torch.device("cuda", 0)
    n = len(blocks)
    in_len = np.array([len(c) for c, _ in blocks], dtype=np.uint32)
    in_off = np.concatenate([[0], np.cumsum(in_len[:-1], dtype=np.uint64)]).astype(np.uint64)
    h_in = np.frombuffer(b"".join(c for c, _ in blocks), dtype=np.uint8).copy()          # NO slack behind the last block
    caps = np.array([len(p) for _, p in blocks], dtype=np.uint32)
    guard = 64
    out_off = np.concatenate([[0], np.cumsum(caps[:-1].astype(np.uint64) + guard)]).astype(np.uint64)
    out_bytes = int(out_off[-1]) + int(caps[-1]) + guard
    max_words = int(in_len.sum()) * 8 + int(caps.sum()) // 8 + n * 1024
    words = np.zeros(max_words, dtype=np.uint32)
    plans = np.zeros(n * 32, dtype=np.uint8)
    olen = np.zeros(n, dtype=np.uint32)
    vp = lambda a: C.c_void_p(a.ctypes.data)
    steps = C.c_uint64(0)
    used = M.lib().plan_compile_batch(vp(h_in), vp(in_off), vp(in_len), vp(out_off), vp(caps), n, vp(plans), vp(words), max_words, vp(olen),
                                      C.byref(steps))
    assert used > 0
    assert (olen == caps).all(), "the model calls a valid block irregular"
    d_in = torch.from_numpy(h_in).to(dev)
    d_words = torch.from_numpy(words[:used].copy()).to(dev)
    d_plans = torch.from_numpy(plans).to(dev)
    d_out = torch.full((out_bytes,), 0xA5, dtype=torch.uint8, device=dev)
    p = lambda t: C.c_void_p(t.data_ptr())
    rc = lib.lz4flex_debug_replay(p(d_in), p(d_out), p(d_plans), p(d_words), n, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert rc == 0
    h_out = d_out.cpu().numpy()
    res = []
    for i in range(n):
        o, k = int(out_off[i]), int(caps[i])
        res.append(h_out[o:o + k].tobytes())
        assert h_out[o + k:o + k + guard].tobytes() == b"\xA5" * guard, "block %d: bytes written behind its output" % i
    return res


def _check(blocks):
    got = _replay(blocks)
    for i, ((c, p), g) in enumerate(zip(blocks, got)):
        assert g == p, "block %d (%d compressed, %d plain bytes)" % (i, len(c), len(p))


def test_replay_adversarial_valid_blocks():
    blocks = []
    for comp, cap in corpus.adversarial_blocks():
        r = O.decompress(comp, cap)
        if r[0] == "ok" and len(r[1]) > 0:
            blocks.append((comp, r[1]))
    assert len(blocks) > 100
    _check(blocks)


def test_replay_fixtures_three_encoders_and_tiles():
    from lz4_flex_amd import workloads
    blocks = []
    for name in corpus.FIXTURES:
        data = O.fixture_plain(name)
        for comp in (O.compress(data), W.compress(data), O.c_compress(data)):
            blocks.append((comp, data))
    plain = O.fixture_plain("compression_66k_JSON")
    for phase in (0, 1, 17, 2047, 30001):
        data = bytes(workloads.json_tiles(plain, 65536, phase=phase).numpy())
        blocks.append((W.compress(data), data))
        blocks.append((O.compress(data), data))
    _check(blocks)


def test_replay_runs_periods_random_and_copy_chains():
    rnd = random.Random(5)
    blocks = []
    for n in (1, 5, 16, 17, 63, 64, 65, 100, 1000, 2047, 2048, 2049, 4096, 70000):
        for data in (bytes(n), bytes([7]) * n, (b"ab" * n)[:n], (b"abc" * n)[:n], (b"0123456789abcde" * n)[:n],
                     (b"0123456789abcdefg" * n)[:n], bytes(rnd.randrange(256) for _ in range(n)),
                     corpus.lcg_bytes(n, n, alphabet=2), corpus.lcg_bytes(n, n + 1, alphabet=4, run=5)):
            for comp in (O.compress(data), W.compress(data)):
                blocks.append((comp, data))
    for comp, plain in corpus.synthetic_blocks(sizes=(150000,) * 3 + (5000,) * 12):
        blocks.append((comp, plain))
    rnd.shuffle(blocks)
    _check(blocks)
