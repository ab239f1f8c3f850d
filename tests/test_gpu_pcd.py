"""GPU (-m gpu): the parallel-chain decoder (lz4_flex_amd/csrc/lz4_decompress_pcd.hip: one workgroup per block, token chain and
copies parallel INSIDE the block) on the shapes it exists for -- few, large blocks -- and against its host model.

Checker: the oracle (lz4_flex's decoder restated).  Both geometries of the kernel run everything: the production one (32 KiB
tiles, 2 048 sequences per batch, 26 + 48 KiB window) and the test one (2 KiB tiles, 256 sequences, 0.5 + 1 KiB window), which
puts tile / part / batch / window boundaries and the giant-sequence path inside ordinary inputs.  The generic decoder matrix
of test_gpu_block.py (KATs, the 2 490-block adversarial batch, mixed batches) runs both geometries too (DECODERS -7, -8)."""
import ctypes as C
import random

import numpy as np
import pytest

import corpus
import oracle_api as O
import pcd_model as M
import wave_model as W

pytestmark = pytest.mark.gpu
REDO = 0x7F000001


@pytest.fixture(scope="module")
def env():
    from lz4_flex_amd import _lib, block
    lib = _lib.load()
    assert lib.lz4flex_device_count() >= 1
    return lib, block


def _ctx(lib, variant, second_pass=1, pair=None):
    ctx = C.c_void_p()
    assert lib.lz4flex_ctx_create(C.byref(ctx), -1) == 0
    assert lib.lz4flex_set_tuning(ctx, b"decompress_variant", variant) == 0
    assert lib.lz4flex_set_tuning(ctx, b"decompress_second_pass", second_pass) == 0
    if pair is not None:      # 0: one workgroup per block; 2: a parser and a copier workgroup per block in every batch of <= 128 blocks
        assert lib.lz4flex_set_tuning(ctx, b"decompress_pcd_pair", pair) == 0
    return ctx


def _batch(block, ctx, comps, caps, slack=64):
    inb = np.frombuffer(b"".join(comps) + bytes(64), dtype=np.uint8)
    in_len = [len(c) for c in comps]
    in_off = np.concatenate([[0], np.cumsum(in_len[:-1], dtype=np.uint64)]).astype(np.uint64)
    out_off = np.concatenate([[0], np.cumsum([k + slack for k in caps[:-1]], dtype=np.uint64)]).astype(np.uint64)
    out = np.full(int(out_off[-1]) + caps[-1] + slack, 0xA5, dtype=np.uint8)
    ol, st, det = block.decompress_batch(inb, in_off, in_len, out, out_off, caps, ctx=ctx)
    return out, out_off, ol, st, det


def big_inputs():
    from lz4_flex_amd import workloads
    rnd = random.Random(21)
    j = O.fixture_plain("compression_66k_JSON")
    t = O.fixture_plain("compression_65k")
    log = bytes(workloads.log_stream(0, 4 << 20, device="cpu").numpy())
    return [
        ("log 4 MiB", log),
        ("json 1 MiB", (j * 17)[:1 << 20]),
        ("text 300 KB", (t * 5)[:300000]),
        ("zeros 1 MiB (one match longer than the window)", bytes(1 << 20)),
        ("random 300 KB (one literal run longer than the window)", bytes(rnd.getrandbits(8) for _ in range(300000))),
        ("period 3 / 7 / 20 / 300 runs", b"abc" * 40000 + bytes(range(7)) * 20000 + bytes(rnd.getrandbits(8) for _ in range(20)) * 9000
         + bytes(rnd.getrandbits(8) for _ in range(300)) * 700),
        ("periods around the limits of the giant-match splat (1 504 in the test geometry, 4 096 in production)",
         b"".join(bytes(rnd.getrandbits(8) for _ in range(q)) * (70000 // q + 2) for q in (1, 2, 16, 17, 64, 1503, 1504, 1505, 4095, 4096, 4097))),
        ("mixed: random, zeros, text, random", bytes(rnd.getrandbits(8) for _ in range(50000)) + bytes(200000) + t + bytes(rnd.getrandbits(8) for _ in range(70000)) + j),
        ("two-letter alphabet (chains rarely meet: many rounds per tile)", bytes(rnd.choice(b"ab") for _ in range(120000))),
        ("tiny", b"q"), ("empty", b""), ("13 zeros", bytes(13)),
    ]


@pytest.mark.parametrize("variant,pair", [(7, 0), (7, 2), (8, 0), (8, 2), (10, 0), (11, 0)])
def test_large_blocks_every_encoder(env, variant, pair):
    """blocks of up to 4 MiB from the reference encoder (oracle), C liblz4 and this library's throughput encoder (model): bytes ==
    oracle, nothing behind the sink, also with a sink larger than needed.  With one workgroup per block and with a parser and a
    copier workgroup per block (the token lists travel through the context's workspace)"""
    lib, block = env
    comps, caps, plains = [], [], []
    for name, d in big_inputs():
        for enc in (O.compress, W.compress) + ((O.c_compress,) if d else ()):
            c = enc(d)
            comps += [c, c]
            caps += [len(d), len(d) + 777]
            plains += [d, d]
    ctx = _ctx(lib, variant, pair=pair)
    try:
        out, out_off, ol, st, det = _batch(block, ctx, comps, caps)
    finally:
        lib.lz4flex_ctx_destroy(ctx)
    assert len(comps) <= 128
    for i, d in enumerate(plains):
        o = int(out_off[i])
        assert st[i] == 0 and ol[i] == len(d), (i, int(st[i]), int(ol[i]), len(d))
        assert out[o:o + len(d)].tobytes() == d, "block %d differs" % i
        assert out[o + caps[i]:o + caps[i] + 64].tobytes() == b"\xA5" * 64, "block %d wrote behind its sink" % i
        if caps[i] > len(d):
            assert out[o + len(d):o + caps[i]].tobytes() == b"\xA5" * (caps[i] - len(d)), "block %d wrote behind its end" % i


@pytest.mark.parametrize("variant", [7, 8, 10, 11])
def test_first_pass_marks_exactly_what_the_model_calls_irregular(env, variant):
    """kernel == model: without the second pass, the blocks the kernel leaves marked are exactly the ones tests/sim/pcd_model.cpp
    (same geometry) calls irregular, every other block is decoded (== oracle); with the second pass every result equals the
    oracle's, error variants and OutputTooSmall{expected, actual} included"""
    lib, block = env
    cases = corpus.adversarial_blocks()
    rnd = random.Random(3)
    big = O.compress(bytes(rnd.choice(b"ab") for _ in range(60000)))
    cases += [(big, 60000), (big, 59990), (big[:-5], 60000)]
    prm = {7: M.defaults(), 8: M.Params(ct=2048, p=64, batch=256, hist=512, wnew=1024, max_iters=34),
           10: M.Params(ct=4096, p=64, batch=512, hist=6144, wnew=8192, max_iters=66),        # GeoMid256 (513 ... 1 024 blocks by default)
           11: M.Params(ct=8192, p=64, batch=1024, hist=8192, wnew=16384, max_iters=130)}[variant]   # GeoMid512 (257 ... 512 blocks)
    model = [M.decode(c, k, prm, seed=i)[0] for i, (c, k) in enumerate(cases)]
    want = [O.decompress(c, k) for c, k in cases]
    comps, caps = [c for c, _ in cases], [k for _, k in cases]
    ctx = _ctx(lib, variant, second_pass=0)
    try:
        out, out_off, ol, st, det = _batch(block, ctx, comps, caps)
    finally:
        lib.lz4flex_ctx_destroy(ctx)
    n_marked = 0
    for i, (c, k) in enumerate(cases):
        o = int(out_off[i])
        if len(c) == 0 or model[i] is None:
            assert int(st[i]) == REDO, (i, int(st[i]))
            n_marked += 1
        else:
            assert st[i] == 0 and ol[i] == len(model[i]) and out[o:o + ol[i]].tobytes() == model[i], (i, int(st[i]))
        assert out[o + k:o + k + 64].tobytes() == b"\xA5" * 64, "block %d wrote behind its sink" % i
    assert n_marked > 100
    ctx = _ctx(lib, variant)
    try:
        out, out_off, ol, st, det = _batch(block, ctx, comps, caps)
    finally:
        lib.lz4flex_ctx_destroy(ctx)
    for i, w in enumerate(want):
        o = int(out_off[i])
        if w[0] == "ok":
            assert st[i] == 0 and out[o:o + ol[i]].tobytes() == w[1], i
        else:
            assert O.ERR_NAMES.get(int(st[i])) == w[0], (i, int(st[i]), w[0])
            if w[0] == "OutputTooSmall":
                assert (int(det[i][0]), int(det[i][1])) == tuple(w[1])


@pytest.mark.parametrize("variant", [7, 8])
def test_two_workgroups_per_block_on_the_adversarial_batch(env, variant):
    """the adversarial blocks (every prefix, corruptions, short sinks ...) in launches of <= 128 blocks with a parser and a copier
    workgroup per block: an irregular tile reaches the copier as such, the copier hands the block to the reference-order kernel:
    bytes, error variants and OutputTooSmall{expected, actual} == the oracle; twice through the same context (the workspace is
    reused) -- and a chained batch (a Linked frame written by the oracle) through the default context in that mode"""
    lib, block = env
    cases = corpus.adversarial_blocks()[::3] + corpus.synthetic_blocks(sizes=(150000,) * 3 + (5000,) * 5)
    cases = [(c, k if isinstance(k, int) else len(k)) for c, k in cases]
    ctx = _ctx(lib, variant, pair=2)
    try:
        for rep in range(2):
            for at in range(0, len(cases), 128):
                part = cases[at:at + 128]
                want = [O.decompress(c, k) for c, k in part]
                out, out_off, ol, st, det = _batch(block, ctx, [c for c, _ in part], [k for _, k in part])
                for i, w in enumerate(want):
                    o = int(out_off[i])
                    if w[0] == "ok":
                        assert st[i] == 0 and ol[i] == len(w[1]) and out[o:o + ol[i]].tobytes() == w[1], (at + i, int(st[i]))
                    else:
                        assert O.ERR_NAMES.get(int(st[i])) == w[0], (at + i, int(st[i]), w[0])
                        if w[0] == "OutputTooSmall":
                            assert (int(det[i][0]), int(det[i][1])) == tuple(w[1])
                    assert out[o + part[i][1]:o + part[i][1] + 64].tobytes() == b"\xA5" * 64
    finally:
        lib.lz4flex_ctx_destroy(ctx)
    if variant == 7:
        from lz4_flex_amd import frame as F
        data = (O.fixture_plain("compression_66k_JSON") * 12)[:700000]
        fr = O.frame_compress(data, block_mode=1, block_size=4)[1]
        assert lib.lz4flex_set_tuning(None, b"decompress_pcd_pair", 2) == 0
        try:
            assert F.decompress_frame(fr, len(data))[0] == data
        finally:
            assert lib.lz4flex_set_tuning(None, b"decompress_pcd_pair", 1) == 0


def test_scalar_decompress_into_of_large_blocks_uses_the_workgroup_decoder(env):
    """block::decompress_into of ONE large block (a 1-block batch picks the parallel-chain decoder): 16 MiB of log lines and of
    JSON, from both encoders; the GPU encoder's own output round-trips"""
    lib, block = env
    from lz4_flex_amd import workloads
    assert lib.lz4flex_set_tuning(None, b"decompress_variant", 0) == 0
    log = bytes(workloads.log_stream(128 * 999, 16 << 20, device="cpu").numpy())
    j = (O.fixture_plain("compression_66k_JSON") * 40)[:2 << 20]
    for d in (log, j):
        for c in (O.compress(d), block.compress(d)):
            assert block.decompress(c, len(d)) == d
            with pytest.raises(block.OutputTooSmall) as ei:
                block.decompress(c, len(d) - 1)
            exp = O.decompress(c, len(d) - 1)
            assert exp[0] == "OutputTooSmall" and (ei.value.expected, ei.value.actual) == tuple(exp[1])


def test_device_batch_of_4mib_blocks_round_trip(env):
    """configs[3]'s decode shape: 4 MiB log blocks, device resident, through compress_batch / decompress_batch"""
    import torch
    from lz4_flex_amd import sharded, workloads
    lib, block = env
    bs = 4 << 20
    src = workloads.log_stream(0, 24 * bs + 128 * 97, device="cuda")
    comp, comp_off, comp_len, in_len = sharded.compress_blocks_device(src, bs, np.zeros(25, dtype=np.uint32))
    out, out_len, st = sharded.decompress_blocks_device(comp, comp_off, comp_len, None, bs)
    torch.cuda.synchronize()
    assert int((st != 0).sum().item()) == 0 and torch.equal(out_len.to(torch.int32), in_len.to(torch.int32))
    assert torch.equal(out[:src.numel()], src)
    h = src[:bs].cpu().numpy().tobytes()
    first = comp[:int(comp_len[0].item())].cpu().numpy().tobytes()
    assert O.decompress(first, bs) == ("ok", h)


@pytest.mark.parametrize("variant,bpw", [(4, 8), (4, 16), (4, 32), (4, 64), (13, 0), (7, 0)])
def test_every_decoder_geometry_on_2304_benchmark_blocks(env, variant, bpw):
    """2 304 JSON tiles (configs[1]'s data) written by the throughput encoder, decoded on the device by every kernel and every
    blocks-per-workgroup geometry of the split decoder.  Regression: with 8 / 16 blocks per workgroup the split parser's spare
    lanes wrote their empty chunks into block 0's ring; block 2 096 of this batch then decoded 3 bytes short with status 0."""
    import torch
    from lz4_flex_amd import _lib as L, workloads
    lib, block = env
    n, B, stride = 2304, 65536, 72128
    dev = torch.device("cuda", 0)
    src = workloads.json_tiles(O.fixture_plain("compression_66k_JSON"), n * B, device=dev)
    comp = torch.empty(n * stride, dtype=torch.uint8, device=dev)
    back = torch.zeros(n * B, dtype=torch.uint8, device=dev)
    ar = torch.arange(n, dtype=torch.int64, device=dev)
    in_off, comp_off = ar * B, ar * stride
    in_len = torch.full((n,), B, dtype=torch.int32, device=dev)
    cap = torch.full((n,), stride, dtype=torch.int32, device=dev)
    clen = torch.zeros(n, dtype=torch.int32, device=dev)
    st = torch.full((n,), -1, dtype=torch.int32, device=dev)
    blen = torch.zeros(n, dtype=torch.int32, device=dev)
    bst = torch.full((n,), -1, dtype=torch.int32, device=dev)
    ctx = _ctx(lib, variant)
    p = lambda t: C.c_void_p(t.data_ptr())
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    try:
        assert lib.lz4flex_set_tuning(ctx, b"compress_mode", 0) == 0
        if bpw:
            assert lib.lz4flex_set_tuning(ctx, b"decompress_blocks_per_wg", bpw) == 0
        assert lib.lz4flex_compress_batch(ctx, p(src), p(in_off), p(in_len), None, n, p(comp), p(comp_off), p(cap), p(clen), p(st),
                                          L.MEM_DEVICE, stream) == 0
        assert lib.lz4flex_decompress_batch(ctx, p(comp), p(comp_off), p(clen), n, p(back), p(in_off), p(in_len), p(blen), p(bst),
                                            None, L.MEM_DEVICE, stream) == 0
        torch.cuda.synchronize()
    finally:
        lib.lz4flex_ctx_destroy(ctx)
    assert int((st != 0).sum().item()) == 0 and int((bst != 0).sum().item()) == 0
    assert int((blen != B).sum().item()) == 0
    bad = torch.nonzero((back.view(n, B) != src.view(n, B)).any(dim=1)).flatten().tolist()
    assert not bad, "blocks that differ from the input: %r" % bad[:10]
