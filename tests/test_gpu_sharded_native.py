"""GPU (-m gpu): the C ABI's multi-rank frame entry points (csrc/sharded.cpp) with MORE THAN ONE RANK on one device.

RCCL refuses two ranks on a device and this build never had a multi-GPU node, so the ranks are threads of this process and
the communicator is tests/sim/mock_rccl.cpp, loaded through LZ4FLEX_RCCL_LIB: the six ncclXxx entry points with host-side
rendezvous and device-to-device copies, which also checks that every rank takes part in every collective with the same
arguments and that every send meets a receive of the same size.  What this exercises is the library's side of the exchange:
sizes all-gather, prefix sums, the segment gather to the root, the block-table broadcast, the byte ranges sent to the ranks,
the verdict broadcasts -- against the oracle's FrameEncoder / FrameDecoder.  What it cannot show is RCCL itself."""
import ctypes as C
import os
import subprocess
import threading

import numpy as np
import pytest

import oracle_api as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MOCK_SRC = os.path.join(ROOT, "tests", "sim", "mock_rccl.cpp")
MOCK_SO = os.path.join(ROOT, "tests", "sim", "libmock_rccl.so")


@pytest.fixture(scope="module")
def env():
    if not os.path.exists(MOCK_SO) or os.path.getmtime(MOCK_SRC) > os.path.getmtime(MOCK_SO):
        hipcc = "/opt/rocm/bin/hipcc"
        subprocess.check_call([hipcc, "-O2", "-std=c++17", "-fPIC", "-shared", "-x", "hip", "--offload-arch=gfx950", MOCK_SRC, "-o", MOCK_SO],
                              stderr=subprocess.DEVNULL)
    os.environ["LZ4FLEX_RCCL_LIB"] = MOCK_SO             # read at the library's first call with more than one rank
    import torch
    from lz4_flex_amd import _lib as L, sharded, workloads
    lib = L.load()
    assert lib.lz4flex_device_count() >= 1
    mock = C.CDLL(MOCK_SO)
    mock.mock_world_create.restype = C.c_void_p
    mock.mock_world_create.argtypes = [C.c_int]
    mock.mock_comm_create.restype = C.c_void_p
    mock.mock_comm_create.argtypes = [C.c_void_p, C.c_int]
    mock.mock_world_errors.argtypes = [C.c_void_p]
    return lib, L, mock, torch, sharded, workloads


def _run(world, fn):
    """fn(rank) in one thread per rank; exceptions travel back"""
    res, err = [None] * world, []

    def body(r):
        try:
            res[r] = fn(r)
        except BaseException as e:      # noqa: the main thread re-raises
            err.append((r, e))
    th = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in th), "a rank is stuck in the exchange"
    if err:
        raise err[0][1]
    return res


@pytest.mark.parametrize("mode", ["exact", "fast"])
@pytest.mark.parametrize("world", [2, 4])
@pytest.mark.parametrize("bs_code,n_blocks,tail,bc", [(4, 11, 12345, False), (4, 3, 0, True), (7, 5, 128 * 999, False)])
def test_frame_across_ranks_through_the_c_abi(env, world, mode, bs_code, n_blocks, tail, bc):
    lib, L, mock, torch, sharded, workloads = env
    bs = {4: 65536, 7: 4 << 20}[bs_code]
    total = n_blocks * bs + tail - (tail % 128)
    nb_all = (total + bs - 1) // bs
    src = workloads.log_stream(0, total, device="cuda")
    if bc:      # an incompressible block in the middle: stored raw
        src[bs:bs + 5000] = torch.from_numpy(np.random.default_rng(3).integers(0, 256, 5000, dtype=np.uint8)).cuda()
    host = src.cpu().numpy().tobytes()
    torch.cuda.synchronize()
    fic = L.FrameInfoC(0, 0, bs_code, 0, 1 if bc else 0, 0, 0)
    wptr = mock.mock_world_create(world)
    ranges = sharded.partition(nb_all, world)
    cap = int(lib.lz4flex_frame_segment_bound(total, C.byref(fic))) + 64
    frame = torch.zeros(cap, dtype=torch.uint8, device="cuda")
    ctxs, comms = [], []
    for r in range(world):
        ctx = C.c_void_p()
        assert lib.lz4flex_ctx_create(C.byref(ctx), 0) == 0
        assert lib.lz4flex_set_tuning(ctx, b"compress_mode", 1 if mode == "exact" else 0) == 0
        ctxs.append(ctx)
        comms.append(C.c_void_p(mock.mock_comm_create(wptr, r)))
    flens = [C.c_uint64(0) for _ in range(world)]

    def comp(r):
        lo, hi = ranges[r]
        a, b = lo * bs, min(hi * bs, total)
        return lib.lz4flex_frame_compress_sharded(ctxs[r], comms[r], r, world, 0, C.c_void_p(src.data_ptr() + a), max(b - a, 0), lo, C.byref(fic),
                                                  C.c_void_p(frame.data_ptr()) if r == 0 else None, cap if r == 0 else 0, C.byref(flens[r]), None)
    res = _run(world, lambda r: (comp(r), L.last_error()))
    assert [x[0] for x in res] == [0] * world, res
    assert [f.value for f in flens[1:]] == [0] * (world - 1)
    got = frame[:flens[0].value].cpu().numpy().tobytes()
    rc, back, used = O.frame_decompress(got, total)
    assert rc == 0 and back == host and used == len(got), "the oracle's FrameDecoder does not return the stream"
    if mode == "exact":
        rc_o, exp = O.frame_compress(host, block_size=bs_code, block_checksums=bc)
        assert rc_o == 0 and got == exp, "the frame gathered from the ranks != the oracle's FrameEncoder bytes"
    # ---- and back: the root holds the frame, every rank receives and decodes its block range
    outs = [torch.zeros(max(ranges[r][1] - ranges[r][0], 1) * bs, dtype=torch.uint8, device="cuda") for r in range(world)]
    olen = [C.c_uint64(0) for _ in range(world)]
    first = [C.c_uint64(99) for _ in range(world)]
    nblk = [C.c_uint64(99) for _ in range(world)]

    def dec(r):
        return lib.lz4flex_frame_decompress_sharded(ctxs[r], comms[r], r, world, 0, C.c_void_p(frame.data_ptr()) if r == 0 else None,
                                                    flens[0].value if r == 0 else 0, C.c_void_p(outs[r].data_ptr()), int(outs[r].numel()),
                                                    C.byref(olen[r]), C.byref(first[r]), C.byref(nblk[r]), None, None, None)
    assert _run(world, dec) == [0] * world, L.last_error()
    for r in range(world):
        lo, hi = ranges[r]
        a, b = lo * bs, min(hi * bs, total)
        assert (first[r].value, nblk[r].value, olen[r].value) == (lo, hi - lo, max(b - a, 0))
        assert outs[r][:b - a].cpu().numpy().tobytes() == host[a:b]
    # ---- a root whose buffer is too small: every rank learns it before anybody sends
    small = torch.zeros(64, dtype=torch.uint8, device="cuda")

    def comp_small(r):
        lo, hi = ranges[r]
        a, b = lo * bs, min(hi * bs, total)
        return lib.lz4flex_frame_compress_sharded(ctxs[r], comms[r], r, world, 0, C.c_void_p(src.data_ptr() + a), max(b - a, 0), lo, C.byref(fic),
                                                  C.c_void_p(small.data_ptr()) if r == 0 else None, 64 if r == 0 else 0, C.byref(flens[r]), None)
    assert _run(world, comp_small) == [-L.FE_OUTPUT_FULL] * world
    # ---- a call-level failure on ONE rank before the size all-gather (an allocation or a launch that fails: here the batch call,
    # through the library's test hook): the rank says so IN the all-gather, nobody is left waiting, everybody returns its code
    assert lib.lz4flex_set_tuning(ctxs[0], b"debug_fail_next_batch", 1) == 0
    assert _run(world, comp) == [-L.E_HIP] * world
    assert _run(world, comp) == [0] * world                       # (and the communicator is still usable)
    assert mock.mock_world_errors(wptr) == 0, "a collective was called with different arguments on different ranks, or a send met a receive of another size"
    for ctx in ctxs:
        lib.lz4flex_ctx_destroy(ctx)


def test_real_rccl_with_a_communicator_of_one_rank():
    """The REAL librccl, bound by csrc/sharded.cpp's dlopen, with the one communicator a one-GPU box can have: a single rank.
    LZ4FLEX_FORCE_COLLECTIVES=1 makes the entry points go through every collective anyway (size all-gather, verdict and block-table
    broadcasts, the segment gather and the range scatter as a grouped send + receive to itself), and the frames are checked against
    the oracle's FrameEncoder / FrameDecoder.  A process of its own: the library resolves its RCCL once, and this file's other
    tests hand it a mock.  (More than one rank of the real RCCL has never run: no multi-GPU node was available to this build.)"""
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_one_rank.py")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=240)
    out = r.stdout.decode(errors="replace")
    if "RCCL-ONE-RANK SKIP" in out:
        pytest.skip(out.strip().splitlines()[-1])
    assert r.returncode == 0 and "RCCL-ONE-RANK OK" in out, out[-3000:]
