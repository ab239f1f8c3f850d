"""GPU (-m gpu): the DEFAULT decoder dispatch at every batch size where it changes kernel or geometry.

The size list is not written down here: it is read from the library (lz4flex_get_tuning "dispatch_threshold_<i>" = the table
launch_decompress_fast and launch_decompress_split use, lz4_device.h) -- every threshold T, and T - 1 and T + 1 -- so a threshold
edit cannot leave a size class untested (round 3: a wrong result in batches of 5 121 ... 16 383 blocks, a geometry that had only
ever run on the adversarial batch).  Data: JSON, English text and log-line tiles in 64 KiB blocks, encoded by the reference-exact
encoder (= the oracle's = lz4_flex's block bytes; tests/test_gpu_block.py pins that equality); every block must decode to its
source with status 0 and its length, through the default context settings (variant 0)."""
import ctypes as C

import pytest

import oracle_api as O

pytestmark = pytest.mark.gpu
B = 65536


def thresholds(lib):
    out = []
    for i in range(64):
        v = lib.lz4flex_get_tuning(None, b"dispatch_threshold_%d" % i)
        if v < 0:
            break
        out.append(v)
    assert len(out) >= 5 and out == sorted(out), out
    return out


@pytest.fixture(scope="module")
def env():
    import torch
    from lz4_flex_amd import _lib, workloads
    lib = _lib.load()
    assert lib.lz4flex_device_count() >= 1
    ts = thresholds(lib)
    sizes = sorted({n for t in ts for n in (t - 1, t, t + 1) if n >= 1})
    return lib, _lib, torch, workloads, sizes


def _source(kind, n, torch, workloads):
    dev = torch.device("cuda", 0)
    if kind == "json":
        return workloads.json_tiles(O.fixture_plain("compression_66k_JSON"), n * B, phase=4099, device=dev)
    if kind == "text":
        return workloads.json_tiles(O.fixture_plain("compression_65k"), n * B, phase=77, device=dev)
    return workloads.log_stream(0, n * B, device=dev)


@pytest.mark.parametrize("kind", ["json", "text", "log"])
def test_default_dispatch_at_every_threshold(env, kind):
    lib, L, torch, workloads, sizes = env
    nmax = max(sizes)
    dev = torch.device("cuda", 0)
    src = _source(kind, nmax, torch, workloads)
    stride = 72128
    comp = torch.empty(nmax * stride, dtype=torch.uint8, device=dev)
    ar = torch.arange(nmax, dtype=torch.int64, device=dev)
    in_off, comp_off = (ar * B).contiguous(), (ar * stride).contiguous()
    in_len = torch.full((nmax,), B, dtype=torch.int32, device=dev)
    cap = torch.full((nmax,), stride, dtype=torch.int32, device=dev)
    clen = torch.zeros(nmax, dtype=torch.int32, device=dev)
    st = torch.full((nmax,), -1, dtype=torch.int32, device=dev)
    p = lambda t: C.c_void_p(t.data_ptr())
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ctx = C.c_void_p()
    assert lib.lz4flex_ctx_create(C.byref(ctx), 0) == 0
    try:
        assert lib.lz4flex_set_tuning(ctx, b"compress_mode", 1) == 0          # the reference's bytes
        assert lib.lz4flex_compress_batch(ctx, p(src), p(in_off), p(in_len), None, nmax, p(comp), p(comp_off), p(cap), p(clen), p(st),
                                          L.MEM_DEVICE, stream) == 0, L.last_error()
        torch.cuda.synchronize()
        assert int((st != 0).sum().item()) == 0
        # a sample of the blocks against the oracle's encoder (the exact encoder's contract), so that "reference bytes" is checked here too
        h = comp[:3 * stride].cpu().numpy()
        hl = clen[:3].cpu().numpy()
        hs = src[:3 * B].cpu().numpy().tobytes()
        for i in range(3):
            assert h[i * stride:i * stride + int(hl[i])].tobytes() == O.compress(hs[i * B:(i + 1) * B])
        back = torch.empty(nmax * B, dtype=torch.uint8, device=dev)
        bcap = torch.full((nmax,), B, dtype=torch.int32, device=dev)
        for n in sizes:
            back.zero_()
            blen = torch.zeros(n, dtype=torch.int32, device=dev)
            bst = torch.full((n,), -1, dtype=torch.int32, device=dev)
            assert lib.lz4flex_decompress_batch(ctx, p(comp), p(comp_off), p(clen), n, p(back), p(in_off), p(bcap), p(blen), p(bst), None,
                                                L.MEM_DEVICE, stream) == 0, L.last_error()
            torch.cuda.synchronize()
            assert int((bst != 0).sum().item()) == 0, (kind, n, bst[bst != 0][:4].tolist())
            assert int((blen != B).sum().item()) == 0, (kind, n)
            if not torch.equal(back[:n * B], src[:n * B]):
                bad = (back[:n * B].view(n, B) != src[:n * B].view(n, B)).any(dim=1).nonzero().flatten()
                raise AssertionError("%s, %d blocks: %d blocks differ, first %s" % (kind, n, bad.numel(), bad[:8].tolist()))
            assert int(back[n * B:(n + 1) * B].max().item()) == 0 if n < nmax else True, "bytes written behind the batch"
    finally:
        lib.lz4flex_ctx_destroy(ctx)
