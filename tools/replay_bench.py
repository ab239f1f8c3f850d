#!/usr/bin/env python3
"""The REPLAY kernel alone (lz4_decompress_replay.hip) on plans compiled on the host by the model (tests/sim/plan_model.cpp):
the configs[1] workload (n x 64 KiB tiles, compressed by the library's encoder) is copied to the host, compiled into copy plans
there, and the plans are replayed on the GPU --reps times (events on the launch stream); the output must equal the source.
Also replays the plans of the adversarial batch's valid blocks.  A tool (kernel development), not the reported bench."""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks", type=int, default=16384)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--data", default="json")
    ap.add_argument("--lib", default=None)
    args = ap.parse_args()
    import numpy as np
    import torch
    import oracle_api as O
    import plan_model as M
    from lz4_flex_amd import _lib as L, workloads
    base = L.load()
    lib = base if args.lib is None else C.CDLL(args.lib)
    lib.lz4flex_debug_replay.restype = C.c_int
    lib.lz4flex_debug_replay.argtypes = [C.c_void_p] * 4 + [C.c_uint, C.c_void_p]
    dev = torch.device("cuda", 0)
    n, B = args.blocks, 65536
    plain = O.fixture_plain("compression_66k_JSON" if args.data == "json" else "compression_65k")
    src = workloads.json_tiles(plain, n * B, device=dev)
    stride = (20 + B * 110 // 100 + 63) // 64 * 64
    comp = torch.empty(n * stride, dtype=torch.uint8, device=dev)
    ar = torch.arange(n, dtype=torch.int64, device=dev)
    in_off, comp_off = ar * B, ar * stride
    in_len = torch.full((n,), B, dtype=torch.int32, device=dev)
    cap = torch.full((n,), stride, dtype=torch.int32, device=dev)
    clen = torch.zeros(n, dtype=torch.int32, device=dev)
    st = torch.full((n,), -1, dtype=torch.int32, device=dev)
    p = lambda t: C.c_void_p(t.data_ptr())
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ctx = C.c_void_p()
    assert base.lz4flex_ctx_create(C.byref(ctx), 0) == 0
    assert base.lz4flex_compress_batch(ctx, p(src), p(in_off), p(in_len), None, n, p(comp), p(comp_off), p(cap), p(clen), p(st),
                                       L.MEM_DEVICE, stream) == 0, L.last_error()
    torch.cuda.synchronize()
    assert int((st != 0).sum().item()) == 0
    # plans on the host
    h_comp = comp.cpu().numpy()
    h_off = comp_off.cpu().numpy().astype(np.uint64)
    h_len = clen.cpu().numpy().astype(np.uint32)
    h_ooff = in_off.cpu().numpy().astype(np.uint64)
    h_cap = np.full(n, B, dtype=np.uint32)
    max_words = int(h_len.sum()) * 3 + n * 1024
    words = np.zeros(max_words, dtype=np.uint32)
    plans = np.zeros(n * 32, dtype=np.uint8)
    olen = np.zeros(n, dtype=np.uint32)
    vp = lambda a: C.c_void_p(a.ctypes.data)
    steps = C.c_uint64(0)
    used = M.lib().plan_compile_batch(vp(h_comp), vp(h_off), vp(h_len), vp(h_ooff), vp(h_cap), n, vp(plans), vp(words), max_words, vp(olen),
                                      C.byref(steps))
    assert used > 0 and int((olen != B).sum()) == 0, used
    print("plans: %d words for %d blocks (%.1f steps per block), %.3f x the compressed bytes" % (used, n, steps.value / n, used * 4 / h_len.sum()))
    d_words = torch.from_numpy(words[:used + 64].copy()).to(dev)
    d_plans = torch.from_numpy(plans).to(dev)
    out = torch.zeros(n * B, dtype=torch.uint8, device=dev)
    times = []
    for r in range(args.reps + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = lib.lz4flex_debug_replay(p(comp), p(out), p(d_plans), p(d_words), n, stream)
        e1.record()
        torch.cuda.synchronize()
        assert rc == 0, rc
        if r:
            times.append(e0.elapsed_time(e1))
    ok = bool(torch.equal(out, src))
    if not ok:
        bad = (out.view(n, B) != src.view(n, B)).any(dim=1).nonzero().flatten()
        print("MISMATCH in %d blocks, first %s" % (bad.numel(), bad[:8].tolist()))
        b0 = int(bad[0])
        pos = (out.view(n, B)[b0] != src.view(n, B)[b0]).nonzero().flatten()
        print("  block %d: %d bytes differ, first at %s" % (b0, pos.numel(), pos[:8].tolist()))
    alg = (n * B + int(h_len.sum())) / 1e9
    print("replay %s x %d: min %.3f ms, median %.3f ms  (%.0f GB/s algorithmic at the minimum)  output %s" % (
        args.data, n, min(times), sorted(times)[len(times) // 2], alg / min(times) * 1e3, "== source" if ok else "WRONG"))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
