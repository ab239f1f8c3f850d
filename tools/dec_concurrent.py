#!/usr/bin/env python3
"""Experiment (GPU box): a medium batch decoded by TWO kernels at once -- the split decoder on one part of the blocks, the
wavefront-per-block decoder on the rest, on two streams -- against either kernel alone.  The split decoder leaves most of a
CU's issue slots empty (its time is its slowest block's chain); does a second kernel fit into them?  Not the reported bench."""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks", type=int, nargs="+", default=[6144, 8192, 12288])
    ap.add_argument("--data", default="json")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--pairs", default="4:5,4:6,4:10,4:4,5:5")
    args = ap.parse_args()
    import torch
    import oracle_api as O
    from lz4_flex_amd import _lib as L, workloads
    lib = L.load()
    dev = torch.device("cuda", 0)
    B = 65536
    p = lambda t: C.c_void_p(t.data_ptr())
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for n in args.blocks:
        fx = "compression_66k_JSON" if args.data == "json" else "compression_65k"
        src = workloads.json_tiles(O.fixture_plain(fx), n * B, device=dev)
        stride = (20 + B * 110 // 100 + 63) // 64 * 64
        comp = torch.empty(n * stride, dtype=torch.uint8, device=dev)
        back = torch.empty(n * B, dtype=torch.uint8, device=dev)
        ar = torch.arange(n, dtype=torch.int64, device=dev)
        in_off, comp_off = ar * B, ar * stride
        in_len = torch.full((n,), B, dtype=torch.int32, device=dev)
        cap = torch.full((n,), stride, dtype=torch.int32, device=dev)
        clen = torch.zeros(n, dtype=torch.int32, device=dev)
        st = torch.full((n,), -1, dtype=torch.int32, device=dev)
        blen = torch.zeros(n, dtype=torch.int32, device=dev)
        bst = torch.full((n,), -1, dtype=torch.int32, device=dev)
        ctx0 = C.c_void_p()
        assert lib.lz4flex_ctx_create(C.byref(ctx0), 0) == 0
        assert lib.lz4flex_compress_batch(ctx0, p(src), p(in_off), p(in_len), None, n, p(comp), p(comp_off), p(cap), p(clen), p(st),
                                          L.MEM_DEVICE, C.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
        torch.cuda.synchronize()

        def dec(ctx, lo, hi, stream):
            k = hi - lo
            if k == 0:
                return
            assert lib.lz4flex_decompress_batch(ctx, p(comp), p(comp_off[lo:]), p(clen[lo:]), k, p(back), p(in_off[lo:]), p(in_len[lo:]), p(blen[lo:]),
                                                p(bst[lo:]), None, L.MEM_DEVICE, C.c_void_p(stream.cuda_stream)) == 0, L.last_error()

        for pair in args.pairs.split(","):
            va, vb = (int(x) for x in pair.split(":"))
            ca, cb = C.c_void_p(), C.c_void_p()
            assert lib.lz4flex_ctx_create(C.byref(ca), 0) == 0 and lib.lz4flex_ctx_create(C.byref(cb), 0) == 0
            assert lib.lz4flex_set_tuning(ca, b"decompress_variant", va) == 0 and lib.lz4flex_set_tuning(cb, b"decompress_variant", vb) == 0
            row = []
            for num, den in ((1, 1), (3, 4), (2, 3), (1, 2), (1, 3), (1, 4), (0, 1)):
                n1 = n * num // den // 64 * 64
                ts = []
                for r in range(args.reps + 1):
                    back.zero_(); bst.fill_(-1)
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    cur = torch.cuda.current_stream()
                    e0.record(cur)
                    s1.wait_event(e0); s2.wait_event(e0)
                    dec(ca, 0, n1, s1); dec(cb, n1, n, s2)
                    d1, d2 = torch.cuda.Event(), torch.cuda.Event()
                    d1.record(s1); d2.record(s2)
                    cur.wait_event(d1); cur.wait_event(d2)
                    e1.record(cur)
                    torch.cuda.synchronize()
                    if r:
                        ts.append(e0.elapsed_time(e1))
                ok = int((bst != 0).sum().item()) == 0 and torch.equal(back, src)
                ts.sort()
                row.append("%d/%d %s%.3f" % (num, den, "" if ok else "WRONG ", ts[len(ts) // 2]))
            print("%s %6d blocks  v%d on the first part | v%d on the rest:  %s" % (args.data, n, va, vb, "  ".join(row)), flush=True)
            lib.lz4flex_ctx_destroy(ca); lib.lz4flex_ctx_destroy(cb)
        lib.lz4flex_ctx_destroy(ctx0)


if __name__ == "__main__":
    main()
