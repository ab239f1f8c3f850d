#!/usr/bin/env python3
"""Experiment (GPU box): the split decoder with 16 / 32 / 64 blocks per workgroup at batch sizes between its dispatch thresholds
(launch_decompress_split picks by batch size: lz4_device.h DISPATCH_SPLIT_*), against the default dispatch.  Not the reported bench."""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks", type=int, nargs="+", default=[4096, 5121, 6144, 8192, 8193, 10240, 12288, 14336, 16383, 16384])
    ap.add_argument("--data", default="json")
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    import torch
    import oracle_api as O
    from lz4_flex_amd import _lib as L, workloads
    lib = L.load()
    dev = torch.device("cuda", 0)
    B = 65536
    p = lambda t: C.c_void_p(t.data_ptr())
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for n in args.blocks:
        fx = "compression_66k_JSON" if args.data == "json" else "compression_65k"
        src = workloads.json_tiles(O.fixture_plain(fx), n * B, device=dev) if args.data != "log" else workloads.log_stream(0, n * B, device=dev)
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
        ctx = C.c_void_p()
        assert lib.lz4flex_ctx_create(C.byref(ctx), 0) == 0
        assert lib.lz4flex_compress_batch(ctx, p(src), p(in_off), p(in_len), None, n, p(comp), p(comp_off), p(cap), p(clen), p(st), L.MEM_DEVICE, stream) == 0
        torch.cuda.synchronize()
        row = []
        for variant, bpw in ((0, 0), (4, 16), (4, 32), (4, 64), (5, 0)):
            assert lib.lz4flex_set_tuning(ctx, b"decompress_variant", variant) == 0
            assert lib.lz4flex_set_tuning(ctx, b"decompress_blocks_per_wg", bpw) == 0
            ts = []
            for r in range(args.reps + 1):
                back.zero_(); bst.fill_(-1)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                assert lib.lz4flex_decompress_batch(ctx, p(comp), p(comp_off), p(clen), n, p(back), p(in_off), p(in_len), p(blen), p(bst), None, L.MEM_DEVICE, stream) == 0
                e1.record()
                torch.cuda.synchronize()
                if r:
                    ts.append(e0.elapsed_time(e1))
            ok = int((bst != 0).sum().item()) == 0 and torch.equal(back, src)
            ts.sort()
            row.append("v%d/%d %s%.3f" % (variant, bpw, "" if ok else "WRONG ", ts[len(ts) // 2]))
        print("%s %6d blocks:  %s" % (args.data, n, "   ".join(row)), flush=True)
        lib.lz4flex_ctx_destroy(ctx)


if __name__ == "__main__":
    main()
