#!/usr/bin/env python3
"""Time the PLAN kernel alone (lz4flex_debug_plan) on n x 64 KiB tiles compressed by the library's encoder.  Kernel experiments
(variant builds through --lib) never reach the replay kernel this way: a damaged plan cannot hang anything."""
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
    ap.add_argument("--data", default="json")
    ap.add_argument("--libs", nargs="*", default=[])
    args = ap.parse_args()
    import torch
    import oracle_api as O
    from lz4_flex_amd import _lib as L, workloads
    base = L.load()
    dev = torch.device("cuda", 0)
    n, B = args.blocks, 65536
    plain = O.fixture_plain("compression_66k_JSON" if args.data == "json" else "compression_65k")
    src = workloads.json_tiles(plain, n * B, device=dev)
    stride = 72128
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
    assert base.lz4flex_compress_batch(ctx, p(src), p(in_off), p(in_len), None, n, p(comp), p(comp_off), p(cap), p(clen), p(st), L.MEM_DEVICE, stream) == 0
    torch.cuda.synchronize()
    bcap = torch.full((n,), B, dtype=torch.int32, device=dev)
    for path in [None] + args.libs:
        lib = base if path is None else C.CDLL(path)
        lib.lz4flex_debug_plan.restype = C.c_int
        lib.lz4flex_debug_plan.argtypes = [C.c_void_p] * 5 + [C.c_uint] + [C.c_void_p] * 5
        lib.lz4flex_debug_plan_slot_words.restype = C.c_uint
        slot = lib.lz4flex_debug_plan_slot_words()
        plans = torch.zeros(n * 32, dtype=torch.uint8, device=dev)
        words = torch.zeros(n * slot, dtype=torch.int32, device=dev)
        olen = torch.zeros(n, dtype=torch.int32, device=dev)
        pst = torch.full((n,), -1, dtype=torch.int32, device=dev)
        ts = []
        for r in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = lib.lz4flex_debug_plan(p(comp), p(comp_off), p(clen), p(in_off), p(bcap), n, p(plans), p(words), p(olen), p(pst), stream)
            e1.record()
            torch.cuda.synchronize()
            assert rc == 0
            if r:
                ts.append(e0.elapsed_time(e1))
        print("%-60s plan kernel, %d %s blocks: min %.3f ms; blocks with a plan %d, decoded lengths right %d" % (
            "default" if path is None else path.split("/")[-2], n, args.data, min(ts), int((pst == 0).sum().item()), int((olen == B).sum().item())), flush=True)


if __name__ == "__main__":
    main()
