#!/usr/bin/env python3
"""Decode time of every decoder kernel over a matrix of batch shapes (one process, one GPU call): which kernel should
launch_decompress_fast (csrc/capi.cpp) pick for which shape.  Not the reported bench (bench.py).

  python tools/dec_shapes.py                      # the default matrix
  python tools/dec_shapes.py --shapes json:65536:256,log:4194304:256 --variants 6,7
"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

DEFAULT = ("json:65536:1,json:65536:64,json:65536:160,json:65536:256,json:65536:512,json:65536:1024,json:65536:2304,json:65536:4096,"
           "json:65536:8192,text:65536:160,text:65536:1024,log:65536:1024,log:4194304:16,log:4194304:256,log:16777216:1,json:1048576:64,"
           "zeros:4194304:16,random:65536:256")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default=DEFAULT)
    ap.add_argument("--variants", default="7,13,4")
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    import torch
    import oracle_api as O
    from lz4_flex_amd import _lib as L, workloads
    lib = L.load()
    dev = torch.device("cuda", 0)
    variants = [int(v) for v in args.variants.split(",")]
    ctx = C.c_void_p()
    assert lib.lz4flex_ctx_create(C.byref(ctx), 0) == 0
    p = lambda t: C.c_void_p(t.data_ptr())
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for shape in args.shapes.split(","):
        data, B, n = shape.split(":")
        B, n = int(B), int(n)
        if data == "json":
            src = workloads.json_tiles(O.fixture_plain("compression_66k_JSON"), n * B, device=dev)
        elif data == "text":
            src = workloads.json_tiles(O.fixture_plain("compression_65k"), n * B, device=dev)
        elif data == "log":
            src = workloads.log_stream(0, n * B, device=dev)
        elif data == "mixed":          # 8 KiB of JSON, 8 KiB of noise, ...: long literal runs between matches
            src = workloads.json_tiles(O.fixture_plain("compression_66k_JSON"), n * B, device=dev)
            noise = torch.randint(0, 256, (n * B,), dtype=torch.uint8, device=dev)
            sel = ((torch.arange(n * B, device=dev) >> 13) & 1).bool()
            src = torch.where(sel, noise, src)
        elif data == "zeros":
            src = torch.zeros(n * B, dtype=torch.uint8, device=dev)
        else:
            src = torch.randint(0, 256, (n * B,), dtype=torch.uint8, device=dev)
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
        assert lib.lz4flex_compress_batch(ctx, p(src), p(in_off), p(in_len), None, n, p(comp), p(comp_off), p(cap), p(clen), p(st),
                                          L.MEM_DEVICE | (L.MEM_BIG_BLOCKS if B > 65536 else 0), stream) == 0, L.last_error()
        torch.cuda.synchronize()
        ratio = float(clen.to(torch.int64).sum().item()) / (n * B)
        row = []
        for v in variants:
            assert lib.lz4flex_set_tuning(ctx, b"decompress_variant", v) == 0
            ts = []
            ok = True
            for r in range(args.reps + 1):
                back.zero_(); bst.fill_(-1)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                assert lib.lz4flex_decompress_batch(ctx, p(comp), p(comp_off), p(clen), n, p(back), p(in_off), p(in_len), p(blen), p(bst),
                                                    None, L.MEM_DEVICE | (L.MEM_BIG_BLOCKS if B > 131072 else 0), stream) == 0, L.last_error()   # (the hint a caller of large blocks gives: include/lz4flex_amd.h)
                e1.record()
                torch.cuda.synchronize()
                if r:
                    ts.append(e0.elapsed_time(e1))
                else:
                    ok = int((bst != 0).sum().item()) == 0 and torch.equal(back, src)
            ts.sort()
            row.append("v%d %s%.3f ms (%.1f GB/s)" % (v, "" if ok else "WRONG ", ts[len(ts) // 2], n * B / ts[len(ts) // 2] / 1e6))
            if v in (7, 8) and hasattr(lib, "lz4flex_debug_pcd_prof"):           # -DLZ4P_PROF variant build
                lib.lz4flex_debug_pcd_prof.argtypes = [C.c_void_p, C.c_int]
                pv = (C.c_ulonglong * 32)()
                lib.lz4flex_debug_pcd_prof(None, 1)
                assert lib.lz4flex_decompress_batch(ctx, p(comp), p(comp_off), p(clen), n, p(back), p(in_off), p(in_len), p(blen), p(bst),
                                                    None, L.MEM_DEVICE | (L.MEM_BIG_BLOCKS if B > 131072 else 0), stream) == 0
                torch.cuda.synchronize()
                lib.lz4flex_debug_pcd_prof(pv, 0)
                pv = list(pv)
                names = ["load", "walk", "resolve", "toklist", "parse+scan", "literals", "search", "matches", "writeback", "giant"]
                tot = max(sum(pv[:10]), 1)
                print("   pcd cycles per block: %.0f k; %s | tiles %.1f rounds/tile %.2f batches %.1f seq/batch %.0f giants %.1f" % (
                    tot / n / 1e3, ", ".join("%s %.1f%%" % (nm, 100.0 * x / tot) for nm, x in zip(names, pv)), pv[16] / n, pv[17] / max(pv[16], 1),
                    pv[18] / n, pv[19] / max(pv[18], 1), pv[20] / n), flush=True)
                print("      wavefront 0 per batch: polling turns with a ready match %.1f, without %.1f, wavefront copies %.1f" % (
                    pv[21] / max(pv[18], 1), pv[22] / max(pv[18], 1), pv[23] / max(pv[18], 1)), flush=True)
                print("      thread 0: hops per walk round %.1f, cycles per hop %.0f (inside its own loop: %.0f; turns with a rare lane in its wavefront: %.2f of its hops)" %
                      (pv[24] / max(pv[17], 1), pv[1] / max(pv[24], 1), pv[25] / max(pv[24], 1), pv[23] / max(pv[24], 1)), flush=True)
                print("      walk rounds: the first %.0f cycles, the later ones %.0f each" % (pv[21] / max(pv[16], 1), pv[22] / max(pv[17] - pv[16], 1)), flush=True)
                nwb = max(pv[18], 1) * 16
                print("      per wavefront and batch: %.0f cycles in the matches without producers, %.0f polling (%.1f turns with a ready match, %.1f without); "
                      "per batch: %.0f matches with producers, %.0f copied by a whole wavefront" % (pv[26] / nwb, pv[27] / nwb, pv[28] / nwb, pv[29] / nwb,
                                                                                                 pv[31] / max(pv[18], 1), pv[30] / max(pv[18], 1)), flush=True)
                for nm, x, cnt in (("walk round", pv[1], pv[17]), ("resolve round", pv[2], pv[17]), ("batch (4..8)", sum(pv[4:9]), pv[18]), ("matches phase", pv[7], pv[18]), ("giant", pv[9], pv[20])):
                    if cnt:
                        print("      cycles per %s: %.0f" % (nm, x / cnt), flush=True)
        print("%-7s block %8d x %5d  ratio %.3f | %s" % (data, B, n, ratio, " | ".join(row)), flush=True)
        del src, comp, back


if __name__ == "__main__":
    main()
