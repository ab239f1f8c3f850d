#!/usr/bin/env python3
"""Quick timing of the batched encoder / decoder on the configs[1] workload (not the reported bench: bench.py)."""
import argparse
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks", type=int, default=16384)
    ap.add_argument("--mode", type=int, default=0)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--data", default="json")
    ap.add_argument("--prof", action="store_true")
    ap.add_argument("--dec", type=int, default=0, help="decompress_variant")
    ap.add_argument("--block", type=int, default=65536, help="block size in bytes (config 4: --data log --block 4194304 --blocks 256)")
    args = ap.parse_args()
    import torch
    import oracle_api as O
    from lz4_flex_amd import _lib as L, workloads
    lib = L.load()
    dev = torch.device("cuda", 0)
    n, B = args.blocks, args.block
    if args.data == "json":
        src = workloads.json_tiles(O.fixture_plain("compression_66k_JSON"), n * B, device=dev)
    elif args.data == "text":
        src = workloads.json_tiles(O.fixture_plain("compression_65k"), n * B, device=dev)
    elif args.data == "log":
        src = workloads.log_stream(0, n * B, device=dev)
    elif args.data == "zeros":
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
    ctx = C.c_void_p()
    assert lib.lz4flex_ctx_create(C.byref(ctx), 0) == 0
    assert lib.lz4flex_set_tuning(ctx, b"compress_mode", args.mode) == 0
    if args.dec:
        assert lib.lz4flex_set_tuning(ctx, b"decompress_variant", args.dec) == 0
    p = lambda t: C.c_void_p(t.data_ptr())
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def comp_once():
        assert lib.lz4flex_compress_batch(ctx, p(src), p(in_off), p(in_len), None, n, p(comp), p(comp_off), p(cap), p(clen), p(st),
                                          L.MEM_DEVICE | (L.MEM_BIG_BLOCKS if B > 65536 else 0), stream) == 0, L.last_error()

    def dec_once():
        assert lib.lz4flex_decompress_batch(ctx, p(comp), p(comp_off), p(clen), n, p(back), p(in_off), p(in_len), p(blen), p(bst),
                                            None, L.MEM_DEVICE, stream) == 0, L.last_error()

    comp_once(); dec_once(); torch.cuda.synchronize()
    if args.prof:
        lib.lz4flex_debug_wave_prof.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        vals = (C.c_ulonglong * 32)()
        assert lib.lz4flex_debug_wave_prof(ctx, 1, None) == 0
        comp_once(); torch.cuda.synchronize()
        assert lib.lz4flex_debug_wave_prof(ctx, 0, vals) == 0
        v = list(vals)
        nw = max(v[7], 1)
        names = ["idx_busy", "idx_barrier", "match(sum 8 waves)", "wait_after_match", "place", "load_window", "wait_after_load"]
        print("per window cycles: " + ", ".join("%s=%.0f" % (nm, x / nw) for nm, x in zip(names, v)) + " windows=%d" % v[7], flush=True)
        if os.environ.get("LZ4W_PROF_WORKERS"):                  # -DLZ4W_PROF_WORKERS variant build
            print("matching cycles per window by worker: " + ", ".join("w%d=%.0f" % (i, x / nw) for i, x in enumerate(v[16:32]) if x), flush=True)
        elif v[13]:
            sn = ["heads", "compact+lengths", "scan", "walk", "merge"]
            print("per superstep cycles (LZ4W_PROF_STEPS build): " + ", ".join("%s=%.0f" % (nm, x / v[13]) for nm, x in zip(sn, v[8:13])) +
                  " loop+cand-wait=%.0f encode_seqs=%.0f supersteps/window=%.1f" % (v[14] / v[13], v[15] / v[13], v[13] / nw), flush=True)
    ok = int((st != 0).sum().item()) == 0 and int((bst != 0).sum().item()) == 0 and torch.equal(back, src)
    if hasattr(lib, "lz4flex_debug_wdec_prof"):          # -DLZ4D_PROF variant build
        lib.lz4flex_debug_wdec_prof.argtypes = [C.c_void_p, C.c_int]
        v = (C.c_ulonglong * 16)()
        lib.lz4flex_debug_wdec_prof(None, 1)
        dec_once(); torch.cuda.synchronize()
        lib.lz4flex_debug_wdec_prof(v, 0)
        v = list(v); nw = max(v[8], 1)
        names = ["input", "spec-parse", "hop", "place+classify", "phaseA", "phaseB", "flush"]
        print("wave decoder, cycles per window: " + ", ".join("%s=%.0f" % (a_, x / nw) for a_, x in zip(names, v)) +
              "; per window: sequences %.1f, phase-B matches %.1f (easy %.1f), far-LP %.1f, near-LP %.1f; windows %d" %
              (v[9] / nw, v[10] / nw, v[11] / nw, v[12] / nw, v[13] / nw, v[8]), flush=True)
    if hasattr(lib, "lz4flex_debug_fused_prof"):           # -DLZ4F_PROF variant build
        lib.lz4flex_debug_fused_prof.argtypes = [C.c_void_p, C.c_int]
        v = (C.c_ulonglong * 16)()
        lib.lz4flex_debug_fused_prof(None, 1)
        dec_once(); torch.cuda.synchronize()
        lib.lz4flex_debug_fused_prof(v, 0)
        v = list(v); wg = max((n + 63) // 64, 1)
        print("fused decoder, per workgroup: parser %.0f cycles / %.0f steps; emitter %.0f cycles / %.0f iterations, per block %.0f placed a piece, %.0f found the "
              "step queue full, %.0f had nothing to do; quads %.0f cycles / %.0f turns per wavefront, per block %.0f turns with steps" %
              (v[0] / wg, v[1] / wg, v[5] / wg, v[6] / wg, v[7] / n, v[8] / n, v[9] / n, v[2] / wg / 4, v[3] / wg / 4, v[4] / n), flush=True)
    if hasattr(lib, "lz4flex_debug_phase_split"):          # -DLZ4FLEX_PROFILE_PHASES variant build
        lib.lz4flex_debug_phase_split.argtypes = [C.c_void_p, C.c_int]
        v = (C.c_ulonglong * 16)()
        lib.lz4flex_debug_phase_split(None, 1)
        dec_once(); torch.cuda.synchronize()
        lib.lz4flex_debug_phase_split(v, 0)
        v = list(v)
        pw = max(n // 64, 1)
        print("split decoder: parser wave cycles %.0f, steps per parser %.0f (%.0f cycles per step); lane steps: live %d, queue full %.1f%%, "
              "ring not ready %.1f%%, exact path %.2f%%, records %d; copier wave cycles %.0f, iterations %.0f, write-backs %.0f, cycles in write-backs %.0f" %
              (v[0] / pw, v[1] / pw, v[0] / max(v[1], 1), v[2], 100.0 * v[3] / max(v[2], 1), 100.0 * v[4] / max(v[2], 1), 100.0 * v[5] / max(v[2], 1), v[6],
               v[8] / (8 * pw), v[9] / (8 * pw), v[10] / (8 * pw), v[14] / (8 * pw)), flush=True)
    tc, td = [], []
    for _ in range(args.reps):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record(); comp_once(); e[1].record(); dec_once(); e[2].record()
        torch.cuda.synchronize()
        tc.append(e[0].elapsed_time(e[1])); td.append(e[1].elapsed_time(e[2]))
    ratio = float(clen.to(torch.int64).sum().item()) / (n * B)
    print("data=%s blocks=%d mode=%d round_trip_ok=%s ratio=%.4f compress_ms=%s decompress_ms=%s" %
          (args.data, n, args.mode, ok, ratio, ["%.3f" % x for x in tc], ["%.3f" % x for x in td]), flush=True)


if __name__ == "__main__":
    main()
