#!/usr/bin/env python3
"""Time and check several builds of the throughput encoder in ONE process (the encoder's counterpart of tools/dec_variants.py):
every path given is a variant library built by `python -m lz4_flex_amd.build --variant NAME -D...` (linked -Bsymbolic).
For each: (1) a handful of inputs (fixtures, runs, short periods, random data, multi-window blocks) go through the scalar
entry point; the oracle (lz4_flex's decoder) must return the input, and the bytes are compared with the scalar model
tests/sim/wave_encoder_model.c ("== model" / "own parse": a variant that changes the parse on purpose differs, and is still
checked by the decoder); (2) the configs[1] workload (16 384 x 64 KiB JSON tiles, or --data text / random / zeros) is
compressed --reps times, timed with events on the launch stream, decoded by the DEFAULT library and compared with the source;
the ratio is printed.  Not the reported bench (bench.py)."""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def bind(path):
    from lz4_flex_amd import _lib as L
    lib = C.CDLL(path)
    for name in ("lz4flex_ctx_create", "lz4flex_ctx_destroy", "lz4flex_set_tuning", "lz4flex_compress_batch", "lz4flex_compress_into",
                 "lz4flex_get_maximum_output_size", "lz4flex_build_id", "lz4flex_last_error"):
        res, args = L.SIGNATURES[name]
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    return lib


def small_inputs():
    import random
    import corpus
    import oracle_api as O
    rnd = random.Random(11)
    out = [O.fixture_plain(s) for s in corpus.FIXTURES]
    out += [b"", b"a", bytes(13), bytes(70000), b"ab" * 40000, bytes(range(7)) * 12000, bytes(range(20)) * 5000,
            bytes(rnd.getrandbits(8) for _ in range(70000)), bytes(rnd.choice(b"abc") for _ in range(150000)),
            O.fixture_plain("compression_65k") * 3, O.fixture_plain("compression_66k_JSON")[:65536]]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--blocks", type=int, default=16384)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--data", default="json", choices=["json", "text", "random", "zeros"])
    ap.add_argument("--no-small", action="store_true", help="skip the small inputs (scalar calls: few blocks, windows dealt to workgroups)")
    args = ap.parse_args()
    import torch
    import oracle_api as O
    import wave_model
    from lz4_flex_amd import _lib as L, workloads
    base = L.load()
    dev = torch.device("cuda", 0)
    n, B = args.blocks, 65536
    if args.data == "json":
        src = workloads.json_tiles(O.fixture_plain("compression_66k_JSON"), n * B, device=dev)
    elif args.data == "text":
        src = workloads.json_tiles(O.fixture_plain("compression_65k"), n * B, device=dev)
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
    p = lambda t: C.c_void_p(t.data_ptr())
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ctx0 = C.c_void_p()
    assert base.lz4flex_ctx_create(C.byref(ctx0), 0) == 0
    inputs = [] if args.no_small else small_inputs()
    import re
    models_by_nseg = {}

    for path in args.libs:
        lib = bind(path)
        mw = re.search(r"_w(\d+)", os.path.basename(os.path.dirname(path)))      # a variant named ..._w10... was built with -DLZ4W_WORKERS=10: the model gets that segment count
        nseg = int(mw.group(1)) if mw else wave_model.NSEG
        if nseg not in models_by_nseg:
            models_by_nseg[nseg] = [wave_model.compress(d, nseg=nseg) for d in inputs]
        models = models_by_nseg[nseg]
        assert lib.lz4flex_set_tuning(None, b"compress_mode", 0) == 0      # this library's default context (the scalar calls)
        tag = "%s [%s]" % (os.path.basename(os.path.dirname(path)), lib.lz4flex_build_id().decode())
        # (1) small inputs through the scalar entry point
        bad, same = [], 0
        for i, (d, m) in enumerate(zip(inputs, models)):
            capn = lib.lz4flex_get_maximum_output_size(len(d))
            out = C.create_string_buffer(capn)
            r = lib.lz4flex_compress_into(d, len(d), out, capn)
            if r < 0:
                bad.append("input %d (%d bytes): error %d %s" % (i, len(d), r, lib.lz4flex_last_error().decode(errors="replace")))
                continue
            c = out.raw[:r]
            if O.decompress(c, len(d)) != ("ok", d):
                bad.append("input %d (%d bytes): the oracle does not return the input" % (i, len(d)))
            same += c == m
        # (2) the bench workload
        ctx = C.c_void_p()
        assert lib.lz4flex_ctx_create(C.byref(ctx), 0) == 0
        assert lib.lz4flex_set_tuning(ctx, b"compress_mode", 0) == 0

        def comp_once():
            assert lib.lz4flex_compress_batch(ctx, p(src), p(in_off), p(in_len), None, n, p(comp), p(comp_off), p(cap), p(clen), p(st),
                                              L.MEM_DEVICE, stream) == 0, lib.lz4flex_last_error()
        comp_once(); comp_once(); torch.cuda.synchronize()
        assert base.lz4flex_decompress_batch(ctx0, p(comp), p(comp_off), p(clen), n, p(back), p(in_off), p(in_len), p(blen), p(bst), None,
                                             L.MEM_DEVICE, stream) == 0, L.last_error()
        torch.cuda.synchronize()
        ok = int((st != 0).sum().item()) == 0 and int((bst != 0).sum().item()) == 0 and torch.equal(back, src)
        ratio = float(clen.to(torch.int64).sum().item()) / (n * B)
        ts = []
        for _ in range(args.reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); comp_once(); e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        print("%-34s small inputs %s (%d of %d == model)  bench round trip %s  ratio %.4f  ms %s" %
              (tag, "OK" if not bad else "FAIL %d" % len(bad), same, len(inputs), ok, ratio, " ".join("%.3f" % t for t in ts)), flush=True)
        for b in bad[:6]:
            print("    " + b, flush=True)
        lib.lz4flex_ctx_destroy(ctx)


if __name__ == "__main__":
    main()
