#!/usr/bin/env python3
"""Time and check several builds of the library's split decoder in ONE process (GPU minutes are scarce): every path given is a
variant library built by `python -m lz4_flex_amd.build --variant NAME -D...` (linked -Bsymbolic, so each library calls its own
kernels).  For each: (1) an adversarial batch -- every prefix of a small block, single-byte corruptions, short sinks, runs,
short-period data, random data -- decoded with the split decoder (64 blocks per workgroup) must equal the oracle's result
block by block (bytes, length, status, OutputTooSmall detail); (2) the configs[1] workload (16 384 x 64 KiB JSON tiles,
compressed once by the default library) is decoded --reps times, timed with events on the launch stream, and compared with
the source.  Not the reported bench (bench.py)."""
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
    for name in ("lz4flex_ctx_create", "lz4flex_ctx_destroy", "lz4flex_set_tuning", "lz4flex_decompress_batch", "lz4flex_build_id",
                 "lz4flex_last_error"):
        res, args = L.SIGNATURES[name]
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    return lib


def adversarial():
    """[(compressed bytes, sink capacity)] -- tests/corpus.py::adversarial_blocks, the batch of tests/test_gpu_block.py"""
    import corpus
    return corpus.adversarial_blocks()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--blocks", type=int, default=16384)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--data", default="json")
    ap.add_argument("--variant", default="4", help="decompress_variant(s) to run with every library, comma separated (4 = split decoder with 64 blocks per workgroup, 13 = sequence decoder)")
    args = ap.parse_args()
    import numpy as np
    import torch
    import oracle_api as O
    from lz4_flex_amd import _lib as L, workloads
    base = L.load()
    dev = torch.device("cuda", 0)
    n, B = args.blocks, 65536
    src = workloads.json_tiles(O.fixture_plain("compression_66k_JSON" if args.data == "json" else "compression_65k"), n * B, device=dev)
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
    ctx0 = C.c_void_p()
    assert base.lz4flex_ctx_create(C.byref(ctx0), 0) == 0
    assert base.lz4flex_compress_batch(ctx0, p(src), p(in_off), p(in_len), None, n, p(comp), p(comp_off), p(cap), p(clen), p(st),
                                       L.MEM_DEVICE, stream) == 0, L.last_error()
    torch.cuda.synchronize()
    assert int((st != 0).sum().item()) == 0

    # the adversarial batch and the oracle's verdicts
    cases = adversarial()
    want = [O.decompress(c, k) for c, k in cases]
    a_in = np.frombuffer(b"".join(c for c, _ in cases) + bytes(64), dtype=np.uint8)
    a_ioff = np.cumsum([0] + [len(c) for c, _ in cases[:-1]]).astype(np.uint64)
    a_ilen = np.array([len(c) for c, _ in cases], dtype=np.uint32)
    a_cap = np.array([k for _, k in cases], dtype=np.uint32)
    a_ooff = np.cumsum([0] + [k + 64 for _, k in cases[:-1]]).astype(np.uint64)
    out_bytes = int(a_ooff[-1]) + int(a_cap[-1]) + 64
    t_in, t_ioff, t_ilen = torch.from_numpy(a_in.copy()).to(dev), torch.from_numpy(a_ioff.astype(np.int64)).to(dev), torch.from_numpy(a_ilen.astype(np.int32)).to(dev)
    t_ooff, t_cap = torch.from_numpy(a_ooff.astype(np.int64)).to(dev), torch.from_numpy(a_cap.astype(np.int32)).to(dev)
    na = len(cases)

    for path, variant in [(q, int(v)) for q in args.libs for v in args.variant.split(",")]:
        lib = bind(path)
        ctx = C.c_void_p()
        assert lib.lz4flex_ctx_create(C.byref(ctx), 0) == 0
        assert lib.lz4flex_set_tuning(ctx, b"decompress_variant", variant) == 0
        if variant == 4:
            assert lib.lz4flex_set_tuning(ctx, b"decompress_blocks_per_wg", 64) == 0
        tag = "%s [%s] v%d" % (os.path.basename(os.path.dirname(path)), lib.lz4flex_build_id().decode(), variant)
        # (1) adversarial batch
        t_out = torch.full((out_bytes,), 0xA5, dtype=torch.uint8, device=dev)
        t_ol = torch.zeros(na, dtype=torch.int32, device=dev)
        t_st = torch.full((na,), -1, dtype=torch.int32, device=dev)
        t_det = torch.zeros(2 * na, dtype=torch.int64, device=dev)
        rc = lib.lz4flex_decompress_batch(ctx, p(t_in), p(t_ioff), p(t_ilen), na, p(t_out), p(t_ooff), p(t_cap), p(t_ol), p(t_st), p(t_det),
                                          L.MEM_DEVICE, stream)
        torch.cuda.synchronize()
        bad = []
        if rc != 0:
            bad.append("rc=%d %s" % (rc, lib.lz4flex_last_error().decode(errors="replace")))
        else:
            h_out, h_ol, h_st, h_det = t_out.cpu().numpy(), t_ol.cpu().numpy(), t_st.cpu().numpy(), t_det.cpu().numpy()
            for i, ((c, k), w) in enumerate(zip(cases, want)):
                o = int(a_ooff[i])
                if w[0] == "ok":
                    if h_st[i] != 0 or h_ol[i] != len(w[1]) or h_out[o:o + len(w[1])].tobytes() != w[1]:
                        bad.append("case %d (in %d cap %d): want ok/%d got st %d len %d" % (i, len(c), k, len(w[1]), h_st[i], h_ol[i]))
                else:
                    name = O.ERR_NAMES.get(int(h_st[i]), str(h_st[i]))
                    if name != w[0] or (w[0] == "OutputTooSmall" and (int(h_det[2 * i]), int(h_det[2 * i + 1])) != tuple(w[1])):
                        bad.append("case %d (in %d cap %d): want %s %s got %s (%d, %d)" % (i, len(c), k, w[0], w[1], name, h_det[2 * i], h_det[2 * i + 1]))
                if h_out[o + k:o + k + 64].tobytes() != b"\xA5" * 64:
                    bad.append("case %d: wrote behind its sink" % i)
        # (2) the bench workload
        back = torch.empty(n * B, dtype=torch.uint8, device=dev)
        blen = torch.zeros(n, dtype=torch.int32, device=dev)
        bst = torch.full((n,), -1, dtype=torch.int32, device=dev)

        def dec_once():
            assert lib.lz4flex_decompress_batch(ctx, p(comp), p(comp_off), p(clen), n, p(back), p(in_off), p(in_len), p(blen), p(bst), None,
                                                L.MEM_DEVICE, stream) == 0, lib.lz4flex_last_error()
        dec_once(); dec_once(); torch.cuda.synchronize()
        ok = int((bst != 0).sum().item()) == 0 and torch.equal(back, src)
        if not ok:
            neq = (back != src).view(n, B).any(dim=1).nonzero().flatten()
            print("    bench: %d blocks with a status, %d blocks with wrong bytes, first %s" % (int((bst != 0).sum().item()), neq.numel(), neq[:8].tolist()), flush=True)
            if neq.numel():
                b0 = int(neq[0]); d = (back.view(n, B)[b0] != src.view(n, B)[b0]).nonzero().flatten()
                print("    block %d: %d wrong bytes, first at %s, last at %d; got %s want %s" % (b0, d.numel(), d[:8].tolist(), int(d[-1]),
                      bytes(back.view(n, B)[b0][int(d[0]) - 8:int(d[0]) + 24].tolist()), bytes(src.view(n, B)[b0][int(d[0]) - 8:int(d[0]) + 24].tolist())), flush=True)
        if variant >= 5:      # how many blocks did the first pass leave to the reference-order kernel?
            assert lib.lz4flex_set_tuning(ctx, b"decompress_second_pass", 0) == 0
            dec_once(); torch.cuda.synchronize()
            print("    first pass left %d of %d bench blocks to the second pass" % (int((bst != 0).sum().item()), n), flush=True)
            assert lib.lz4flex_set_tuning(ctx, b"decompress_second_pass", 1) == 0
            dec_once(); torch.cuda.synchronize()
        if hasattr(lib, "lz4flex_debug_phase_split"):
            lib.lz4flex_debug_phase_split.argtypes = [C.c_void_p, C.c_int]
            v = (C.c_ulonglong * 16)()
            lib.lz4flex_debug_phase_split(None, 1)
            dec_once(); torch.cuda.synchronize()
            lib.lz4flex_debug_phase_split(v, 0)
            v = list(v)
            pw = max(n // 64, 1)
            cw = max(v[9], 1)
            print("  phases: parser wave cycles %.0f, steps %.0f (%.0f cycles per step); lane steps live %d, queue full %.1f%%, ring not ready %.1f%%, "
                  "exact %.2f%%, records %d; copier: sum of wave cycles per workgroup %.0f, iterations per workgroup %.0f, write-back visits %.0f, cycles in them %.0f; "
                  "group-steps piece %d idle %d blocked %d" %
                  (v[0] / pw, v[1] / pw, v[0] / max(v[1], 1), v[2], 100.0 * v[3] / max(v[2], 1), 100.0 * v[4] / max(v[2], 1), 100.0 * v[5] / max(v[2], 1), v[6],
                   v[8] / pw, v[9] / pw, v[10] / pw, v[14] / pw, v[11], v[12], v[13]), flush=True)
        if variant == 13 and hasattr(lib, "lz4flex_debug_seq_prof"):
            lib.lz4flex_debug_seq_prof.argtypes = [C.c_void_p, C.c_int]
            v = (C.c_ulonglong * 32)()
            lib.lz4flex_debug_seq_prof(None, 1)
            dec_once(); torch.cuda.synchronize()
            lib.lz4flex_debug_seq_prof(v, 0)
            v = [x / n for x in v]
            names = ["stage", "walk1", "resolve+rewalk", "poslist", "chunk setup", "ensure", "far req + literals", "far write", "rounds", "exact_seq", "finish"]
            tot = sum(v[:16])
            print("  seq prof, cycles per block: total %.0f | " % tot + " | ".join("%s %.0f" % (nm, v[i]) for i, nm in enumerate(names)) + " | other %.0f" % v[15], flush=True)
            print("  seq prof, per block: tiles %.2f chunks %.1f rounds %.1f (%.2f per chunk) exact %.2f sequences %.0f resolve passes %.2f far lanes %.0f near lanes %.0f" %
                  (v[16], v[17], v[18], v[18] / max(v[17], 1e-9), v[19], v[20], v[21], v[23], v[24]), flush=True)
        ts = []
        for _ in range(args.reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); dec_once(); e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        print("%-34s adversarial %s (%d cases)  bench round trip %s  ms %s" %
              (tag, "OK" if not bad else "FAIL %d" % len(bad), na, ok, " ".join("%.3f" % t for t in ts)), flush=True)
        for b in bad[:6]:
            print("    " + b, flush=True)
        lib.lz4flex_ctx_destroy(ctx)


if __name__ == "__main__":
    main()
