#!/usr/bin/env python3
"""A longer random session than the test suite's (GPU box; tools, not a test): structured random inputs (fragments repeated at
random distances, runs, random bytes, text) of 0 .. 3 MiB are compressed by both encoder modes (scalar entry point), checked
with the oracle's decoder and -- the default mode -- against the encoder's scalar model, then decoded by every decoder kernel
in one batch per round (the oracle's result is the reference: bytes, length, status), also cut short and with short sinks.
usage: python tools/gpu_fuzz.py [--seconds 60] [--seed 1]"""
import argparse
import ctypes as C
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np


def make_input(rnd):
    r = rnd.random()
    n = rnd.randint(0, 200) if r < 0.15 else rnd.randint(200, 70000) if r < 0.6 else rnd.randint(70000, 400000) if r < 0.9 else rnd.randint(400000, 3 << 20)
    out = bytearray()
    frags = [bytes(rnd.getrandbits(8) for _ in range(rnd.randint(1, 64))) for _ in range(rnd.randint(1, 30))]
    alphabet = rnd.choice([256, 256, 16, 4, 2])
    while len(out) < n:
        k = rnd.random()
        if k < 0.35:
            out += rnd.choice(frags)
        elif k < 0.55 and out:
            d = rnd.randint(1, min(len(out), 70000))
            m = rnd.randint(4, 300)
            for _ in range(m):
                out.append(out[-d])
        elif k < 0.65:
            out += bytes([rnd.getrandbits(8)]) * (rnd.randint(1, 2000) if rnd.random() < 0.85 else rnd.randint(6000, 300000))     # (long runs: the encoder's run windows, the workgroup decoder's giants -- round 6)
        elif k < 0.7:
            out += np.random.default_rng(rnd.getrandbits(32)).integers(0, alphabet, rnd.randint(1, 5000), dtype=np.uint8).tobytes()
        else:
            out += bytes(rnd.getrandbits(8) % alphabet for _ in range(rnd.randint(1, 40)))
    return bytes(out[:n])


def frames(args):
    """the frame layer: random options x random write / read chunking, both compress modes, against the oracle's FrameEncoder /
    FrameDecoder restatement (exact mode: the frame's bytes; default mode: the oracle decodes it) and back"""
    import io
    import oracle_api as O
    from lz4_flex_amd import block, frame as F
    rnd = random.Random(args.seed)
    t_end = time.time() + args.seconds
    n = 0
    sizes = {4: F.BlockSize.Max64KB, 5: F.BlockSize.Max256KB, 6: F.BlockSize.Max1MB, 7: F.BlockSize.Max4MB}
    while time.time() < t_end:
        d = make_input(rnd)
        bs = rnd.choice([4, 4, 5, 6, 7])
        linked, bc, cc = rnd.random() < 0.4, rnd.random() < 0.3, rnd.random() < 0.3
        for mode in ("fast", "exact"):
            block.set_compress_mode(mode)
            fi = F.FrameInfo(block_size=sizes[bs], block_mode=F.BlockMode.Linked if linked else F.BlockMode.Independent,
                             block_checksums=bc, content_checksum=cc)
            sink = io.BytesIO()
            enc = F.FrameEncoder(sink, fi)
            pos = 0
            chunks = []
            flushed = False
            while pos < len(d):
                k = rnd.choice([1, 7, 4096, 65536, 65537, 1 << 20, len(d)])
                k = min(k, len(d) - pos)
                enc.write(d[pos:pos + k])
                chunks.append(k)
                pos += k
                if rnd.random() < 0.05:
                    enc.flush()
                    flushed = True
            enc.finish()
            fr = sink.getvalue()
            if mode == "exact" and not flushed and len(chunks) <= 4096:      # the reference's bytes, block for block
                want = O.frame_compress(d, chunks=chunks, block_mode=1 if linked else 0, block_size=bs, block_checksums=int(bc), content_checksum=int(cc))[1]
                assert fr == want, ("exact mode: frame != the oracle's FrameEncoder", bs, linked, bc, cc, len(d), len(chunks))
            rc, back, used = O.frame_decompress(fr, len(d))
            assert rc == 0 and back == d and used == len(fr), ("the oracle's FrameDecoder does not return the input", mode, bs, linked, bc, cc, len(d))
            dec = F.FrameDecoder(io.BytesIO(fr))
            got = bytearray()
            while True:
                part = dec.read(rnd.choice([1, 100, 65536, 1 << 20]))
                if not part:
                    break
                got += part
            assert bytes(got) == d, ("FrameDecoder", mode, bs, linked, bc, cc, len(d))
            assert F.decompress_frame(fr, len(d))[0] == d
            n += 1
        block.set_compress_mode("fast")
        ref = O.frame_compress(d, block_mode=1 if linked else 0, block_size=bs, block_checksums=int(bc), content_checksum=int(cc))[1]
        assert F.decompress_frame(ref, len(d))[0] == d, ("a frame written by the oracle", bs, linked, bc, cc, len(d))
    print("gpu_fuzz --frames: seed %d, %d frames written (random options, chunking, both modes), each decoded by the oracle and by both front ends; %d oracle-written frames decoded" % (args.seed, n, n // 2))


def many(args):
    """lz4flex_frame_{compress,decompress}_many: batches of random streams x random frame options x both compress modes.  Every frame
    written is decoded by the oracle's FrameDecoder (exact mode: equals the oracle's FrameEncoder); the frames of a batch -- this
    library's, the oracle's, some with a flush boundary, some cut or with a flipped bit -- go back through decompress_frames and
    every stream must come out as lz4flex_frame_decompress returns it alone (bytes, or the same error class)."""
    import io
    import oracle_api as O
    from lz4_flex_amd import block, frame as F
    rnd = random.Random(args.seed)
    t_end = time.time() + args.seconds
    sizes = {0: F.BlockSize.Auto, 4: F.BlockSize.Max64KB, 5: F.BlockSize.Max256KB, 6: F.BlockSize.Max1MB, 7: F.BlockSize.Max4MB}
    batches = written = read = 0
    while time.time() < t_end:
        streams = [make_input(rnd) for _ in range(rnd.randint(1, 40))]
        bs = rnd.choice([0, 4, 4, 4, 5, 6, 7])
        linked, bc, cc, cs = rnd.random() < 0.6, rnd.random() < 0.25, rnd.random() < 0.25, rnd.random() < 0.25
        fi = F.FrameInfo(block_size=sizes[bs], block_mode=F.BlockMode.Linked if linked else F.BlockMode.Independent,
                         block_checksums=bc, content_checksum=cc, content_size=0 if cs else None)
        frames_in = []
        for mode in ("fast", "exact"):
            block.set_compress_mode(mode)
            try:
                frs = F.compress_frames(streams, fi)
            finally:
                block.set_compress_mode("fast")
            for d, fr in zip(streams, frs):
                rc, back, used = O.frame_decompress(fr, len(d))
                assert rc == 0 and back == d and used == len(fr), ("many: the oracle's FrameDecoder does not return the stream", mode, bs, linked, bc, cc, cs, len(d))
                if mode == "exact":
                    want = O.frame_compress(d, block_mode=1 if linked else 0, block_size=bs, block_checksums=int(bc), content_checksum=int(cc),
                                            content_size=len(d) if cs else None)[1]
                    assert fr == want, ("many, exact mode: frame != the oracle's FrameEncoder", bs, linked, bc, cc, cs, len(d))
                written += 1
            frames_in.append(frs)
        # the way back: a mix of frames per stream
        mix, caps = [], []
        for i, d in enumerate(streams):
            k = rnd.random()
            if k < 0.3:
                fr = frames_in[0][i]
            elif k < 0.5:
                fr = frames_in[1][i]
            elif k < 0.7:
                fr = O.frame_compress(d, block_mode=rnd.randint(0, 1), block_size=rnd.choice([4, 5, 6, 7]), block_checksums=rnd.randint(0, 1),
                                      content_checksum=rnd.randint(0, 1))[1]
            elif k < 0.8 and len(d) > 2:
                sink = io.BytesIO()
                enc = F.FrameEncoder(sink, F.FrameInfo(block_size=F.BlockSize.Max64KB, block_mode=F.BlockMode.Linked))
                cut = rnd.randint(1, len(d) - 1)
                enc.write(d[:cut]); enc.flush(); enc.write(d[cut:]); enc.finish()
                fr = sink.getvalue()
            elif k < 0.9:
                fr = bytearray(frames_in[0][i])
                fr[rnd.randrange(len(fr))] ^= 1 << rnd.randrange(8)
                fr = bytes(fr)
            else:
                fr = frames_in[1][i][:rnd.randint(0, len(frames_in[1][i]))]
            mix.append(fr)
            caps.append(len(d) + rnd.choice([0, 0, 0, 1, 70000]))
        # (every other batch with "decompress_level_chains" 1: the Linked streams of the call are decoded a level per launch -- the path that
        # calls of >= 1 024 streams take; block sizes above 64 KiB then go through the sequence decoder's / the workgroup decoder's prefix mode)
        from lz4_flex_amd import _lib as L_
        level = batches % 2 == 1
        assert L_.load().lz4flex_set_tuning(None, b"decompress_level_chains", 1 if level else 1024) == 0
        try:
            got = F.decompress_frames(mix, caps, return_errors=True)
        finally:
            assert L_.load().lz4flex_set_tuning(None, b"decompress_level_chains", 1024) == 0
        for i, (fr, cap) in enumerate(zip(mix, caps)):
            try:
                alone = F.decompress_frame(fr, cap)[0]
            except Exception as e:
                alone = e
            if isinstance(alone, Exception):
                assert type(got[i]) is type(alone), ("many: error class", i, repr(got[i])[:80], repr(alone)[:80])
            else:
                assert got[i] == alone, ("many: bytes", i, len(fr), cap)
            read += 1
        batches += 1
    print("gpu_fuzz --many: seed %d, %d batches, %d frames written by compress_frames (random options, both modes) and decoded by the oracle, "
          "%d frames (ours, the oracle's, flushed, corrupted, cut) through decompress_frames == lz4flex_frame_decompress alone" % (args.seed, batches, written, read))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--frames", action="store_true")
    ap.add_argument("--many", action="store_true")
    args = ap.parse_args()
    if args.frames:
        return frames(args)
    if args.many:
        return many(args)
    import oracle_api as O
    import wave_model as W
    from lz4_flex_amd import _lib as L, block
    lib = L.load()
    rnd = random.Random(args.seed)
    # (decompress_variant, blocks per workgroup of the split decoder | for 7 / 8: 2 = a parser and a copier workgroup per block, in
    # launches of 100 blocks)
    # every decoder configuration the library names (lz4flex_get_tuning "decoder_config_<i>": the same list the test matrix is made of),
    # plus the workgroup decoder with two workgroups per block: (variant, blocks per workgroup or pair mode)
    decoders, i = [], 0
    while True:
        v = lib.lz4flex_get_tuning(None, b"decoder_config_%d" % i)
        if v < 0:
            break
        decoders.append((v // 1000, v % 1000 if v // 1000 == 4 else 0))
        i += 1
    decoders += [(7, 2), (8, 2)]
    t_end = time.time() + args.seconds
    rounds = inputs = blocks = 0
    while time.time() < t_end:
        rounds += 1
        cases = []
        for _ in range(24):
            d = make_input(rnd)
            inputs += 1
            for mode in (0, 1):
                assert lib.lz4flex_set_tuning(None, b"compress_mode", mode) == 0
                c = block.compress(d) if hasattr(block, "compress") else None
                if c is None:
                    cap = int(lib.lz4flex_get_maximum_output_size(len(d)))
                    out = C.create_string_buffer(cap)
                    r = lib.lz4flex_compress_into(d, len(d), out, cap)
                    assert r >= 0, (r, L.last_error())
                    c = out.raw[:r]
                assert O.decompress(c, len(d)) == ("ok", d), ("encoder mode %d: the oracle does not return the input" % mode, len(d))
                if mode == 0 and c != W.compress(d):
                    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                    with open(os.path.join(ROOT, "gpurun_out", "fuzz_fail_input.bin"), "wb") as f:
                        f.write(d)
                    with open(os.path.join(ROOT, "gpurun_out", "fuzz_fail_gpu.lz4"), "wb") as f:
                        f.write(c)
                    raise AssertionError(("default encoder != its model (input and GPU output written to gpurun_out/fuzz_fail_*)", len(d)))
                if mode == 1:
                    assert c == O.compress(d), ("exact encoder != the reference's bytes", len(d))
                cases.append((c, len(d)))
                if len(c) > 4:
                    cases.append((c[:rnd.randint(1, len(c) - 1)], len(d)))
                    cases.append((c, max(0, len(d) - rnd.randint(1, 40))))
        assert lib.lz4flex_set_tuning(None, b"compress_mode", 0) == 0
        want = [O.decompress(c, k) for c, k in cases]
        inb = np.frombuffer(b"".join(c for c, _ in cases) + bytes(64), dtype=np.uint8)
        in_off = np.cumsum([0] + [len(c) for c, _ in cases[:-1]])
        out_off = np.cumsum([0] + [k + 64 for _, k in cases[:-1]])
        caps = [k for _, k in cases]
        for variant, bpw in decoders:
            out = np.full(int(out_off[-1]) + caps[-1] + 64, 0xA5, dtype=np.uint8)
            ctx = C.c_void_p()
            assert lib.lz4flex_ctx_create(C.byref(ctx), -1) == 0
            assert lib.lz4flex_set_tuning(ctx, b"decompress_variant", variant) == 0
            lens = [len(c) for c, _ in cases]
            if variant >= 7 and bpw == 2:
                assert lib.lz4flex_set_tuning(ctx, b"decompress_pcd_pair", 2) == 0
                ol, st = np.zeros(len(cases), dtype=np.uint32), np.zeros(len(cases), dtype=np.int32)
                for at in range(0, len(cases), 100):
                    e = min(at + 100, len(cases))
                    o2, s2, _d2 = block.decompress_batch(inb, list(in_off[at:e]), lens[at:e], out, list(out_off[at:e]), caps[at:e], ctx=ctx)
                    ol[at:e], st[at:e] = o2, s2
            else:
                if bpw:
                    assert lib.lz4flex_set_tuning(ctx, b"decompress_blocks_per_wg", bpw) == 0
                ol, st, det = block.decompress_batch(inb, list(in_off), lens, out, list(out_off), caps, ctx=ctx)
            lib.lz4flex_ctx_destroy(ctx)
            for i, ((c, k), w) in enumerate(zip(cases, want)):
                o = int(out_off[i])
                if w[0] == "ok":
                    assert st[i] == 0 and ol[i] == len(w[1]) and out[o:o + len(w[1])].tobytes() == w[1], ("decoder", variant, bpw, i, len(c), k, int(st[i]), int(ol[i]))
                else:
                    assert O.ERR_NAMES.get(int(st[i])) == w[0], ("decoder", variant, bpw, i, len(c), k, int(st[i]), w[0])
                assert out[o + k:o + k + 64].tobytes() == b"\xA5" * 64, ("decoder wrote behind a sink", variant, bpw, i)
            blocks += len(cases)
    print("gpu_fuzz: seed %d, %d rounds, %d inputs through both encoders, %d block decodes through %d decoder configurations (every kernel, the workgroup decoder in four geometries and with two workgroups per block): all equal the oracle" % (args.seed, rounds, inputs, blocks, len(decoders)))


if __name__ == "__main__":
    main()
