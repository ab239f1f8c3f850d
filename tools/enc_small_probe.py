#!/usr/bin/env python3
"""One small input of tools/enc_variants.py through ONE variant library's scalar entry point, in a process of its own (a kernel
fault or a hang then names the input): enc_small_probe.py LIB INDEX"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    import enc_variants as E
    import oracle_api as O
    import wave_model
    lib = E.bind(sys.argv[1])
    i = int(sys.argv[2])
    d = E.small_inputs()[i]
    assert lib.lz4flex_set_tuning(None, b"compress_mode", 0) == 0
    capn = lib.lz4flex_get_maximum_output_size(len(d))
    out = C.create_string_buffer(capn)
    r = lib.lz4flex_compress_into(d, len(d), out, capn)
    ok = r >= 0 and O.decompress(out.raw[:r], len(d)) == ("ok", d)
    print("input %d (%d bytes): r=%d decode_ok=%s == model %s" % (i, len(d), r, ok, r >= 0 and out.raw[:r] == wave_model.compress(d)), flush=True)


if __name__ == "__main__":
    main()
