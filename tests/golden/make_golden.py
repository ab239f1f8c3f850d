#!/usr/bin/env python3
"""Generate tests/golden/* from the reference's own fixtures.  Run in the build container only
(/root/reference is not present on the GPU box; nothing in tests/ or bench.py reads it at run time).

The reference's text fixtures (benches/compression_{1k,34k,65k,66k_JSON}.txt, used by
tests/tests.rs:18-21 and named by BASELINE.json configs[0..1]) are stored here as LZ4 *blocks*
produced by the oracle's restatement of lz4_flex::block::compress -- so each file is at once
  (a) a golden vector for the encoder (oracle and GPU encoder must reproduce it byte for byte), and
  (b) the source of the fixture's plain bytes (decode it; md5 of the plain text is in manifest.json).
Each block is cross-checked here against C liblz4 1.9.3 (LZ4_decompress_safe), the library the
reference's own tests use as their cross-implementation anchor (tests/tests.rs:25-56).
"""
import ctypes
import hashlib
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/benches"
FIXTURES = ["compression_1k.txt", "compression_34k.txt", "compression_65k.txt", "compression_66k_JSON.txt"]


def main():
    o = ctypes.CDLL(os.path.join(ROOT, "oracle", "liblz4flex_oracle.so"))
    o.lz4o_compress_into.restype = ctypes.c_int64
    o.lz4o_compress_into.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]
    o.lz4o_get_maximum_output_size.restype = ctypes.c_size_t
    o.lz4o_get_maximum_output_size.argtypes = [ctypes.c_size_t]
    lz4 = ctypes.CDLL("liblz4.so.1")
    manifest = {}
    for name in FIXTURES:
        plain = open(os.path.join(REF, name), "rb").read()
        cap = o.lz4o_get_maximum_output_size(len(plain))
        out = ctypes.create_string_buffer(cap)
        n = o.lz4o_compress_into(plain, len(plain), out, cap)
        assert n > 0
        blk = out.raw[:n]
        back = ctypes.create_string_buffer(len(plain))
        m = lz4.LZ4_decompress_safe(blk, back, len(blk), len(plain))
        assert m == len(plain) and back.raw == plain, name
        stem = name.rsplit(".", 1)[0]
        with open(os.path.join(HERE, stem + ".lz4blk"), "wb") as f:
            f.write(blk)
        manifest[stem] = {
            "reference_file": "benches/" + name,
            "plain_len": len(plain),
            "plain_md5": hashlib.md5(plain).hexdigest(),
            "block_len": len(blk),
            "block_md5": hashlib.md5(blk).hexdigest(),
        }
    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=2, sort_keys=True)
        f.write("\n")
    print(json.dumps(manifest, indent=2))


if __name__ == "__main__":
    main()
