"""`lz4`-like command line over the frame API (reference lz4_bin/src/main.rs:34-166), block bytes from the GPU.

    python -m lz4_flex_amd.cli FILE            -> FILE.lz4          python -m lz4_flex_amd.cli FILE.lz4 -> FILE
    python -m lz4_flex_amd.cli [-d] [-o OUT]   (stdin -> stdout / OUT)
Options as the reference's: --clean (delete the original), -f/--force (overwrite), -d/--decompress, -o/--out.
"""
import argparse
import os
import sys

from .frame import FrameDecoder, FrameEncoder

LZ_EXTENSION = ".lz4"
CHUNK = 4 << 20


class _TrackWriteSize:   # main.rs:166-190
    def __init__(self, inner):
        self.inner, self.written = inner, 0

    def write(self, b):
        self.inner.write(b)
        self.written += len(b)
        return len(b)


def _copy(src, dst):     # io::copy
    n = 0
    while True:
        b = src.read(CHUNK)
        if not b:
            return n
        dst.write(b)
        n += len(b)


def handle_file(path, out, clean, force, force_decompress, print_info=True):   # main.rs:82-164
    decompress = path.endswith(LZ_EXTENSION)
    if force_decompress and not decompress:
        raise SystemExit("Can't determine an output filename")
    if out is None:
        out = path[:-len(LZ_EXTENSION)] if decompress else path + LZ_EXTENSION
        if print_info:
            print("%s filename will be: %s" % ("Decompressed" if decompress else "Compressed", out))
        if not force and os.path.exists(out):
            sys.stdout.write("%s already exists, do you want to overwrite? (y/N) " % out)
            sys.stdout.flush()
            if not sys.stdin.readline().startswith("y"):
                print("Not overwriting")
                return
    if decompress:
        with open(path, "rb") as fin, open(out, "wb") as fout:
            _copy(FrameDecoder.new(fin), fout)
    else:
        with open(path, "rb") as fin, open(out, "wb") as fout:
            tw = _TrackWriteSize(fout)
            enc = FrameEncoder.new(tw)
            n_in = _copy(fin, enc)
            enc.finish()
            if print_info:
                print("Compressed %d bytes into %d ==> %.2f%%" % (n_in, tw.written, tw.written * 100.0 / max(n_in, 1)))
    if clean:
        os.remove(path)


def main(argv=None):
    ap = argparse.ArgumentParser(prog="lz4_flex_amd.cli", description="[De]Compress data in the lz4 format.")
    ap.add_argument("--clean", action="store_true", help="delete original files (default: false)")
    ap.add_argument("-f", "--force", action="store_true", help="overwrite output files")
    ap.add_argument("-d", "--decompress", action="store_true", help="force decompress")
    ap.add_argument("input_file", nargs="?", help="file to compress/decompress ('-' or absent: stdin)")
    ap.add_argument("-o", "--out", help="output file to write to. defaults to stdout")
    o = ap.parse_args(argv)
    if o.input_file and o.input_file != "-":
        handle_file(o.input_file, o.out, o.clean, o.force, o.decompress)
        return 0
    fin = sys.stdin.buffer
    fout = open(o.out, "wb") if o.out else sys.stdout.buffer
    try:
        if o.decompress:
            _copy(FrameDecoder.new(fin), fout)
        else:
            enc = FrameEncoder.new(fout)
            _copy(fin, enc)
            enc.finish()
    finally:
        if o.out:
            fout.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
