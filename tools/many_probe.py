#!/usr/bin/env python3
"""Times lz4flex_frame_{compress,decompress}_many on device-resident streams: N streams x size, Linked (or Independent) frames of 64 KiB
blocks, under several decoder geometries (decompress_variant 0 = the library's choice, 7 / 11 / 10 = 1 024 / 512 / 256 lanes per block).
Not the reported bench (bench.py --config 5 carries the 256 x 4 MiB figure)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import oracle_api as O
    from lz4_flex_amd import _lib, frame as F, workloads
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    plain = O.fixture_plain("compression_66k_JSON")
    shapes = [(256, 4 << 20), (1024, 1 << 20), (4096, 256 << 10), (64, 16 << 20), (16, 4 << 20)]
    for mode in (1, 0):
        for n, size in shapes:
            fi = F.FrameInfo(block_size=F.BlockSize.Max64KB, block_mode=F.BlockMode(mode))
            src = workloads.json_tiles(plain, n * size, device=dev)
            cap = int(lib.lz4flex_frame_compress_bound(size, fi._c()))
            frames = torch.zeros(n * cap, dtype=torch.uint8, device=dev)
            back = torch.zeros(n * size, dtype=torch.uint8, device=dev)
            in_off, f_off = [i * size for i in range(n)], [i * cap for i in range(n)]
            tc = []
            for _ in range(4):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                flen, st = F.compress_frames_device(src, in_off, [size] * n, fi, frames, f_off, [cap] * n)
                tc.append((time.perf_counter() - t0) * 1e3)
            assert st == [0] * n
            line = "mode %d  %5d x %8d  compress %7.3f ms  ratio %.4f  decompress:" % (mode, n, size, min(tc[1:]), sum(flen) / (n * size))
            for v in (0, 7, 11, 10):
                assert lib.lz4flex_set_tuning(None, b"decompress_variant", v) == 0
                td = []
                for _ in range(4):
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                    olen, st = F.decompress_frames_device(frames, f_off, flen, back, in_off, [size] * n)
                    td.append((time.perf_counter() - t0) * 1e3)
                assert st == [0] * n and olen == [size] * n and torch.equal(back, src)
                line += "  v%d %7.3f" % (v, min(td[1:]))
            assert lib.lz4flex_set_tuning(None, b"decompress_variant", 0) == 0
            print(line, flush=True)
            del src, frames, back


if __name__ == "__main__":
    main()
