#!/usr/bin/env python3
"""Drives tools/doubling_study.cpp (CPU only): blocks written by the throughput encoder's model and by the oracle (lz4_flex's encoder
restated) for JSON / text tiles of 64 KiB and 4 MiB of log lines -> rounds, pieces and cuts of a pointer-doubling copy phase per batch
of the workgroup decoder.  Output: profiles/r06_doubling_study.txt ("first a host model ... is it really <= 8 rounds?")."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import oracle_api as O
    import wave_model as W
    from lz4_flex_amd import workloads
    exe = "/tmp/doubling_study"
    subprocess.check_call(["g++", "-O2", "-o", exe, os.path.join(ROOT, "tools", "doubling_study.cpp")])
    js, tx = O.fixture_plain("compression_66k_JSON"), O.fixture_plain("compression_65k")
    tile = lambda b, n=65536, ph=1000: (b * (n // len(b) + 2))[ph:ph + n]
    log = bytes(workloads.log_stream(0, 4 << 20, device="cpu").numpy())
    cases = [("JSON tile, 64 KiB, throughput encoder", W.compress(tile(js), sub=1)), ("JSON tile, 64 KiB, reference encoder (oracle)", O.compress(tile(js))),
             ("text tile, 64 KiB, throughput encoder", W.compress(tile(tx), sub=1)), ("text tile, 64 KiB, reference encoder (oracle)", O.compress(tile(tx))),
             ("log lines, 4 MiB, throughput encoder", W.compress(log)), ("log lines, 4 MiB, reference encoder (oracle)", O.compress(log))]
    for name, c in cases:
        with open("/tmp/doubling_blk.lz4", "wb") as f:
            f.write(c)
        print("== " + name)
        print(subprocess.run([exe, "/tmp/doubling_blk.lz4"], capture_output=True, text=True).stdout.replace("/tmp/doubling_blk.lz4: ", "  "))


if __name__ == "__main__":
    main()
