#!/usr/bin/env python3
"""CPU study for intra-block parallel DECODING (SURVEY H3 on the decode side; runs without a GPU): an LZ4 token chain
started at an arbitrary byte of a block -- the byte taken for a token -- falls into step with the true chain after a few
sequences.  For the benchmark's data (blocks from the throughput encoder's scalar model and from the oracle = the reference
encoder) this prints the distance from a random start to the first true token position the speculative chain lands on.
Once two chains share a position they are identical from there on, so lanes that start at every K-th byte of a block's
compressed stream and parse until they land on a position of their successor's chain recover the whole chain in parallel
(DESIGN.md section 9).  Uses tests/ helpers (oracle, scalar model): test infrastructure, not the product."""
import sys, random, statistics
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import oracle_api as O, wave_model as W
import torch
from lz4_flex_amd import workloads

def token_chain(c):
    """true token positions of a valid block"""
    pos, ip, n = [], 0, len(c)
    while ip < n:
        pos.append(ip)
        t = c[ip]; ip += 1
        lit = t >> 4
        if lit == 15:
            while True:
                e = c[ip]; ip += 1; lit += e
                if e != 255: break
        ip += lit
        if ip >= n: break
        ip += 2
        if (t & 15) == 15:
            while True:
                e = c[ip]; ip += 1
                if e != 255: break
    return pos

def spec_merge(c, s, truth, limit=1 << 20):
    """parse from byte s as if it were a token; -> bytes until the chain lands on a true token (None: ran off / gave up)"""
    ip, n = s, len(c)
    while ip < n and ip - s < limit:
        if ip in truth: return ip - s
        t = c[ip]; ip += 1
        lit = t >> 4
        if lit == 15:
            while ip < n:
                e = c[ip]; ip += 1; lit += e
                if e != 255: break
        ip += lit
        if ip + 2 > n: return None
        if c[ip] == 0 and c[ip + 1] == 0: return None      # offset 0: an impossible sequence, the chain dies (a real decoder knows it is off track)
        ip += 2
        if (t & 15) == 15:
            while ip < n:
                e = c[ip]; ip += 1
                if e != 255: break
    return None

def study(name, c, stride):
    truth = set(token_chain(c))
    rnd = random.Random(1)
    starts = [k * stride + rnd.randrange(stride) for k in range(1, len(c) // stride - 1)]
    d = [spec_merge(c, s, truth) for s in starts]
    ok = sorted(x for x in d if x is not None)
    dead = sum(x is None for x in d)
    q = lambda p: ok[min(len(ok) - 1, int(p * len(ok)))]
    print("%-28s %8d B compressed, %7d sequences (%.1f B each), %5d starts: merge after median %4d B, 90%% %5d, 99%% %6d, max %7d; dead chains %d (%.1f%%)" %
          (name, len(c), len(truth), len(c) / len(truth), len(starts), q(.5), q(.9), q(.99), ok[-1], dead, 100.0 * dead / len(starts)))

js = O.fixture_plain("compression_66k_JSON"); tx = O.fixture_plain("compression_65k")
log = workloads.log_stream(0, 4 << 20, device="cpu").numpy().tobytes()
for name, plain in (("log stream, one 4 MiB block", log), ("JSON tiled to 1 MiB", (js * 20)[:1 << 20]), ("text tiled to 1 MiB", (tx * 20)[:1 << 20])):
    c = W.compress(plain)
    assert O.decompress(c, len(plain)) == ("ok", plain)
    study(name + " (wave enc)", c, 1024)
    c2 = O.compress(plain)
    study(name + " (reference enc)", c2, 1024)
