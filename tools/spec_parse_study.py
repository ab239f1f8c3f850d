#!/usr/bin/env python3
"""CPU study for intra-block parallel DECODING (SURVEY H3 on the decode side; runs without a GPU): an LZ4 token chain
started at an arbitrary byte of a block -- the byte taken for a token -- falls into step with the true chain after a few
sequences.  For the benchmark's data (blocks from the throughput encoder's scalar model and from the oracle = the reference
encoder) this prints the distance from a random start to the first true token position the speculative chain lands on.
Once two chains share a position they are identical from there on, so lanes that start at every K-th byte of a block's
compressed stream and parse until they land on a position of their successor's chain recover the whole chain in parallel
(DESIGN.md section 9).  Uses tests/ helpers (oracle, scalar model): test infrastructure, not the product."""
import bisect, sys, random, statistics
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

def seqs(c):
    out, ip, n, op = [], 0, len(c), 0
    while ip < n:
        t = c[ip]; ip += 1
        lit = t >> 4
        if lit == 15:
            while True:
                e = c[ip]; ip += 1; lit += e
                if e != 255: break
        ip += lit
        if ip >= n:
            out.append((op, lit, 0, 0)); break
        off = c[ip] | (c[ip + 1] << 8); ip += 2
        ml = 4 + (t & 15)
        if (t & 15) == 15:
            while True:
                e = c[ip]; ip += 1; ml += e
                if e != 255: break
        out.append((op, lit, ml, off))
        op += lit + ml
    return out

# ---- second question: with the sequences known, how parallel are the COPIES?  A match can be executed once the bytes it reads
# exist; matches that read the output of earlier matches of the same group of G sequences form chains, and a group needs as
# many rounds as its longest chain ("multi-round resolution").
def depth_stats(name, c, G):
    S = seqs(c)
    depths, tot_rounds, groups = [], 0, 0
    for g0 in range(0, len(S) - 1, G):
        grp = S[g0:g0 + G]
        # match output ranges of the group: [mstart, mend)
        ms = [(op + lit, op + lit + ml) for op, lit, ml, off in grp]
        starts = [a for a, b in ms]
        level = [0] * len(grp)
        for i, (op, lit, ml, off) in enumerate(grp):
            if ml == 0: continue
            a, b = ms[i]
            s0, s1 = a - off, min(a - off + ml, a)          # source range outside its own output (self-overlap is lane-internal)
            lv = 0
            # earlier matches of the group whose output intersects [s0, s1)
            j = bisect.bisect_right(starts, s1 - 1, 0, i) - 1
            while j >= 0 and ms[j][1] > s0:
                if ms[j][0] < s1 and ms[j][1] > s0 and grp[j][2]:
                    lv = max(lv, level[j] + 1)
                j -= 1
            level[i] = lv
        d = max(level) + 1
        depths.append(d); tot_rounds += d; groups += 1
    depths.sort()
    q = lambda p: depths[min(len(depths) - 1, int(p * len(depths)))]
    print("%-22s G=%3d: rounds per group mean %.2f median %d 90%% %d max %d  -> %.1f sequences per round" %
          (name, G, tot_rounds / groups, q(.5), q(.9), depths[-1], len(S) / tot_rounds))

js = O.fixture_plain("compression_66k_JSON"); tx = O.fixture_plain("compression_65k")
log = workloads.log_stream(0, 4 << 20, device="cpu").numpy().tobytes()
for name, plain in (("log stream, one 4 MiB block", log), ("JSON tiled to 1 MiB", (js * 20)[:1 << 20]), ("text tiled to 1 MiB", (tx * 20)[:1 << 20])):
    c = W.compress(plain)
    assert O.decompress(c, len(plain)) == ("ok", plain)
    study(name + " (wave enc)", c, 1024)
    c2 = O.compress(plain)
    study(name + " (reference enc)", c2, 1024)
for name, plain in (("log 1 MiB", log[:1 << 20]), ("JSON 1 MiB", (js * 20)[:1 << 20]), ("text 1 MiB", (tx * 20)[:1 << 20])):
    c = W.compress(plain)
    for G in (16, 64, 256):
        depth_stats(name, c, G)
