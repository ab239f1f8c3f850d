"""Host study for the lane-per-sequence chunk decoder (round 6): how many dependency rounds does a chunk of 64
consecutive sequences need, how many matches are far for a given LDS ring, which length classes occur.
Test infrastructure (uses the oracle and the encoder model); python tools/chunk_study.py"""
import os
import sys
import collections

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_api as O      # noqa: E402
import wave_model as W      # noqa: E402
import json, hashlib        # noqa: E402


def fixture(stem):
    g = os.path.join(ROOT, "tests", "golden")
    m = json.load(open(os.path.join(g, "manifest.json")))[stem]
    blk = open(os.path.join(g, stem + ".lz4blk"), "rb").read()
    st, plain = O.decompress(blk, m["plain_len"])
    assert st == "ok" and hashlib.md5(plain).hexdigest() == m["plain_md5"]
    return plain


def parse(blk):
    """-> list of (tokpos, lit, off, ml) ; ml = 0 for the last sequence"""
    p, n, out = 0, len(blk), []
    while True:
        tp = p
        t = blk[p]; p += 1
        lit = t >> 4
        if lit == 15:
            while True:
                b = blk[p]; p += 1; lit += b
                if b != 255: break
        p += lit
        if p >= n:
            out.append((tp, lit, 0, 0)); break
        off = blk[p] | (blk[p + 1] << 8); p += 2
        ml = 4 + (t & 15)
        if ml == 19:
            while True:
                b = blk[p]; p += 1; ml += b
                if b != 255: break
        out.append((tp, lit, off, ml))
    return out


def study(name, blocks, C=64, rings=(2048, 4096, 8192, 16384)):
    tot_seq = tot_chunks = 0
    rounds_prefix = collections.Counter()
    rounds_exact = collections.Counter()
    rounds_relink = collections.Counter(); rounds_relink_any = collections.Counter(); relink_far = [0]; relink_far_any = [0]
    far = {r: 0 for r in rings}
    nmatch = 0
    litc = collections.Counter(); mlc = collections.Counter()
    periodic = 0
    chunk_out_max = 0
    chunk_out_sum = 0
    for blk in blocks:
        seqs = parse(blk)
        tot_seq += len(seqs)
        op = 0
        pos = []
        for (tp, lit, off, ml) in seqs:
            pos.append(op); op += lit + ml
        for c0 in range(0, len(seqs), C):
            ch = seqs[c0:c0 + C]
            base = pos[c0]
            end = pos[c0 + len(ch) - 1] + ch[-1][1] + ch[-1][3]
            chunk_out_max = max(chunk_out_max, end - base)
            chunk_out_sum += end - base
            tot_chunks += 1
            # in-order prefix rule: lane i ready when every sequence that starts before its source end is done
            starts = [pos[c0 + i] for i in range(len(ch))]
            need = []     # number of leading sequences of the chunk that must be done
            for i, (tp, lit, off, ml) in enumerate(ch):
                if ml == 0:
                    need.append(0); continue
                d = starts[i] + lit
                s_end = min(d - off + ml, d)       # periodic: reads up to its own start
                # first sequence j whose start >= s_end  -> need j sequences done (0 if s_end <= base)
                j = 0
                while j < i and starts[j] < s_end: j += 1
                # own literals count as done (written in the literal phase): if s_end > starts[i] the source includes own literals only
                need.append(j)
                nmatch += 1
                if off < ml: periodic += 1
                for r in rings:
                    # far: source starts before what the ring is guaranteed to hold (ring - chunk's own output so far)
                    if off > r - 1024: far[r] += 1
            done = [ml == 0 for (_, _, _, ml) in ch]
            r = 0
            while not all(done):
                dp = 0
                while dp < len(ch) and done[dp]: dp += 1
                nd = list(done)
                for i in range(len(ch)):
                    if not done[i] and need[i] <= dp: nd[i] = True
                    elif not done[i] and need[i] == i and all(done[:i]): nd[i] = True
                done = nd; r += 1
            rounds_prefix[r] += 1
            # exact rule: ready when all sequences overlapping the source are done
            done = [ml == 0 for (_, _, _, ml) in ch]
            r = 0
            while not all(done):
                nd = list(done)
                for i, (tp, lit, off, ml) in enumerate(ch):
                    if done[i]: continue
                    d = starts[i] + lit
                    s0, s1 = d - off, min(d - off + ml, d)
                    ok = True
                    for j in range(i):
                        mj0 = starts[j] + ch[j][1]; mj1 = mj0 + ch[j][3]
                        if not done[j] and mj0 < s1 and mj1 > s0: ok = False; break
                    if ok: nd[i] = True
                done = nd; r += 1
            rounds_exact[r] += 1
            # relinking (second session): a match whose whole source lies inside the MATCH of one earlier sequence j of the chunk reads j's
            # source instead (out[x] = out[x - off_j] over j's match), again and again while that holds and the new source stays within
            # KEEP_BACK bytes in front of the chunk (what a slide of the window keeps); then the prefix rule on the relinked sources
            for keep_back, ctr, farctr in ((1280, rounds_relink, relink_far), (1 << 30, rounds_relink_any, relink_far_any)):
                srcs = []
                for i, (tp, lit, off, ml) in enumerate(ch):
                    d = starts[i] + lit
                    s0 = d - off
                    if ml == 0 or off < ml:
                        srcs.append((s0, min(s0 + ml, d))); continue
                    for _ in range(16):
                        hit = None
                        for j in range(i):
                            mj0 = starts[j] + ch[j][1]; mj1 = mj0 + ch[j][3]
                            if ch[j][3] and ch[j][2] >= ch[j][3] and mj0 <= s0 and s0 + ml <= mj1:
                                hit = j; break
                        if hit is None or s0 - ch[hit][2] < base - keep_back: break
                        s0 -= ch[hit][2]
                    if s0 < d - off and s0 < base - 1280: farctr[0] += 1
                    srcs.append((s0, s0 + ml))
                done = [ml == 0 for (_, _, _, ml) in ch]
                r = 0
                while not all(done):
                    dp = 0
                    while dp < len(ch) and done[dp]: dp += 1
                    S = starts[dp]
                    nd = list(done)
                    for i in range(len(ch)):
                        if not done[i] and (srcs[i][1] <= S or i == dp): nd[i] = True
                    done = nd; r += 1
                ctr[r] += 1
        for (tp, lit, off, ml) in seqs:
            litc[0 if lit == 0 else 1 if lit <= 4 else 2 if lit <= 16 else 3 if lit <= 32 else 4] += 1
            if ml: mlc[0 if ml <= 8 else 1 if ml <= 16 else 2 if ml <= 32 else 3 if ml <= 64 else 4] += 1
    def mean(c): return sum(k * v for k, v in c.items()) / max(1, sum(c.values()))
    print("== %s: %d blocks, %.0f seq/block, %d chunks, chunk out mean %.0f max %d" % (name, len(blocks), tot_seq / len(blocks), tot_chunks, chunk_out_sum / tot_chunks, chunk_out_max))
    print("   rounds (prefix rule) mean %.2f  hist %s" % (mean(rounds_prefix), sorted(rounds_prefix.items())[:14]))
    print("   rounds (exact rule)  mean %.2f  hist %s" % (mean(rounds_exact), sorted(rounds_exact.items())[:14]))
    print("   rounds (prefix rule, sources relinked while they stay within 1 280 bytes in front of the chunk) mean %.2f ; relinked anywhere: mean %.2f (%.1f %% of the matches then lie further back)" %
          (mean(rounds_relink), mean(rounds_relink_any), 100.0 * relink_far_any[0] / max(nmatch, 1)))
    print("   far share by ring: %s ; periodic %.3f%%" % ({r: round(far[r] / nmatch, 3) for r in rings}, 100.0 * periodic / nmatch))
    n = sum(litc.values())
    print("   lit classes 0 / 1-4 / 5-16 / 17-32 / >32: %s" % [round(litc[k] / n, 3) for k in range(5)])
    n = sum(mlc.values())
    print("   ml classes <=8 / <=16 / <=32 / <=64 / >64: %s" % [round(mlc[k] / n, 3) for k in range(5)])


if __name__ == "__main__":
    js = fixture("compression_66k_JSON")
    tx = fixture("compression_65k")
    nb = 6
    def tiles(plain, phase0):
        out = []
        for b in range(nb):
            ph = (phase0 + b * 65536) % len(plain)
            rep = plain * 3
            out.append(rep[ph:ph + 65536])
        return out
    for nm, plain in (("json", js), ("text", tx)):
        t = tiles(plain, 0)
        study(nm + " / oracle parse", [O.compress(x) for x in t])
        study(nm + " / wave-encoder parse", [W.compress(x, sub=1) for x in t])
    import torch
    from lz4_flex_amd import workloads
    lg = workloads.log_stream(0, 4 * 65536).numpy().tobytes()
    study("log / wave-encoder parse", [W.compress(lg[i * 65536:(i + 1) * 65536], sub=1) for i in range(4)])
