/*
 * parallel_chain_model.c -- scalar C model of PARALLEL TOKEN-CHAIN RECOVERY inside one LZ4 block (the
 * numbers behind it: tools/spec_parse_study.py).  TEST INFRASTRUCTURE and the specification of a kernel that does not
 * exist yet: K lanes start at K byte positions of the compressed stream (lane 0 at 0, the others wherever the stream is
 * cut), each walks the token structure of src/block/decompress.rs:244-332 from the byte it takes for a token, and marks
 * the positions it visits.  A chain that lands on a position another chain visited is identical to it from there on, so
 * the block's true chain is lane 0's chain up to the first position that the lane owning that part of the stream has
 * visited, then that lane's chain, and so on.  No lane needs the output position: lengths are summed afterwards.
 *
 * What a lane does NOT check: anything that needs the absolute output position (offset <= position, sink capacity).  A
 * block with any irregularity -- a chain that runs past the end, an offset of zero ON THE TRUE CHAIN, a last sequence that
 * does not end with the block -- is reported as irregular; a decoder built on this hands such blocks to the
 * reference-order path (as lz4_decompress_wave.hip does today).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    uint32_t ip;       /* token position */
    uint32_t lit;      /* literal length */
    uint32_t ml;       /* match length, 0 for the block's last sequence */
    uint32_t off;      /* match offset */
} pcm_seq;

/* one sequence at ip; returns the next token position, 0xFFFFFFFF if the block's last sequence ended exactly at n,
 * 0xFFFFFFFE on anything a well-formed chain cannot contain (ran past the end, offset 0) */
static uint32_t pcm_step(const uint8_t *c, uint32_t n, uint32_t ip, pcm_seq *s) {
    const uint32_t t = c[ip];
    uint32_t p = ip + 1, lit = t >> 4, ml = t & 15;
    if (lit == 15) {
        for (;;) {
            if (p >= n) return 0xFFFFFFFEu;
            const uint32_t e = c[p++];
            lit += e;
            if (e != 255) break;
        }
    }
    if (lit > n - p) return 0xFFFFFFFEu;
    p += lit;
    s->ip = ip; s->lit = lit; s->ml = 0; s->off = 0;
    if (p == n) return 0xFFFFFFFFu;                  /* :366-368 the last sequence: literals only */
    if (n - p < 2) return 0xFFFFFFFEu;
    const uint32_t off = c[p] | ((uint32_t)c[p + 1] << 8);
    p += 2;
    if (off == 0) return 0xFFFFFFFEu;
    ml += 4;
    if (ml == 19) {
        for (;;) {
            if (p >= n) return 0xFFFFFFFEu;
            const uint32_t e = c[p++];
            ml += e;
            if (e != 255) break;
        }
    }
    if (p >= n) return 0xFFFFFFFEu;                  /* :439-443 a match is never the end of a block */
    s->ml = ml; s->off = off;
    return p;
}

/* serial reference: the chain from position 0.  Returns the number of sequences, -1 if the block is irregular. */
long pcm_serial(const uint8_t *c, uint32_t n, pcm_seq *out, uint64_t *out_len) {
    long k = 0;
    uint64_t op = 0;
    if (n == 0) return -1;
    for (uint32_t ip = 0;;) {
        pcm_seq s;
        const uint32_t nx = pcm_step(c, n, ip, &s);
        if (nx == 0xFFFFFFFEu) return -1;
        out[k++] = s;
        op += (uint64_t)s.lit + s.ml;
        if (nx == 0xFFFFFFFFu) break;
        ip = nx;
    }
    *out_len = op;
    return k;
}

/* parallel recovery with K lanes; starts[j] = lane j's first byte (starts[0] == 0, increasing, < n).  Every lane parses
 * from its start until it has left its own part and landed on a position its successor's part owner visited, or until its
 * chain ends or dies (`lane_parsed[j]` = bytes it walked: the work, for the overlap statistics).  Then the true chain is
 * read off.  Returns the number of sequences (== pcm_serial's), -1 if irregular. */
long pcm_parallel(const uint8_t *c, uint32_t n, const uint32_t *starts, uint32_t K, pcm_seq *out, uint64_t *out_len,
                  uint32_t *lane_parsed) {
    if (n == 0 || K == 0 || starts[0] != 0) return -1;
    /* visited[p] = lane whose chain has a token at p (+1), 0: nobody; a lane only marks positions from its start on, and
     * only the first lane to mark a position keeps it (later ones have merged: they stop there) */
    uint8_t *owner = (uint8_t *)calloc(n + 1, 1);       /* K <= 255 */
    uint32_t *next = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(n + 1));   /* next[p]: the token after the one at p (same for every lane that visits p) */
    pcm_seq *seq_at = (pcm_seq *)malloc(sizeof(pcm_seq) * (size_t)(n + 1));
    if (!owner || !next || !seq_at || K > 255) { free(owner); free(next); free(seq_at); return -1; }
    /* lanes are independent; the order they run in here does not matter for the result (a lane that misses a merge because
     * its successor has not been there yet only walks further): run them last to first, which is the order that makes every
     * lane stop as early as a real wavefront's lanes would at best */
    for (uint32_t jj = K; jj-- > 0;) {
        const uint32_t part_end = jj + 1 < K ? starts[jj + 1] : n;
        uint32_t ip = starts[jj], walked = 0;
        for (;;) {
            if (owner[ip] != 0) break;                   /* landed on a later lane's chain: identical from here on */
            pcm_seq s;
            const uint32_t nx = pcm_step(c, n, ip, &s);
            owner[ip] = (uint8_t)(jj + 1);
            seq_at[ip] = s;
            next[ip] = nx;
            if (nx >= 0xFFFFFFFEu) { walked += n - ip < 64 ? n - ip : 64; break; }   /* end of block, or a dead chain */
            walked += nx - ip;
            ip = nx;
            (void)part_end;
        }
        if (lane_parsed) lane_parsed[jj] = walked;
    }
    /* read the true chain off: it starts at 0 (lane 0) and follows next[] -- every position on it was visited by SOME lane */
    long k = 0;
    uint64_t op = 0;
    for (uint32_t ip = 0;;) {
        if (owner[ip] == 0) { k = -1; break; }          /* cannot happen: lane 0 marks 0, and every next[] target is marked or ends */
        out[k++] = seq_at[ip];
        op += (uint64_t)seq_at[ip].lit + seq_at[ip].ml;
        const uint32_t nx = next[ip];
        if (nx == 0xFFFFFFFEu) { k = -1; break; }       /* the TRUE chain is malformed: irregular block */
        if (nx == 0xFFFFFFFFu) break;
        ip = nx;
    }
    if (k >= 0) *out_len = op;
    free(owner); free(next); free(seq_at);
    return k;
}

/* ---- the whole decoder on top of the recovered chain: output positions by a prefix sum over the sequences, all literals
 * at once, then the matches of every group of G consecutive sequences in ROUNDS -- a round executes every match whose
 * source bytes exist (written by literals, by matches of earlier rounds or groups; a match that overlaps its own output is
 * one lane's periodic copy) -- which is how a wavefront would execute them.  Returns the decoded length; -1: irregular
 * chain, -2: an offset reaches before the output, -3: the sink is too small (a decoder built on this hands all three to the
 * reference-order path, which names the exact error).  *rounds = rounds summed over the groups (the depth statistics of
 * tools/spec_parse_study.py). */
long pcm_decode(const uint8_t *c, uint32_t n, const uint32_t *starts, uint32_t K, uint32_t G, uint8_t *out, uint64_t cap,
                uint64_t *rounds) {
    pcm_seq *sq = (pcm_seq *)malloc(sizeof(pcm_seq) * ((size_t)n + 2));
    uint64_t total = 0;
    const long ns = pcm_parallel(c, n, starts, K, sq, &total, NULL);
    if (ns < 0) { free(sq); return -1; }
    if (total > cap) { free(sq); return -3; }
    uint64_t *mpos = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)ns);      /* where each sequence's match starts */
    uint8_t *have = (uint8_t *)calloc((size_t)total + 1, 1);                   /* output bytes that exist */
    uint64_t op = 0;
    for (long i = 0; i < ns; i++) {                                             /* prefix sum + literals (independent of each other) */
        const uint32_t hdr = 1u + (sq[i].lit >= 15u ? (sq[i].lit - 15u) / 255u + 1u : 0u);
        memcpy(out + op, c + sq[i].ip + hdr, sq[i].lit);
        memset(have + op, 1, sq[i].lit);
        op += sq[i].lit;
        mpos[i] = op;
        if (sq[i].ml != 0 && sq[i].off > op) { free(sq); free(mpos); free(have); return -2; }
        op += sq[i].ml;
    }
    uint64_t nr = 0;
    uint8_t *done = (uint8_t *)calloc((size_t)ns, 1), *ready = (uint8_t *)malloc((size_t)(G ? G : 1));
    for (long g0 = 0; g0 < ns; g0 += G) {
        const long g1 = g0 + (long)G < ns ? g0 + (long)G : ns;
        for (;;) {
            long left = 0, nready = 0;
            for (long i = g0; i < g1; i++) {                                    /* decisions from the state at the round's start */
                ready[i - g0] = 0;
                if (done[i] || sq[i].ml == 0) continue;
                left++;
                const uint64_t s0 = mpos[i] - sq[i].off;
                const uint64_t s1 = s0 + sq[i].ml < mpos[i] ? s0 + sq[i].ml : mpos[i];   /* the part of the source outside its own output */
                int ok = 1;
                for (uint64_t p = s0; p < s1 && ok; p++) ok = have[p];
                ready[i - g0] = (uint8_t)ok;
                nready += ok;
            }
            if (left == 0) break;
            if (nready == 0) { free(sq); free(mpos); free(have); free(done); free(ready); return -1; }   /* cannot happen: the first open match's source is complete */
            for (long i = g0; i < g1; i++) {
                if (!ready[i - g0]) continue;
                for (uint32_t k = 0; k < sq[i].ml; k++) out[mpos[i] + k] = out[mpos[i] + k - sq[i].off];
                memset(have + mpos[i], 1, sq[i].ml);
                done[i] = 1;
            }
            nr++;
        }
    }
    if (rounds) *rounds = nr;
    free(sq); free(mpos); free(have); free(done); free(ready);
    return (long)total;
}
