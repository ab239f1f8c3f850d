/*
 * wave_encoder_model.c -- scalar C model of the throughput ("wave") LZ4 encoder of
 * lz4_flex_amd/csrc/lz4_compress_wave.hip.  TEST INFRASTRUCTURE: the GPU tests compare the kernels'
 * bytes with this model (same parse decisions, position by position) and tools/ use it to explore ratio.
 * It is NOT a restatement of the reference encoder (that is oracle/lz4flex_block.c): the throughput mode
 * produces a valid LZ4 block that lz4_flex decodes to the input (BASELINE.json north_star), with its own
 * parse.  Block format rules it must respect: src/block/mod.rs:37-61 of the reference (MFLIMIT 12,
 * LAST_LITERALS 5, MINMATCH 4), token / length encoding src/block/compress.rs:237-247,463-486.
 *
 * Algorithm (one 64 KiB window at a time; a block longer than 64 KiB is a sequence of windows):
 *  index  : positions are visited in steps of 64; every position p <= n-12 looks its 4-byte hash up in a
 *           4096-entry table of 16-bit positions (all lookups of a step before all inserts of the step,
 *           highest position wins an insert conflict) and records d[p] = distance to that candidate.
 *  match  : a position whose distance differs from its predecessor's and whose candidate starts with the same 4
 *           bytes is a "head": its true match length is
 *           counted byte by byte (<= CAP, never across a segment end, never into the last 5 bytes), unless
 *           the position is already buried >= SKIPD bytes deep in a match found in an earlier superstep.
 *           Every position then takes the match, from any head at or before it in its segment, that
 *           reaches furthest ("best end", a prefix maximum).
 *  select : greedy with one-step lazy evaluation (a position yields to its successor if that one reaches
 *           further by more than a byte), left to right inside a segment.
 *  emit   : segments are independent parses of [s0, s1) that may reference any earlier byte of the
 *           window; the trailing literals of a segment are carried into the first sequence of the next.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define WAVE 64
#define HBITS 12
#define WINDOW 65536u
#define LONGK 84u
#define NEARP 20u
#define LONGN 8u
#define MAXHEADS 128u   /* heads per superstep: the kernel counts them in chunks of 64 (one wavefront) */
#define RUN_MIN 8192u   /* run windows: see lz4w_parse */
static int g_no_runs;   /* (tools: lz4w_set_no_runs(1) = the parse of rounds 2 - 5, without run windows) */
void lz4w_set_no_runs(int v) { g_no_runs = v; }

typedef struct {
    uint32_t nseg;   /* segments per 64 KiB window (the kernel's worker wavefronts); boundaries are multiples of 512 */
    uint32_t cap;    /* longest match a head counts */
    uint32_t skipd;  /* a position buried this deep in a running match is not evaluated */
    uint32_t hist;   /* 0, or HIST: `in` starts HIST bytes before the block (a Linked frame's previous bytes); they are history only */
    uint32_t slide;  /* windows of a block without history in front of it advance by: 0 = 64 KiB, 1 = 32 KiB (HIST: every window start of a long block sees >= 32 KiB behind it), 2 = 48 KiB (>= 16 KiB) */
    uint32_t sub;    /* 2 / 3 / 4: a block of at most 64 KiB (and more than 64 KiB / sub, without history) is cut into sub-windows: window k = [0, (k + 1) * 64 KiB / sub)
                        of the block, parsed from k * 64 KiB / sub on (the kernel's Item::sub: small batches); else 0 / 1 */
} lz4w_params;
#define HIST (WINDOW / 2u)

static inline uint32_t ld32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }

/* A block longer than 64 KiB is a sequence of windows.  Window wi covers [wi * 64 KiB, ...) -- except the LAST window of a block
 * whose length is not a multiple of 64 KiB: it is anchored at the block's END (base = n - 64 KiB), so it overlaps the window
 * before it.  Only the positions behind that window's end ("newfrom") are parsed; the overlap is history (it is indexed
 * again, and matches may start in it).  Without this a 66 675-byte block would be 65 536 + 1 139 bytes with no history for
 * the tail, and the reference's ratio pin for its JSON fixture (tests/tests.rs:168-170) fails. */
/* With history in front of the block (hist == HIST; `n` counts it) the windows advance by HIST instead of WINDOW, so every parsed
 * position has 32 to 64 KiB of the stream behind it in its window. */
static uint32_t g_slide;   /* (set by lz4w_compress from the parameters: 0, or the bytes the windows advance by -- HIST with hist != 0, else by `slide`) */
static uint32_t g_subq;    /* (... 0, or the parsed bytes per sub-window) */
static uint32_t win_count(uint32_t n, uint32_t hist) {
    (void)hist;
    if (g_subq) return (n + g_subq - 1) / g_subq;
    if (g_slide) return n <= WINDOW ? 1 : 1 + (n - WINDOW + g_slide - 1) / g_slide;
    return (n + WINDOW - 1) / WINDOW;
}
static uint32_t win_base(uint32_t n, uint32_t wi, uint32_t hist) {
    if (g_subq) return 0;
    return (wi + 1 == win_count(n, hist) && n > WINDOW) ? n - WINDOW : wi * (g_slide ? g_slide : WINDOW);
}
static uint32_t win_from(uint32_t wi, uint32_t hist) {   /* parsed from here on */
    if (g_subq) return wi * g_subq;
    return wi == 0 ? hist : (wi - 1) * (g_slide ? g_slide : WINDOW) + WINDOW;
}
static uint32_t win_end(uint32_t n, uint32_t wi, uint32_t hist) {   /* the window's end */
    if (g_subq) return wi + 1 == win_count(n, hist) ? n : (wi + 1) * g_subq;
    const uint32_t base = win_base(n, wi, hist);
    return (n - base < WINDOW) ? n : base + WINDOW;
}

/* index pass: d[p] for every p (0 = no candidate).  The table is cleared at every window start, so a candidate never lies
 * before its position's window; steps of 64 positions are counted from the window's base. */
void lz4w_index(const uint8_t *in, uint32_t n, uint16_t *d, uint32_t hist) {
    uint16_t *tab = (uint16_t *)calloc(1u << HBITS, 2);
    for (uint32_t p = 0; p < n; p++) d[p] = 0;
    for (uint32_t wi = 0; wi < win_count(n, hist); wi++) {
        const uint32_t base = win_base(n, wi, hist), newfrom = win_from(wi, hist);
        const uint32_t wend = win_end(n, wi, hist);
        memset(tab, 0, 2u << HBITS);
        for (uint32_t b = base; b < wend; b += WAVE) {
            uint32_t idx[WAVE];
            int act[WAVE];
            for (int i = 0; i < WAVE; i++) {
                const uint32_t p = b + i;
                act[i] = (p < wend && n >= 12 && p <= n - 12);
                if (!act[i]) continue;
                idx[i] = (ld32(in + p) * 2654435761u) >> (32 - HBITS);
                if (p >= newfrom) d[p] = (uint16_t)((p - base) - tab[idx[i]]);
            }
            for (int i = 0; i < WAVE; i++)
                if (act[i]) tab[idx[i]] = (uint16_t)(b + i - base);
        }
    }
    free(tab);
}

static size_t put_len(uint8_t *out, size_t o, uint32_t r) {   /* compress.rs:237-247 write_integer */
    while (r >= 255) { out[o++] = 255; r -= 255; }
    out[o++] = (uint8_t)r;
    return o;
}

typedef struct { uint32_t lit_start, lit_len, off, mlen; } lz4w_seq;

/* is p a head of segment [s0, s1) given the running best match `carry` (window-relative end << 16 | distance)? */
static int is_head(const uint8_t *in, uint32_t n, const uint16_t *d, const lz4w_params *P, uint32_t wbase, uint32_t s0, uint32_t s1,
                   uint32_t carry, uint32_t p) {
    if (p < s0 || p >= s1 || n < 12 || p > n - 12) return 0;
    const uint32_t dp = d[p], dprev = (p > s0) ? d[p - 1] : 0;
    if (dp == 0 || dp == dprev) return 0;
    if (p - wbase < dp) return 0;                                 /* candidate before the window */
    const uint32_t cend = carry >> 16;                            /* end of the running best match, window-relative */
    if (cend > (p - wbase) && cend - (p - wbase) >= P->skipd) return 0;   /* buried deep in a match already found */
    return ld32(in + p) == ld32(in + p - dp);                     /* the candidate's first 4 bytes match (p <= n - 12: readable) */
}

/* encode_seqs' first act: among the `cnt` sequences of one call (seqs[from ...]), a sequence without literals whose match has
 * its predecessor's distance continues that match -- the two are one.  The sequences behind the call move down. */
static size_t merge_batch(lz4w_seq *seqs, size_t from, size_t cnt, size_t ns) {
    size_t o = from;
    for (size_t i = from; i < ns; i++) {
        if (i > from && i < from + cnt && seqs[i].lit_len == 0 && seqs[i].off == seqs[o - 1].off) { seqs[o - 1].mlen += seqs[i].mlen; continue; }
        seqs[o++] = seqs[i];
    }
    return o;
}

/* parse of one block; returns the number of sequences (the last one has mlen == 0: final literals).
 * A segment is walked in "supersteps": 256 positions at a time when they hold at most 128 heads (the kernel
 * compacts the heads of a superstep into the 64 lanes of its wavefront, 64 heads at a time), else 128. */
size_t lz4w_parse(const uint8_t *in, uint32_t n, const uint16_t *d, const lz4w_params *P, lz4w_seq *seqs) {
    size_t ns = 0;
    const uint32_t hist = P->hist;
    uint32_t anchor = hist;
    uint32_t best[256], own[256];
    const uint32_t nwin = win_count(n, hist);
    for (uint32_t sj = 0; sj < nwin * P->nseg; sj++) {
        const uint32_t wbase = win_base(n, sj / P->nseg, hist);    /* candidates must lie in the segment's 64 KiB window */
        const uint32_t newfrom = win_from(sj / P->nseg, hist);     /* the window's positions before this are history */
        const uint32_t wj = sj % P->nseg;
        if (wj == 0 && !g_no_runs) {
            /* RUN WINDOWS (the kernel's index_window / run_geom): a window of >= RUN_MIN bytes that is one byte repeated is ONE sequence,
             * [offset 1: from its first parsed position that has a byte in front of it in the window, to its last match end], when the
             * block format allows that match; its segments are not walked */
            const uint32_t wend = win_end(n, sj / P->nseg, hist), wl = wend - wbase;
            const uint32_t skip = newfrom - wbase;
            const uint32_t ms = skip > 1 ? skip : 1;
            uint32_t me = n >= 5 + wbase ? n - 5 - wbase : 0;
            if (me > wl) me = wl;
            if (me > 65535u) me = 65535u;
            const uint32_t act_abs = n >= 12 ? n - 11 : 0;
            int run = wl >= RUN_MIN && wbase + ms < act_abs && me >= ms + 4;
            for (uint32_t p = wbase + 1; run && p < wend; p++) run = in[p] == in[wbase];
            if (run) {
                seqs[ns].lit_start = anchor; seqs[ns].lit_len = wbase + ms - anchor; seqs[ns].off = 1; seqs[ns].mlen = me - ms;
                ns++;
                anchor = wbase + me;
                sj += P->nseg - 1;
                continue;
            }
        }
        /* eleven segments (the kernel's workers since round 6) are 12 13 13 13 12 11 12 12 10 10 10 groups of 512 long, eight
         * (rounds 2 - 5) 16 17 17 17 15 16 15 15 (later segments cost more per position: the kernel's workers finish together),
         * any other count tiles the window evenly */
        static const uint32_t lo8[9] = {0, 16, 33, 50, 67, 82, 98, 113, 128};
        static const uint32_t lo11[12] = {0, 12, 25, 38, 51, 63, 74, 86, 98, 108, 118, 128};
        const uint32_t g_lo = P->nseg == 11 ? lo11[wj] : (P->nseg == 8 ? lo8[wj] : (128u * wj) / P->nseg);
        const uint32_t g_hi = P->nseg == 11 ? lo11[wj + 1] : (P->nseg == 8 ? lo8[wj + 1] : (128u * (wj + 1)) / P->nseg);
        uint32_t s0 = wbase + 512u * g_lo;
        uint32_t s1 = wbase + 512u * g_hi;
        if (g_slide || g_subq) {
            /* with history (or sliding windows, or sub-windows) the segments share the PARSED part of the window in the same proportions (clipped, half of the
             * kernel's workers would idle); starts other than the first are multiples of 512 */
            const uint32_t skip = newfrom - wbase;
            const uint32_t send = g_subq ? win_end(n, sj / P->nseg, hist) - wbase : WINDOW;      /* what the segments share ends here */
            const uint32_t span = send > skip ? send - skip : 0;
            const uint32_t g0 = (s0 - wbase) / 512u, g1 = (s1 - wbase) / 512u;
            uint32_t r0 = (skip + span * g0 / 128u) & ~511u, r1 = (skip + span * g1 / 128u) & ~511u;
            if (wj == 0 || r0 < skip) r0 = skip;
            if (r1 < skip) r1 = skip;
            if (wj + 1 == P->nseg) r1 = send;
            s0 = wbase + r0; s1 = wbase + r1;
        }
        if (s0 < newfrom) s0 = newfrom;
        if (s1 < newfrom) s1 = newfrom;
        if (s0 > n) s0 = n;
        if (s1 > n) s1 = n;
        if (s0 == s1) continue;
        uint32_t mend = (n >= 5) ? ((s1 < n - 5) ? s1 : n - 5) : 0;         /* matches end here at the latest */
        if (mend > wbase + 65535u) mend = wbase + 65535u;                    /* ends are 16-bit window-relative numbers */
        size_t batch_from = ns, npend = 0;   /* the sequences waiting in the kernel's lanes for a full wavefront (encode_seqs takes <= 64 at a time) */
        uint32_t carry = 0;    /* the match that reaches furthest so far in this segment: window-relative end << 16 | distance */
        uint32_t cursor = s0;
        uint32_t b = wbase + ((s0 - wbase) & ~255u);              /* supersteps are aligned in the window; positions before s0 are no heads */
        while (b < s1) {
            /* superstep size: the largest aligned power of two <= 256 whose positions hold <= MAXHEADS heads (128: always) */
            uint32_t size = 256, e1 = 0;
            for (;; size >>= 1) {
                if (((b - wbase) & (size - 1)) == 0) {
                    e1 = (b + size < s1) ? b + size : s1;
                    uint32_t h = 0;
                    for (uint32_t p = b; p < e1; p++) h += (uint32_t)is_head(in, n, d, P, wbase, s0, s1, carry, p);
                    if (h <= MAXHEADS || size == 64) break;
                }
            }
            const uint32_t cnt = e1 - b;
            /* heads and their own lengths */
            uint32_t klen[256];
            uint8_t is_long[256], chunk[256];
            uint32_t rank = 0;
            for (uint32_t i = 0; i < cnt; i++) {
                const uint32_t p = b + i;
                own[i] = 0; klen[i] = 0; is_long[i] = 0; chunk[i] = 0xFF;
                if (!is_head(in, n, d, P, wbase, s0, s1, carry, p)) continue;
                chunk[i] = (uint8_t)(rank++ / WAVE);                  /* heads are counted 64 at a time, in position order */
                const uint32_t dp = d[p];
                uint32_t lim = (mend > p) ? mend - p : 0;
                if (lim > P->cap) lim = P->cap;
                uint32_t k = 0;
                while (k < lim && in[p + k] == in[p - dp + k]) k++;
                klen[i] = k;
                is_long[i] = (k >= LONGK && lim > LONGK);            /* still matching at the check behind the third compare round */
            }
            /* a head that still matches after LONGK bytes and is followed within NEARP positions by another such head of its
             * chunk stops counting there (runs: every position would count to the cap); the last one of a group goes on */
            for (uint32_t c = 0; c * WAVE < rank; c++) {
                uint32_t next_long = 0xFFFFFFFFu, n_long = 0;
                for (uint32_t i = 0; i < cnt; i++) n_long += (chunk[i] == c) & is_long[i];
                for (uint32_t i = cnt; n_long >= LONGN && i-- > 0;) {     /* (a handful of long matches is ordinary data: left alone) */
                    if (!is_long[i] || chunk[i] != c) continue;
                    if (next_long != 0xFFFFFFFFu && next_long - i <= NEARP) klen[i] = LONGK;
                    next_long = i;
                }
            }
            for (uint32_t i = 0; i < cnt; i++)
                if (klen[i] >= 4) own[i] = ((b + i - wbase + klen[i]) << 16) | d[b + i];
            /* prefix maximum, carried across supersteps */
            uint32_t run = carry;
            for (uint32_t i = 0; i < cnt; i++) { if (own[i] > run) run = own[i]; best[i] = run; }
            carry = run;
            /* greedy walk with one-step lazy evaluation: a position yields to its successor if that one reaches further
             * by more than a byte (the last position of a superstep never yields: its successor is not known yet) */
            while (cursor < e1) {
                const uint32_t i = cursor - b;           /* cursor >= b: a match never ends before the superstep it was taken in... */
                const uint32_t p = cursor, rel = p - wbase, e = best[i] >> 16;
                const int can = (e >= rel + 4) && !(n < 12 || p > n - 12);
                if (!can) { cursor++; continue; }
                if (i + 1 < cnt && (best[i + 1] >> 16) > e + 1) { cursor++; continue; }
                const uint32_t len = e - rel;
                seqs[ns].lit_start = anchor; seqs[ns].lit_len = p - anchor; seqs[ns].off = best[i] & 0xFFFF; seqs[ns].mlen = len;
                ns++;
                cursor = anchor = p + len;
            }
            if (cursor < e1) cursor = e1;                /* (not reached: the loop runs until cursor >= e1) */
            b = e1;
            const size_t nsel = ns - (batch_from + npend);
            if (npend + nsel > 64) {                     /* no room in the lanes: the waiting ones are encoded, the new ones wait */
                ns = merge_batch(seqs, batch_from, npend, ns);
                batch_from = ns - nsel; npend = nsel;
            } else npend += nsel;
        }
        ns = merge_batch(seqs, batch_from, npend, ns);   /* the segment's end: the rest is encoded */
    }
    seqs[ns].lit_start = anchor; seqs[ns].lit_len = n - anchor; seqs[ns].off = 0; seqs[ns].mlen = 0;
    ns++;
    return ns;
}

/* whole encoder: returns the compressed size (out must hold get_maximum_output_size(n)) */
size_t lz4w_compress(const uint8_t *in, uint32_t n, uint8_t *out, const lz4w_params *P, uint32_t *n_seq) {
    uint16_t *d = (uint16_t *)calloc((size_t)n + WAVE, 2);
    lz4w_seq *seqs = (lz4w_seq *)malloc(sizeof(lz4w_seq) * ((size_t)n / 4 + 2));
    g_slide = P->hist != 0 ? HIST : (P->slide == 2 ? 49152u : (P->slide == 1 ? HIST : 0u));
    g_subq = 0;
    if (P->sub >= 2 && P->sub <= 4 && P->hist == 0 && n <= WINDOW) {
        const uint32_t q = ((WINDOW + P->sub - 1) / P->sub + 511u) & ~511u;   /* 32 768, 22 016, 16 384 */
        if (n > q) g_subq = q;
    }
    if (n) lz4w_index(in, n, d, P->hist);
    const size_t ns = n ? lz4w_parse(in, n, d, P, seqs) : 0;
    size_t o = 0;
    if (n == 0) { out[o++] = 0; }
    for (size_t i = 0; i < ns; i++) {
        const lz4w_seq *s = &seqs[i];
        uint8_t *tok = out + o++;
        *tok = (uint8_t)((s->lit_len >= 15 ? 15 : s->lit_len) << 4);
        if (s->lit_len >= 15) o = put_len(out, o, s->lit_len - 15);
        memcpy(out + o, in + s->lit_start, s->lit_len);
        o += s->lit_len;
        if (s->mlen == 0) break;                      /* final literals: no offset (compress.rs handle_last_literals) */
        out[o++] = (uint8_t)s->off; out[o++] = (uint8_t)(s->off >> 8);
        const uint32_t ml = s->mlen - 4;
        *tok |= (uint8_t)(ml >= 15 ? 15 : ml);
        if (ml >= 15) o = put_len(out, o, ml - 15);
    }
    if (n_seq) *n_seq = (uint32_t)ns;
    free(d); free(seqs);
    return o;
}
