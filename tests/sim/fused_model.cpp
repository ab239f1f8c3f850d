// fused_model.cpp -- HOST MODEL of the fused decoder (lz4_flex_amd/csrc/lz4_decompress_fused.hip).  Test infrastructure: compiled by
// tests/fused_model.py with g++, never shipped, never timed.
//
// One block at a time: the REAL parser (lz4_split_parser.h) and the REAL cutter and the REAL step packing (lz4_fused_common.h), host builds of the code the
// kernel runs, work on a 2 KiB "LDS" area laid out as on the device, and a lane-exact model of the block's quad executes the steps:
// four lanes, 16-byte moves whatever a piece's length, reads at once and writes in lane order, memory sources requested LOOKAHEAD
// steps before their step executes (and read THEN: a byte that is not in memory yet fails the run), a 1 KiB ring with 16 bytes of
// pad, complete lines written back every fourth step, special steps served at the end of their turn.  The three stages are stepped
// in a pseudo-random order (seed), so every queue is seen full and empty.  Guards fail the run when a lane reads outside the
// compressed block / unwritten output or writes outside the sink.
#define LZ4FLEX_HOST_SIM 1
#include <stdint.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>

#include <vector>

#include "../../lz4_flex_amd/csrc/lz4_fused_common.h"

using namespace lz4flex_dev;
using namespace lz4flex_dev::fused;

namespace {

struct Model {
    const uint8_t* in;       // the block (a private copy, poisoned around it)
    uint32_t ilen, cap;
    uint8_t* out;
    std::vector<uint8_t> lds;
    std::vector<uint8_t> written;   // 2: the byte is in memory
    struct Slot { uint32_t r[G]; uint8_t v[G][16]; };
    Slot sl[LOOKAHEAD];
    uint32_t op = 0, F = 0, pi = 0, opf = 0, hold = 0, done = 0, sp = 0, sp_src = 0, sp_len = 0;
    uint64_t turns = 0, steps = 0, rest_turns = 0;
    int err = 0;

    uint8_t* ring() { return lds.data() + Layout::OUT_OFF; }
    uint32_t ld32(uint32_t off) { uint32_t v; memcpy(&v, lds.data() + off, 4); return v; }
    void st32(uint32_t off, uint32_t v) { memcpy(lds.data() + off, &v, 4); }

    // slot job of a lane: n [31:27] | kind [26:25] | rel [24:19] | ring address [18:0]; kind K_END: a special (field = SP_*, v = position, length)
    static uint32_t job(uint32_t kind, uint32_t n, uint32_t rel, uint32_t field) { return (n << 27) | (kind << 25) | (rel << 19) | field; }
    void front(Slot& s, uint32_t pt) {
        uint32_t w[4];
        for (uint32_t k = 0; k < 4u; ++k) w[k] = ld32(Layout::PQ_OFF + 4u * ((pi + k) & (PQ - 1u)));
        const uint32_t have = pt - pi;
        const uint32_t avail = (hold | done) ? 0u : (have < 4u ? have : 4u);
        uint32_t take = 0, total = 0, special = 0;
        for (uint32_t g = 0; g < G; ++g) {
            const LaneJob J = pack_step(w[0], w[1], w[2], w[3], avail, opf, g);
            if (g == 0u) { take = J.take; total = J.total; special = J.special; }
            else if (take != J.take || total != J.total || special != J.special) { err = -30; return; }     // the quad's lanes agree
            memset(s.v[g], 0xAB, 16);
            if (special) { s.r[g] = job(K_END, 0u, 0u, special); memcpy(s.v[g], &w[1], 4); memcpy(s.v[g] + 4, &w[2], 4); continue; }
            s.r[g] = job(J.kind, J.n, J.rel, J.src & MASK);
            if (J.n == 0u) continue;
            if (J.n > 16u || J.rel + J.n > PIECE) { err = -17; return; }
            if (J.kind == K_LIT) {
                if ((uint64_t)J.src + 16u > ilen) { err = -2; return; }                       // a lane reads behind the compressed block
                memcpy(s.v[g], in + J.src, 16);
            } else if (J.kind == K_FAR) {
                if ((uint64_t)J.src + 16u > cap) { err = -3; return; }                        // ... behind the sink
                for (uint32_t k = 0; k < J.n; ++k) if (written[J.src + k] != 2) { err = -4; return; }   // a byte the lane needs is not in memory yet
                memcpy(s.v[g], out + J.src, 16);
            } else if (J.kind != K_NEAR) { err = -31; return; }
        }
        pi += take;
        opf += total + (special ? w[2] : 0u);
        if (special) hold = 1;
        fetched += take;
    }
    void back(Slot& s) {
        const bool rest = ((s.r[0] >> 25) & 3u) == K_END;
        uint8_t x[G][16];
        uint32_t total = 0;
        for (uint32_t g = 0; g < G && !rest; ++g) {
            const uint32_t r = s.r[g], n = r >> 27;
            if (n != 0u && ((r >> 19) & 63u) != total) { err = -18; return; }       // lanes in ascending, gapless order
            total += n;
            if (n == 0u) continue;
            if (((r >> 25) & 3u) == K_NEAR) {
                const uint32_t a = r & MASK;
                memcpy(x[g], ring() + a, 16);                                       // (a + 16 <= W + RING_PAD)
                for (uint32_t k = 0; k < n; ++k) if (a + k >= W) { err = -5; return; }   // the lane's own bytes never wrap
            } else {
                memcpy(x[g], s.v[g], 16);
            }
        }
        for (uint32_t g = 0; g < G && !rest; ++g) {
            const uint32_t r = s.r[g], n = r >> 27;
            if (n == 0u) continue;
            const uint32_t pos = op + ((r >> 19) & 63u), d = pos & MASK;
            if (d + n > W) { err = -19; return; }                             // the lane's own bytes never wrap
            if ((uint64_t)pos + n > cap) { err = -20; return; }               // the parser checked the sink
            memcpy(ring() + d, x[g], 16);
        }
        op += total;
        if (!rest && total) steps++;
        const uint32_t code = rest ? (s.r[0] & 3u) : 0u;
        if (code != 0u) { sp = code; memcpy(&sp_src, s.v[0], 4); memcpy(&sp_len, s.v[0] + 4, 4); }
    }
    void flush() {
        while (F + PIECE <= op) {
            if ((uint64_t)F + PIECE > cap) { err = -7; return; }
            for (uint32_t k = 0; k < PIECE; ++k) { out[F + k] = ring()[(F + k) & MASK]; written[F + k] = 2; }
            F += PIECE;
        }
    }
    void serve() {
        uint32_t src = sp_src, n = sp_len;
        if ((uint64_t)src + n > ilen || (uint64_t)op + n > cap) { if (getenv("FUSED_DEBUG")) fprintf(stderr, "serve: sp %u src %u n %u ilen %u op %u cap %u opf %u pi %u\n", sp, src, n, ilen, op, cap, opf, pi); err = -21; return; }
        while (n != 0u) {
            if (op - F >= 64u) { err = -22; return; }
            const uint32_t room = W - 128u - (op - F);
            uint32_t chunk = n < room ? n : room;
            const uint32_t to_wrap = W - (op & MASK);
            chunk = chunk < to_wrap ? chunk : to_wrap;
            for (uint32_t o = 0; o < chunk; o += 16u) {                      // (the lanes' units in ascending order: they do not overlap)
                uint8_t v[16];
                for (uint32_t k = 0; k < 16u; ++k) v[k] = (src + o + k < ilen) ? in[src + o + k] : (uint8_t)0;
                memcpy(ring() + ((op + o) & MASK), v, 16);
            }
            op += chunk; src += chunk; n -= chunk;
            flush();
            if (err) return;
        }
        if (sp == SP_FINISH) {
            flush();
            for (uint32_t k = F; k < op; ++k) { out[k] = ring()[k & MASK]; written[k] = 2; }
            F = op;
            done = 1;
        }
        sp = 0;
        hold = 0;
    }
    void turn() {
        const uint32_t pt = ld32(PIECE_TAIL);
        const uint32_t pi0 = pi;
        for (uint32_t i = 0; i < LOOKAHEAD && !err; ++i) {
            back(sl[i]);
            if (err) return;
            if (i % FLUSH_EVERY == FLUSH_EVERY - 1u) flush();
            if (err) return;
            front(sl[i], pt);
        }
        st32(PIECE_HEAD, pi);
        if (sp != 0u && !err) {
            if (op != opf - 0u && false) err = -32;
            serve();
        }
        turns++;
        rest_turns += pi == pi0;
    }
    uint64_t fetched = 0;
};

}  // namespace

// returns the status code (0 ok, 1..5 the reference's error variants) or a negative guard code; *out_len = bytes produced;
// detail[0..1] = expected, actual for OutputTooSmall; stats[0..3] = parser steps, emitter iterations, steps executed, quad turns
extern "C" int fused_model_run(const uint8_t* in, uint32_t in_len, uint8_t* out, uint32_t cap, uint32_t* out_len, uint64_t* detail,
                               uint32_t misalign, uint32_t seed, uint64_t* stats) {
    if (in_len > MAX_FIELD || cap > MAX_FIELD) return -100;              // the kernel leaves such blocks to the reference-order kernel
    Model M;
    M.lds.assign(Layout::BLK_LDS, 0);
    std::vector<uint8_t> buf(64 + 4 + (size_t)in_len + 64, 0xEE);
    uint8_t* gin = buf.data() + 64;
    gin += (4 - ((uintptr_t)gin & 3)) & 3;
    gin += misalign & 3;
    if (in_len) memcpy(gin, in, in_len);
    M.in = gin; M.ilen = in_len; M.cap = cap; M.out = out;
    M.written.assign((size_t)cap + 64u, 0);
    for (uint32_t i = 0; i < LOOKAHEAD; ++i) { for (uint32_t g = 0; g < G; ++g) M.sl[i].r[g] = 0u; memset(M.sl[i].v, 0, sizeof M.sl[i].v); }

    v5::ParserT<Layout> p;
    p.q.blk = M.lds.data();
    p.init_window(gin, in_len);
    p.rare_below = 0u;
    p.lit_slack = LANE_B - 1u;
    p.cap = cap;
    for (uint32_t i = 0; i < v5::TAIL_BUF; ++i) M.lds[Layout::TAIL_OFF + i] = (p.tstart + i < in_len) ? gin[p.tstart + i] : 0;
    p.ip = 0; p.op = 0; p.tok_over = 0; p.qtail = 0; p.status = 0; p.expected = 0; p.done = 0;
    lz4_sim_chaos_state = (misalign & 16u) ? 0x9E3779B9u ^ in_len ^ (cap << 7) : 0u;
    p.prime();
    if (in_len == 0) p.fail(LZ4FLEX_DEV_E_EXPECTED_ANOTHER_BYTE);
    Cutter em;
    em.init(M.lds.data());
    bool em_alive = true;
    uint32_t rng = seed * 2654435761u + 12345u;
    uint64_t psteps = 0, idle = 0;
    while (!M.done) {
        rng = rng * 1664525u + 1013904223u;
        const uint32_t pick = seed == 0u ? (uint32_t)(idle % 3u) * 2u + 1u : (rng >> 24) % 7u;    // seed 0: round robin (1, 3, 5); else the stages run at uneven rates
        const uint64_t before = 0;
        const uint32_t qt0 = p.qtail;
        if (pick < 2u || (seed & 1u && pick == 6u)) {
            if (!p.done) { p.step(); psteps++; }
        } else if (pick < 5u) {
            if (em_alive) em_alive = em.iterate(true);
        } else {
            M.turn();
            if (M.err) return M.err;
        }
        (void)before; (void)qt0;
        if (++idle > 400ull * ((uint64_t)in_len + 4096u)) {                    // no progress
            if (stats) { stats[0] = psteps; stats[1] = em.n_iter; stats[2] = M.steps; stats[3] = M.turns; stats[4] = em.n_pieces; stats[5] = M.rest_turns;
                         stats[6] = ((uint64_t)p.qtail << 32) | em.head; stats[7] = ((uint64_t)em.ptail << 32) | M.pi; }
            return -1003;
        }
    }
    if (em_alive) return -1006;                                                // the cutter ends with SP_FINISH
    if (!p.done) return -1004;
    *out_len = p.status == 0 ? p.op : 0u;
    if (p.status == 0 && p.op != M.op) return -1005;
    if (p.status == 0) for (uint32_t k = 0; k < p.op; ++k) if (M.written[k] != 2) return -14;
    detail[0] = p.status == LZ4FLEX_DEV_E_OUTPUT_TOO_SMALL ? p.expected : 0;
    detail[1] = p.status == LZ4FLEX_DEV_E_OUTPUT_TOO_SMALL ? cap : 0;
    if (stats) { stats[0] = psteps; stats[1] = em.n_iter; stats[2] = M.steps; stats[3] = M.turns; stats[4] = em.n_pieces; stats[5] = M.rest_turns; stats[6] = M.steps; }
    return p.status;
}
