// lz4_fused_common.h -- the FUSED decoder (lz4_decompress_fused.hip): LDS layout, piece words, the CUTTER (the stage between the split
// decoder's parser -- lz4_split_parser.h, unchanged: one lane per block walks the token chain in the reference's check order,
// src/block/decompress.rs:244-443, and queues one 16-byte record per sequence -- and the copy engine) and the quads' step packing.
//
// Round 4 measured the two halves of the split decoder apart: its copiers cut every sequence into pieces with FOUR lanes doing the same
// arithmetic (85 wave-instructions per <= 64-byte piece, 4 330 pieces per JSON block: the copiers, not the parser, are what a block
// waits for), while the replay engine (lz4_decompress_replay.hip) runs a precompiled plan at 4 wave-instructions per step of 2.4 pieces
// -- but its planner (one lane per 64-byte PART of a block, 64 lanes in 64 places of branchy code) cost 3.0 ms.  A wavefront issues one
// instruction per four cycles whatever its lanes do, so a block's serial work has to be spread over wavefronts that each do little per
// sequence.  The pipeline per workgroup of 64 blocks, a SIMD per stage:
//   parser wavefront (one lane per block)  -> 16-byte sequence records (LDS queue, 16 per block)
//   cutter wavefront (one lane per block)  -> 4-byte PIECE words (LDS queue, 128 per block): one piece per iteration, nothing else
//   four quad wavefronts (4 lanes per block) pack the pieces that do not depend on each other into STEPS of up to four pieces / 64 bytes
//   (every lane of a quad does that arithmetic for itself: these wavefronts have the issue slots to spare), request memory sources
//   LOOKAHEAD steps ahead, and execute: 16 bytes per lane into a 1 KiB output ring per block, complete lines to memory.
// (The first version of this file packed the steps in the cutter's wavefront: 235 instructions per piece, 2.96 ms per GiB against the
// split decoder's 1.65.)
//
// Per-lane scalar code without wave-level operations: compiles for the host with -DLZ4FLEX_HOST_SIM (tests/sim/fused_model.cpp runs
// parser, cutter and a lane-exact model of the quads against the oracle; test infrastructure).
#pragma once
#include <stdint.h>

#include "lz4_split_parser.h"

namespace lz4flex_dev {
namespace fused {

using v5::u32x4;
using v5::lds_u8;
using v5::lds_vu32;
using v5::lds_vu128;

// ---- pieces ------------------------------------------------------------------------------------------------------------------------
// A PIECE is a copy of 1..64 bytes that never reads what it writes (a short-period match is a sequence of pieces with a doubling
// effective distance) and never crosses the end of the ring, source or destination.
// Piece word (u32):  [31:25] m (1..64 bytes)   [24:23] kind   [18:0] field
//   K_NEAR  source still in the block's LDS ring: field = distance back from the piece's first output byte (<= NEAR_MAX)
//   K_FAR   source has left the ring:             field = distance back (the bytes are in memory when the piece is requested)
//   K_LIT   literal bytes: field = their position in the compressed block (their 16-byte reads stay inside it: the parser's lit_slack)
//   K_END   a SPECIAL: THREE words -- this one (m = 0, field = SP_CAREFUL / SP_FINISH), the position of a literal run in the compressed
//           block, its length: the quad copies it with exact bounds (it may end at the block's last byte, it may be longer than the
//           ring); SP_FINISH ends the block (also behind an error: the parser has the status, the bytes so far are written)
constexpr uint32_t W = 1024u;            // bytes of a block's LDS output ring
constexpr uint32_t MASK = W - 1u;
constexpr uint32_t RING_PAD = 16u;       // behind the ring: a lane's 16-byte move may start at the ring's last byte
constexpr uint32_t G = 4u;               // lanes per block
constexpr uint32_t LANE_B = 16u;         // bytes a lane moves
constexpr uint32_t PIECE = G * LANE_B;   // bytes per step
constexpr uint32_t NEAR_MAX = W - 64u;   // a near source must still be in the ring when its piece executes (minus the bytes lanes write past their end)
constexpr uint32_t LOOKAHEAD = 8u;       // a memory source is requested this many steps before its piece executes
constexpr uint32_t FLUSH_EVERY = 4u;     // steps between the write-backs of a block's complete 64-byte lines
static_assert((LOOKAHEAD - 1u + FLUSH_EVERY) * PIECE + 63u + 64u <= NEAR_MAX, "far sources must be stored before they are requested");
constexpr uint32_t M_SHIFT = 25u, PK_SHIFT = 23u;
constexpr uint32_t MAX_FIELD = (1u << 19) - 1u;   // literal positions a word can name: compressed blocks below 512 KiB
constexpr uint32_t K_NEAR = 0u, K_LIT = 1u, K_FAR = 2u, K_END = 3u;
constexpr uint32_t SP_CAREFUL = 1u, SP_FINISH = 2u;
LZ4_FN uint32_t piece_word(uint32_t kind, uint32_t m, uint32_t field) { return (m << M_SHIFT) | (kind << PK_SHIFT) | field; }
LZ4_FN uint32_t pw_m(uint32_t w) { return w >> M_SHIFT; }
LZ4_FN uint32_t pw_kind(uint32_t w) { return (w >> PK_SHIFT) & 3u; }
LZ4_FN uint32_t pw_field(uint32_t w) { return w & MAX_FIELD; }

// ---- LDS of one block: output ring | piece queue | sequence queue | heads and tails | tail copy | sink | the parser's ring ----------
constexpr uint32_t PQ = 128u;            // words per piece queue (a power of two)
struct Layout {
    static constexpr uint32_t QD = 16u;                          // sequence records per queue
    static constexpr uint32_t OUT_OFF = 0u;                      // W + RING_PAD
    static constexpr uint32_t PQ_OFF = W + RING_PAD;             // PQ x 4 B
    static constexpr uint32_t Q_OFF = PQ_OFF + 4u * PQ;          // QD x 16 B
    static constexpr uint32_t CTL_OFF = Q_OFF + 16u * QD;        // sequence head, tail (the parser's QueueT), piece head, tail
    static constexpr uint32_t TAIL_OFF = CTL_OFF + 16u;
    static constexpr uint32_t SINK_OFF = TAIL_OFF + v5::TAIL_BUF;
    static constexpr uint32_t RING_OFF = (SINK_OFF + 16u + 63u) & ~63u;
    // + 16: the lanes of the parser and of the cutter (one per block) and of the quads (four per block) address LDS with this stride;
    // 2 048 bytes put all 64 of them on ONE of the 32 banks (measured on the first version: 3.4 instead of 3.0 ms), 2 064 (516 dwords
    // = 4 x 129) spreads them over all eight 16-byte bank groups
    static constexpr uint32_t BLK_LDS = ((RING_OFF + v5::RING_BYTES + 63u) & ~63u) + 16u;
    static constexpr bool RING_ALIGNED = false;
    static_assert(BLK_LDS == 2064u && RING_OFF % 64u == 0u && PQ_OFF % 16u == 0u, "2 KiB per block: 64 blocks are 129 KiB of a CU's 160");
};
constexpr uint32_t PIECE_HEAD = Layout::CTL_OFF + 8u, PIECE_TAIL = Layout::CTL_OFF + 12u;

// ---- cutter: one lane = one block ----------------------------------------------------------------------------------------------------
// Pops the parser's records and cuts them into pieces (lz4_plan_common.h's rules), ONE PIECE = ONE WORD PER ITERATION:
//   plain record (kind 0): literals and / or a match
//   R_CAREFUL / R_FINISH : a special (three words); R_FINISH ends the lane
struct Cutter {
    v5::QueueT<Layout> q;        // the parser's queue (this lane is its consumer)
    uint32_t head;               // next sequence record
    uint32_t ptail;              // words written
    uint32_t lsrc, lrem, mrem, off, per, wr;      // the record being cut
    uint32_t op;                 // output position
    uint32_t fin;
    // what the next iteration needs from LDS is read during this one (its round trip overlaps the arithmetic: an iteration is ~100
    // instructions, a round trip under load several hundred cycles): the sequence queue's tail, the piece queue's head -- both only
    // ever grow, a stale value only delays -- and the record at `head` (read BEHIND the tail: a tail that covers it proves it)
    uint32_t qt_n, ph_n;
    u32x4 e_n;
#ifdef LZ4FLEX_HOST_SIM
    uint64_t n_iter, n_pieces;
#endif

    LZ4_FN void init(lds_u8* blk) {
        q.blk = blk;
        head = 0u; ptail = 0u; lsrc = 0u; lrem = 0u; mrem = 0u; off = 0u; per = 1u; wr = 0u; op = 0u; fin = 0u;
        qt_n = 0u; ph_n = 0u; e_n = u32x4{0u, 0u, 0u, 0u};
#ifdef LZ4FLEX_HOST_SIM
        n_iter = n_pieces = 0;
#endif
    }
    static LZ4_FN uint32_t umin(uint32_t a, uint32_t b) { return a < b ? a : b; }
    LZ4_FN uint32_t piece_head() const { return *reinterpret_cast<lds_vu32*>(q.blk + PIECE_HEAD); }
    LZ4_FN void put_word(uint32_t at, uint32_t w) const { *reinterpret_cast<lds_vu32*>(q.blk + Layout::PQ_OFF + 4u * (at & (PQ - 1u))) = w; }

    // one iteration; live: this lane owns a block that is not finished.  Returns false once SP_FINISH is out.
    // (LZ4_ANY: a wave-level test on the device -- specials are rare, their three words are written under a branch of the wavefront)
    LZ4_FN bool iterate(bool live) {
        const uint32_t qt = qt_n, ph = ph_n;
        const u32x4 e = e_n;
        const bool room = (ptail - ph) <= PQ - 3u;                          // three free words
        const bool idle = (lrem | mrem) == 0u;
        const bool pop = live && room && idle && head != qt;
        const uint32_t kind = pop ? e.w >> 16 : 0u;
        const bool sp = kind >= v5::R_CAREFUL;                             // (R_RARE never arrives: the parser runs with rare_below = 0)
        lsrc = pop ? e.x : lsrc;
        lrem = (pop && !sp) ? e.y : lrem;
        mrem = pop ? e.z : mrem;
        per = pop ? (e.w & 0xFFFFu) : per;
        off = pop ? (e.w & 0xFFFFu) : off;
        wr = pop ? 0u : wr;
        head += pop ? 1u : 0u;
        q.set_head(head);
        qt_n = q.tail();
        ph_n = piece_head();
        e_n = q.get(head);
        if (LZ4_ANY(sp)) {
            if (sp) {
                const uint32_t code = kind == v5::R_FINISH ? SP_FINISH : SP_CAREFUL;
                put_word(ptail, piece_word(K_END, 0u, code));
                put_word(ptail + 1u, e.x);
                put_word(ptail + 2u, e.y);
                ptail += 3u;
                op += e.y;                                                 // the quad copies the run when it gets there
                fin = code == SP_FINISH ? 1u : fin;
            }
        }
        // ---- the piece
        const bool isl = lrem != 0u, ism = !isl && mrem != 0u;
        const bool place = live && room && (isl | ism);
        const uint32_t to_wrap = W - (op & MASK);
        const uint32_t cut = per >= PIECE ? PIECE : (per >= LANE_B ? (per & ~(LANE_B - 1u)) : per);
        const bool near = per <= NEAR_MAX;
        uint32_t m = umin(umin(isl ? lrem : mrem, isl ? PIECE : cut), to_wrap);
        m = (ism && near) ? umin(m, W - ((op - per) & MASK)) : m;
        const uint32_t word = isl ? piece_word(K_LIT, m, lsrc) : piece_word(near ? K_NEAR : K_FAR, m, per);
        {
            const uint32_t at = place ? Layout::PQ_OFF + 4u * (ptail & (PQ - 1u)) : Layout::SINK_OFF;
            *reinterpret_cast<lds_vu32*>(q.blk + at) = word;
            ptail += place ? 1u : 0u;
            *reinterpret_cast<lds_vu32*>(q.blk + PIECE_TAIL) = ptail;      // (this lane owns the word)
        }
        const uint32_t mp = place ? m : 0u;
        op += mp;
        lsrc += isl ? mp : 0u;
        lrem -= isl ? mp : 0u;
        mrem -= ism ? mp : 0u;
        wr += ism ? mp : 0u;
        // bytes [match start - offset, op) now repeat with period `off`: 2 * per reaches back to op - 2 * per, which must not lie
        // before match start - offset
        per = (ism && place && per < PIECE && wr + off >= 2u * per) ? per * 2u : per;
#ifdef LZ4FLEX_HOST_SIM
        n_iter++; n_pieces += place;
#endif
        return fin == 0u;
    }
};

// ---- quads: packing ----------------------------------------------------------------------------------------------------------------
// The next step of a block = the longest run of the pieces at the head of its queue (w0 .. w3: words the cutter has written; `avail` of
// them are valid) that fits four lanes and in which no piece reads a byte the step writes.  Every lane of the quad computes this for
// itself.  opf = the output position of the step's first byte (the front end's count).
struct LaneJob {
    uint32_t take;       // words the step consumes (0: nothing there; 3: a special)
    uint32_t special;    // 0, SP_CAREFUL, SP_FINISH (w1 = the run's position in the compressed block, w2 = its length)
    uint32_t n, kind;    // this lane: n bytes (0: rests) of kind K_NEAR / K_LIT / K_FAR ...
    uint32_t rel;        // ... to the step's first output byte + rel ...
    uint32_t src;        // ... from ring address / position in the compressed block / output position `src`
    uint32_t total;      // the step's bytes
};
LZ4_FN LaneJob pack_step(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, uint32_t avail, uint32_t opf, uint32_t g) {
    LaneJob J;
    const uint32_t k0 = pw_kind(w0);
    const bool sp = avail >= 3u && k0 == K_END;
    const bool none = avail == 0u || (k0 == K_END && avail < 3u);          // nothing, or a special whose words are not all there yet
    const uint32_t m0 = pw_m(w0), m1 = avail > 1u ? pw_m(w1) : 0u, m2 = avail > 2u ? pw_m(w2) : 0u, m3 = avail > 3u ? pw_m(w3) : 0u;
    const uint32_t c1 = (m0 + 15u) >> 4, c2 = c1 + ((m1 + 15u) >> 4), c3 = c2 + ((m2 + 15u) >> 4), c4 = c3 + ((m3 + 15u) >> 4);
    const uint32_t b1 = m0, b2 = b1 + m1, b3 = b2 + m2;
    // piece i joins if it is a piece (m != 0: a special's first word has m = 0), fits, and its source ends before the step's first byte
    // (near: distance - m >= the bytes before it in the step; far sources are a ring away)
    const bool j1 = !sp && !none && m1 != 0u && c2 <= G && (pw_kind(w1) != K_NEAR || pw_field(w1) - m1 >= b1);
    const bool j2 = j1 && m2 != 0u && c3 <= G && (pw_kind(w2) != K_NEAR || pw_field(w2) - m2 >= b2);
    const bool j3 = j2 && m3 != 0u && c4 <= G && (pw_kind(w3) != K_NEAR || pw_field(w3) - m3 >= b3);
    const uint32_t cnt = 1u + (j1 ? 1u : 0u) + (j2 ? 1u : 0u) + (j3 ? 1u : 0u);
    const uint32_t used = j3 ? c4 : (j2 ? c3 : (j1 ? c2 : c1));
    // a step that could still grow -- every piece in sight joined, lanes are left, the cutter is still writing -- waits (the quads are
    // faster than the cutter: taken as they come, every piece would be a step of its own)
    const bool grow = !sp && !none && cnt == avail && avail < 4u && used < G;
    const bool none2 = none | grow;
    J.take = none2 ? 0u : (sp ? 3u : cnt);
    J.special = sp ? (w0 & 3u) : 0u;
    // this lane's piece
    const bool a1 = j1 && g >= c1, a2 = j2 && g >= c2, a3 = j3 && g >= c3;
    const uint32_t w = a3 ? w3 : (a2 ? w2 : (a1 ? w1 : w0));
    const uint32_t cb = a3 ? c3 : (a2 ? c2 : (a1 ? c1 : 0u));
    const uint32_t bb = a3 ? b3 : (a2 ? b2 : (a1 ? b1 : 0u));
    const uint32_t j16 = (g - cb) << 4, m = pw_m(w);
    const bool rests = sp || none2 || m <= j16;
    J.n = rests ? 0u : (m - j16 < LANE_B ? m - j16 : LANE_B);
    J.kind = rests ? K_NEAR : pw_kind(w);                                   // (a resting lane may be looking at a stale word)
    J.rel = rests ? 0u : bb + j16;
    const uint32_t f = pw_field(w);
    const uint32_t back = opf + J.rel - f;                                 // a match's source: `f` bytes behind the lane's output bytes
    J.src = J.kind == K_LIT ? f + j16 : (J.kind == K_NEAR ? (back & MASK) : back);
    J.total = (sp || none2) ? 0u : (j3 ? b3 + m3 : (j2 ? b3 : (j1 ? b2 : b1)));
    return J;
}

}  // namespace fused
}  // namespace lz4flex_dev
