// lz4_fused_common.h -- the FUSED decoder (lz4_decompress_fused.hip): LDS layout, step records and the EMITTER, the stage between the
// split decoder's parser (lz4_split_parser.h, unchanged: one lane per block walks the token chain in the reference's check order,
// src/block/decompress.rs:244-443, and queues one 16-byte record per sequence) and the replay decoder's copy engine
// (lz4_decompress_replay.hip: four lanes per block execute packed steps of up to four 4-byte records, 16 bytes per lane, no decisions).
//
// Round 4 measured the two halves apart: the split decoder's copiers cut every sequence into pieces with FOUR lanes doing the same
// arithmetic (85 wave-instructions per <= 64-byte piece, 4 330 pieces per JSON block: the copiers, not the parser, are what a block
// waits for), while the replay engine runs a precompiled plan at 4 wave-instructions per step of 2.4 pieces -- but its planner (one
// lane per 64-byte PART of a block, 64 lanes in 64 places of branchy code) cost 3.0 ms.  Here the planner is the emitter below: ONE
// LANE PER BLOCK in lockstep like the parser, one piece per iteration and lane, a wavefront of its own on a SIMD of its own.  The
// pipeline per workgroup of 64 blocks: parser wavefront -> 16-byte sequence records (LDS queue, 16 per block) -> emitter wavefront
// -> 16-byte steps (LDS queue, 32 per block) -> four quad wavefronts (16 blocks each) -> 1 KiB output ring per block -> memory.
//
// Per-lane scalar code without wave-level operations: compiles for the host with -DLZ4FLEX_HOST_SIM (tests/sim/fused_model.cpp runs
// parser, emitter and a lane-exact model of the quads against the oracle; test infrastructure).
#pragma once
#include <stdint.h>

#include "lz4_split_parser.h"

namespace lz4flex_dev {
namespace fused {

using v5::u32x4;
using v5::lds_u8;
using v5::lds_vu32;
using v5::lds_vu128;

// ---- steps ------------------------------------------------------------------------------------------------------------------------
// A STEP is 16 bytes: up to four PIECE words.  A piece is a copy of 1..64 bytes that never reads what it writes and never crosses the
// end of the ring (source or destination); it takes ceil(m / 16) of the quad's four lanes, the pieces of a step take the lanes in
// order, and none of them reads a byte the step writes.  The quad's lanes find their piece and their 16 bytes of it themselves
// (decode_lane below: ~25 instructions per step in wavefronts that have them to spare; the emitter, whose iteration is what a block
// waits for, writes one word per piece).
// Piece word (u32):  [31:25] m (1..64 bytes; 0: no piece)   [24:23] kind   [18:0] field
//   K_NEAR  source still in the block's LDS ring: field = ring address of the piece's first byte
//   K_LIT   literal bytes: field = their position in the compressed block
//   K_FAR   source has left the ring: field = absolute output position of the piece's first source byte
//   K_END   word 0 of a step: no copy.  field 0: nothing (padding); SP_CAREFUL / SP_FINISH: a SPECIAL step -- words 1 and 2 hold a
//           literal run's position in the compressed block and its length as plain numbers: the quad copies it with exact bounds
//           (it may end at the block's last byte, it may be longer than the ring), and SP_FINISH ends the block
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
constexpr uint32_t MAX_FIELD = (1u << 19) - 1u;   // positions a word can name: blocks (compressed and decoded) below 512 KiB
constexpr uint32_t K_NEAR = 0u, K_LIT = 1u, K_FAR = 2u, K_END = 3u;
constexpr uint32_t NOP_WORD = 0u;
constexpr uint32_t END_WORD = K_END << PK_SHIFT;
constexpr uint32_t SP_CAREFUL = 1u, SP_FINISH = 2u;
LZ4_FN uint32_t piece_word(uint32_t kind, uint32_t m, uint32_t field) { return (m << M_SHIFT) | (kind << PK_SHIFT) | field; }
LZ4_FN uint32_t pw_m(uint32_t w) { return w >> M_SHIFT; }
LZ4_FN uint32_t pw_kind(uint32_t w) { return (w >> PK_SHIFT) & 3u; }
LZ4_FN uint32_t pw_field(uint32_t w) { return w & MAX_FIELD; }
LZ4_FN bool step_rests(uint32_t w0) { return pw_kind(w0) == K_END; }       // padding or a special step
// What lane g of the quad does in the step (w0, w1, w2, w3): n bytes (0: rests) of kind `kind` from `field` (the lane's own 16 bytes:
// ring address / position in the compressed block / output position) to the step's first output byte + rel; total = the step's bytes.
struct LaneJob { uint32_t n, kind, rel, field, total; };
LZ4_FN LaneJob decode_lane(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, uint32_t g) {
    const uint32_t m0 = pw_m(w0), m1 = pw_m(w1), m2 = pw_m(w2), m3 = pw_m(w3);
    const uint32_t c1 = (m0 + 15u) >> 4, c2 = c1 + ((m1 + 15u) >> 4), c3 = c2 + ((m2 + 15u) >> 4);     // lanes before pieces 1, 2, 3
    const bool a1 = g >= c1, a2 = g >= c2, a3 = g >= c3;                                             // (a piece takes at least one lane: a3 -> a2 -> a1)
    const uint32_t w = a3 ? w3 : (a2 ? w2 : (a1 ? w1 : w0));
    const uint32_t cb = a3 ? c3 : (a2 ? c2 : (a1 ? c1 : 0u));
    const uint32_t bb = a3 ? m0 + m1 + m2 : (a2 ? m0 + m1 : (a1 ? m0 : 0u));                         // bytes before the lane's piece
    const uint32_t j16 = (g - cb) << 4, m = pw_m(w);
    LaneJob J;
    J.n = m > j16 ? (m - j16 < LANE_B ? m - j16 : LANE_B) : 0u;
    J.kind = pw_kind(w);
    J.rel = bb + j16;
    J.field = pw_field(w) + j16;
    J.total = m0 + m1 + m2 + m3;
    return J;
}

// ---- LDS of one block: output ring | step queue | sequence queue | heads and tails | tail copy | sink | the parser's ring ------------
constexpr uint32_t QS = 32u;             // steps per step queue (a power of two; the quads read LOOKAHEAD steps ahead, a turn of LOOKAHEAD at a time)
struct Layout {
    static constexpr uint32_t QD = 16u;                          // sequence records per queue
    static constexpr uint32_t OUT_OFF = 0u;                      // W + RING_PAD
    static constexpr uint32_t STEPQ_OFF = W + RING_PAD;          // QS x 16 B
    static constexpr uint32_t Q_OFF = STEPQ_OFF + 16u * QS;      // QD x 16 B
    static constexpr uint32_t CTL_OFF = Q_OFF + 16u * QD;        // sequence head, tail (the parser's QueueT), step head, tail
    static constexpr uint32_t TAIL_OFF = CTL_OFF + 16u;
    static constexpr uint32_t SINK_OFF = TAIL_OFF + v5::TAIL_BUF;
    static constexpr uint32_t RING_OFF = (SINK_OFF + 16u + 63u) & ~63u;
    // + 16: the lanes of the parser and of the emitter (one per block) and of the quads (four per block) address LDS with this stride;
    // 2 048 bytes put all 64 of them on ONE of the 32 banks (measured: the whole kernel 3.4 ms), 2 064 (516 dwords = 4 x 129) spreads
    // them over all eight 16-byte bank groups (lz4_split_parser.h Layout)
    static constexpr uint32_t BLK_LDS = ((RING_OFF + v5::RING_BYTES + 63u) & ~63u) + 16u;
    static constexpr bool RING_ALIGNED = false;
    static_assert(BLK_LDS == 2064u && RING_OFF % 64u == 0u && STEPQ_OFF % 16u == 0u, "2 KiB per block: 64 blocks are 129 KiB of a CU's 160");
};
constexpr uint32_t STEP_HEAD = Layout::CTL_OFF + 8u, STEP_TAIL = Layout::CTL_OFF + 12u;
static_assert(QS >= 3u * LOOKAHEAD, "the emitter runs ahead of the quads' front end");

// ---- emitter: one lane = one block ---------------------------------------------------------------------------------------------------
// Pops the parser's records and cuts them into pieces exactly as lz4_plan_common.h does (a piece never reads what it writes -- a
// short-period match is a sequence of pieces with a doubling effective offset --, never crosses the end of the ring, source or
// destination; pieces that do not depend on each other share a step), ONE PIECE PER ITERATION, branch-free; at most one step is
// written to the queue per iteration.
//   plain record (kind 0): literals (their 16-byte reads stay inside the block: the parser's lit_slack) and / or a match
//   R_CAREFUL / R_FINISH : the open step is closed, then a SPECIAL step, then 2 x LOOKAHEAD resting steps (the quads request memory
//                          sources LOOKAHEAD steps ahead and serve a special step at the end of the turn of LOOKAHEAD steps it lies in:
//                          nothing that follows a literal run of any length may be requested before the run is in memory); behind
//                          SP_FINISH resting steps until the quads have seen the end
struct Emitter {
    v5::QueueT<Layout> q;        // the parser's queue (this lane is its consumer)
    uint32_t head;               // next sequence record
    uint32_t stail;              // steps written
    // the record being cut
    uint32_t lsrc, lrem, mrem, off, per, wr;
    uint32_t spec, ssrc, slen;   // a special record waiting for its step
    uint32_t pad;                // resting steps still to write
    uint32_t fin;                // SP_FINISH is out: only padding from here on
    // the open step
    uint32_t op, lanes, np, start, w0, w1, w2, w3;      // output position; lanes and pieces taken; output position of the step's first byte; its words
#ifdef LZ4FLEX_HOST_SIM
    uint64_t n_iter, n_steps, n_pieces;
#endif

    LZ4_FN void init(lds_u8* blk) {
        q.blk = blk;
        head = 0u; stail = 0u; lsrc = 0u; lrem = 0u; mrem = 0u; off = 0u; per = 0u; wr = 0u; spec = 0u; ssrc = 0u; slen = 0u; pad = 0u; fin = 0u;
        op = 0u; lanes = 0u; np = 0u; start = 0u; w0 = w1 = w2 = w3 = NOP_WORD;
#ifdef LZ4FLEX_HOST_SIM
        n_iter = n_steps = n_pieces = 0;
#endif
    }
    static LZ4_FN uint32_t umin(uint32_t a, uint32_t b) { return a < b ? a : b; }
    LZ4_FN uint32_t step_head() const { return *reinterpret_cast<lds_vu32*>(q.blk + STEP_HEAD); }

    // one iteration; live: this lane owns a block.  Returns false once everything (the end's padding included) is out.
    LZ4_FN bool iterate(bool live) {
        const uint32_t qt = q.tail();
        const uint32_t sh = step_head();
        const bool room = (stail - sh) < QS;                              // a free step slot
        // ---- pop (the record's words are read whether or not they are taken: the read overlaps the arithmetic)
        const u32x4 e = q.get(head);
        const bool idle0 = (lrem | mrem | spec | pad | fin) == 0u;
        const bool pop = live && idle0 && head != qt;
        const uint32_t kind = pop ? e.w >> 16 : 0u;
        const bool sp = kind >= v5::R_CAREFUL;                            // (R_RARE never arrives: the parser runs with rare_below = 0)
        lsrc = pop ? e.x : lsrc;
        lrem = pop ? (sp ? 0u : e.y) : lrem;
        mrem = pop ? e.z : mrem;
        off = pop ? (e.w & 0xFFFFu) : off;
        per = pop ? (e.w & 0xFFFFu) : per;
        wr = pop ? 0u : wr;
        spec = sp ? (kind == v5::R_FINISH ? SP_FINISH : SP_CAREFUL) : spec;
        ssrc = sp ? e.x : ssrc;
        slen = sp ? e.y : slen;
        head += pop ? 1u : 0u;
        q.set_head(head);
        // ---- the piece this iteration would place
        const bool isl = lrem != 0u, ism = !isl && mrem != 0u, piece = isl | ism;
        const uint32_t to_wrap = W - (op & MASK);
        const uint32_t cut = per >= PIECE ? PIECE : (per >= LANE_B ? (per & ~(LANE_B - 1u)) : per);
        const bool near = per <= NEAR_MAX;
        const uint32_t msrc = op - per;
        uint32_t m = umin(umin(isl ? lrem : mrem, isl ? PIECE : cut), to_wrap);
        m = (ism && near) ? umin(m, W - (msrc & MASK)) : m;
        const uint32_t pk = isl ? K_LIT : (near ? K_NEAR : K_FAR);
        const uint32_t src = isl ? lsrc : msrc;
        const uint32_t need = (m + LANE_B - 1u) / LANE_B;
        const bool dep = ism && lanes != 0u && src + m > start;          // the match would read bytes the open step writes
        // ---- at most one step leaves per iteration: the open step (full, or in the way, or nothing else to do), a special step, padding
        const bool special_now = !piece && spec != 0u;
        const bool nothing = !piece && spec == 0u && pad == 0u && fin == 0u && head == qt;     // caught up with the parser: do not sit on an open step
        const bool close = lanes != 0u && ((piece && (lanes + need > G || dep)) || special_now || nothing);
        const bool emit_sp = special_now && lanes == 0u;
        const bool emit_pad = !piece && spec == 0u && pad != 0u;
        const bool wr_step = live && room && (close | emit_sp | emit_pad);
        const uint32_t s0 = close ? w0 : (emit_sp ? (END_WORD | spec) : (fin ? END_WORD : NOP_WORD));
        const uint32_t s1 = close ? w1 : (emit_sp ? ssrc : NOP_WORD);
        const uint32_t s2 = close ? w2 : (emit_sp ? slen : NOP_WORD);
        const uint32_t s3 = close ? w3 : NOP_WORD;
        {
            const u32x4 v = {s0, s1, s2, s3};
            const uint32_t at = wr_step ? Layout::STEPQ_OFF + 16u * (stail & (QS - 1u)) : Layout::SINK_OFF;
            *reinterpret_cast<lds_vu128*>(q.blk + at) = v;
            stail += wr_step ? 1u : 0u;
            *reinterpret_cast<lds_vu32*>(q.blk + (wr_step ? STEP_TAIL : Layout::SINK_OFF)) = stail;
        }
#ifdef LZ4FLEX_HOST_SIM
        n_iter++; n_steps += wr_step;
#endif
        const bool closed = wr_step && close;
        lanes = closed ? 0u : lanes; np = closed ? 0u : np;
        w0 = closed ? NOP_WORD : w0; w1 = closed ? NOP_WORD : w1; w2 = closed ? NOP_WORD : w2; w3 = closed ? NOP_WORD : w3;
        {   // special step out: 2 x LOOKAHEAD resting steps behind it; behind the end LOOKAHEAD (the quads fetch whole turns: the one
            // that holds SP_FINISH must be complete -- and no more than the queue takes while nobody reads it any more)
            const bool did = wr_step && emit_sp;
            pad = did ? (spec == SP_FINISH ? LOOKAHEAD : 2u * LOOKAHEAD) : (wr_step && emit_pad ? pad - 1u : pad);
            fin = (did && spec == SP_FINISH) ? 1u : fin;
            op += did ? slen : 0u;                                        // the quad copies the run when it reaches the step
            spec = did ? 0u : spec;
        }
        // ---- place the piece (if the open step takes it now: it was just closed, or there was no need to): one word
        const bool place = live && piece && (closed || !close);
        start = (place && lanes == 0u) ? op : start;
        const uint32_t word = piece_word(pk, m, pk == K_NEAR ? (src & MASK) : src);
        w0 = (place && np == 0u) ? word : w0;
        w1 = (place && np == 1u) ? word : w1;
        w2 = (place && np == 2u) ? word : w2;
        w3 = (place && np == 3u) ? word : w3;
        np += place ? 1u : 0u;
        const uint32_t mp = place ? m : 0u;
        lanes += place ? need : 0u;
        op += mp;
        lsrc += isl ? mp : 0u;
        lrem -= isl ? mp : 0u;
        mrem -= ism ? mp : 0u;
        wr += ism ? mp : 0u;
        // bytes [match start - offset, op) now repeat with period `off`: 2 * per reaches back to op - 2 * per, which must not lie
        // before match start - offset
        per = (ism && place && per < PIECE && wr + off >= 2u * per) ? per * 2u : per;
#ifdef LZ4FLEX_HOST_SIM
        n_pieces += place;
#endif
        return !(fin != 0u && pad == 0u);
    }
};

}  // namespace fused
}  // namespace lz4flex_dev
