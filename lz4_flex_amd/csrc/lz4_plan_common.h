// lz4_plan_common.h -- the COPY PLAN of the plan / replay decoder (lz4_decompress_plan.hip, lz4_decompress_replay.hip):
// record format, geometry, and the per-lane code that turns one LZ4 sequence into plan records.  Per-lane scalar code
// without wave-level operations: the same source compiles for the host with -DLZ4FLEX_HOST_SIM, where tests/sim/plan_model.cpp
// (test infrastructure) compiles blocks into plans, replays them byte by byte the way the kernel's lanes do, and compares
// with the oracle.
//
// Why two kernels.  A block's decode (src/block/decompress.rs:244-443) is two serial chains: the token chain (where does
// the next sequence start) and the copy chain (a match may read what the previous sequence wrote).  The first one can be
// cut -- a chain started at a wrong byte falls into step with the true one after a few sequences -- the second one cannot
// (a part of a block that starts decoding before its predecessors are done finds most of its sources unwritten).  So the
// PLAN kernel does everything that can be done in parallel (token chain in parts, lengths, bounds checks, output
// positions by prefix sums, cutting every copy into pieces a lane group can execute without a decision, packing pieces that
// do not depend on each other into one step) and leaves a flat array of 4-byte records; the REPLAY kernel walks that array,
// four lanes per block, and does nothing else.  The serial part of a block is then as short as it can be made.
//
// A STEP of the replay is four records, one per lane of the block's lane group; a lane moves at most 16 bytes:
// Record (u32):  [31:27] n (0..16 bytes; 0: this lane rests)   [26:25] kind   [24:19] where the bytes go, relative to the
//                step's first output byte (0..63)   [18:0] field
//   K_NEAR  source still in the block's LDS ring: field = ring address of the lane's bytes (11 bits)
//   K_LIT   literal bytes: field = their position in the compressed block
//   K_FAR   source has left the ring: field = absolute output position of the lane's bytes
//   K_END   end of the plan, in all four lanes (the padding behind it is K_END too)
// A copy of up to 64 bytes (a PIECE: never reads what it writes -- n <= effective offset, a short-period match is a
// sequence of pieces with a doubling effective offset --, never crosses the end of the LDS ring, source or destination) takes
// ceil(n / 16) consecutive lanes.  Short pieces SHARE a step (JSON: 2.4 pieces per step, text: 3.8) as long as none of them
// reads a byte the step writes: the lanes read their sources at once and write in lane order, each lane 16 bytes whatever
// its n, so that a lane's surplus bytes are overwritten by the next lane (and the last lane's by the next step).  A literal
// lane is only in the plan if its 16-byte read stays inside the compressed block; the pieces behind the first one that does
// not (the last few bytes of a block) form the TAIL, executed byte by byte at the end.  (Memory only sees whole 64-byte lines
// of final bytes, written from the ring, and the bytes behind the last full line byte by byte.)
#pragma once
#include <stdint.h>

#ifdef LZ4FLEX_HOST_SIM
#define PLAN_FN static inline
#define PLAN_MEM inline
#else
#define PLAN_FN __device__ __forceinline__
#define PLAN_MEM __device__ __forceinline__
#endif

namespace lz4flex_dev {
namespace plan {

constexpr uint32_t W = 2048u;            // bytes of a block's LDS output ring (64 blocks per CU: 128 KiB)
constexpr uint32_t MASK = W - 1u;
constexpr uint32_t RING_PAD = 16u;       // behind the ring: a lane's 16-byte move may start at the ring's last byte
constexpr uint32_t RING_STRIDE = W + RING_PAD;
constexpr uint32_t G = 4u;               // lanes per block
constexpr uint32_t LANE_B = 16u;         // bytes a lane moves
constexpr uint32_t PIECE = G * LANE_B;   // bytes per step
// a near source must still be in the ring when the piece executes: the ring holds the last W bytes, minus the 15 bytes the
// previous step's last lane may have written past its end, minus slack
constexpr uint32_t NEAR_MAX = W - 64u;
// a far source is read LOOKAHEAD steps before the piece executes: it must have been stored by then.  Worst case the steps
// in between write LOOKAHEAD * PIECE bytes.  LOOKAHEAD is as deep as that allows: the loads of a wavefront return in order, so
// every load has to be LOOKAHEAD steps' worth of time away from its use or the slowest one (a far source that left the L2)
// sets the pace of all of them
constexpr uint32_t LOOKAHEAD = 24u;
constexpr uint32_t FLUSH_EVERY = 4u;     // steps between the write-backs of a block's complete 64-byte lines
// (+ what FLUSH_EVERY steps may leave in the ring + the line that is not complete yet + the 16-byte granules of the read)
static_assert((LOOKAHEAD - 1u + FLUSH_EVERY) * PIECE + 63u + 64u <= NEAR_MAX, "far sources must be stored before they are requested");
constexpr uint32_t TURN_STEPS = LOOKAHEAD;               // steps per turn of the replay kernel's loop
constexpr uint32_t TURN_WORDS = G * TURN_STEPS;          // four steps are 64 bytes, 16 per lane: step s, lane g -> word 16 (s / 4) + 4 g + s % 4
PLAN_FN uint32_t word_of(uint32_t step, uint32_t g) { return 16u * (step / 4u) + 4u * g + step % 4u; }
constexpr uint32_t END_TURNS = 3u;       // turns of K_END behind the turn with the last step (the replay kernel fetches two turns ahead)
constexpr uint32_t N_SHIFT = 27u, KIND_SHIFT = 25u, REL_SHIFT = 19u;
constexpr uint32_t KIND_MASK = 3u << KIND_SHIFT;
constexpr uint32_t MAX_FIELD = (1u << REL_SHIFT) - 1u;   // positions a record can name: blocks (compressed and decoded) below 512 KiB

constexpr uint32_t K_NEAR = 0u, K_LIT = 1u, K_FAR = 2u, K_END = 3u;
PLAN_FN uint32_t rec(uint32_t kind, uint32_t n, uint32_t rel, uint32_t field) { return (n << N_SHIFT) | (kind << KIND_SHIFT) | (rel << REL_SHIFT) | field; }
PLAN_FN uint32_t rec_kind(uint32_t r) { return (r >> KIND_SHIFT) & 3u; }
PLAN_FN uint32_t rec_n(uint32_t r) { return r >> N_SHIFT; }
PLAN_FN uint32_t rec_rel(uint32_t r) { return (r >> REL_SHIFT) & 63u; }
PLAN_FN uint32_t rec_field(uint32_t r) { return r & MAX_FIELD; }
constexpr uint32_t END_REC = (K_END << KIND_SHIFT);   // n = 0
constexpr uint32_t NOP_REC = 0u;                      // a resting lane
// tail records (executed one after the other, byte by byte, in global memory): K_LIT with field = position in the compressed
// block, K_FAR with field = the match OFFSET; n <= 16

// Per-block header the plan kernel leaves for the replay kernel (32 bytes).
struct BlockPlan {
    uint64_t in_off;        // of the compressed block in the batch's input buffer
    uint64_t out_off;       // of the block's sink in the batch's output buffer
    uint32_t first_word;    // index of the plan's first turn in the plan array (a multiple of TURN_WORDS)
    uint32_t tail_word;     // index of the first tail record
    uint32_t tail_op;       // output position the tail starts at
    uint16_t n_tail;        // tail records
    uint16_t flags;         // 0 = replay this block; else skip it (irregular block: left to the reference-order kernel)
};
static_assert(sizeof(BlockPlan) == 32, "BlockPlan");
constexpr uint32_t MAX_TAIL = 0xFFFFu;

// running state of a block's emission
struct Emit {
    uint32_t op;        // output position
    uint32_t in_len;    // compressed length
    uint32_t tail;      // 1 once a lane's read would leave the compressed block: everything behind it goes to the tail
    // the step being filled
    uint32_t lanes;     // lanes taken
    uint32_t bytes;     // bytes they move
    uint32_t start;     // output position of the step's first byte
    uint32_t w0, w1, w2, w3;    // (four names, not an array: an array indexed by `lanes` lives in scratch memory on the device)
};
static_assert(G == 4u, "Emit");
PLAN_FN void emit_init(Emit& e, uint32_t in_len) {
    e.op = 0u; e.in_len = in_len; e.tail = 0u; e.lanes = 0u; e.bytes = 0u; e.start = 0u;
    e.w0 = e.w1 = e.w2 = e.w3 = NOP_REC;
}
PLAN_FN void emit_set(Emit& e, uint32_t lane, uint32_t r) {
    e.w0 = lane == 0u ? r : e.w0; e.w1 = lane == 1u ? r : e.w1; e.w2 = lane == 2u ? r : e.w2; e.w3 = lane == 3u ? r : e.w3;
}

PLAN_FN uint32_t min_u32(uint32_t a, uint32_t b) { return a < b ? a : b; }

// Sink: step(w0, w1, w2, w3) / tail(record)
template <class Sink>
PLAN_FN void close_step(Emit& e, Sink& sink) {
    if (e.lanes != 0u) {
        sink.step(e.w0, e.w1, e.w2, e.w3);
        e.w0 = e.w1 = e.w2 = e.w3 = NOP_REC;
        e.lanes = 0u; e.bytes = 0u;
    }
}
// a piece of m <= PIECE bytes at e.op: kind, and where its first byte comes from (src: K_NEAR / K_FAR the absolute output
// position, K_LIT the position in the compressed block).  Joins the open step if there are lanes left and it reads nothing the
// step writes.
template <class Sink>
PLAN_FN void put_piece(Emit& e, uint32_t kind, uint32_t m, uint32_t src, Sink& sink) {
    const uint32_t need = (m + LANE_B - 1u) / LANE_B;
    if (e.lanes + need > G || (kind != K_LIT && e.lanes != 0u && src + m > e.start)) close_step(e, sink);
    if (e.lanes == 0u) e.start = e.op;
    for (uint32_t j = 0; j < need; ++j) {
        const uint32_t nj = min_u32(LANE_B, m - LANE_B * j);
        const uint32_t f = kind == K_NEAR ? ((src & MASK) + LANE_B * j) : src + LANE_B * j;
        emit_set(e, e.lanes + j, rec(kind, nj, e.bytes + LANE_B * j, f));
    }
    e.lanes += need; e.bytes += m; e.op += m;
}
template <class Sink>
PLAN_FN void put_tail(Emit& e, uint32_t kind, uint32_t m, uint32_t field, Sink& sink) {      // m <= PIECE
    for (uint32_t j = 0; j < m; j += LANE_B) sink.tail(rec(kind, min_u32(LANE_B, m - j), 0u, kind == K_LIT ? field + j : field));
    e.op += m;
}

// literals [src, src + n) of the compressed block -> records
template <class Sink>
PLAN_FN void emit_literals(Emit& e, uint32_t src, uint32_t n, Sink& sink) {
    while (n != 0u) {
        uint32_t m;
        if (e.tail) {
            m = min_u32(n, PIECE);
            put_tail(e, K_LIT, m, src, sink);
        } else {
            m = min_u32(min_u32(n, PIECE), W - (e.op & MASK));
            const uint32_t reads = (e.in_len - src) / LANE_B;          // 16-byte reads that stay inside the compressed block
            if ((m + LANE_B - 1u) / LANE_B > reads) {
                m = LANE_B * reads;                                     // whole lanes only; the rest opens the tail
                e.tail = 1u;
            }
            if (m != 0u) put_piece(e, K_LIT, m, src, sink);
            if (e.tail) close_step(e, sink);
        }
        src += m; n -= m;
    }
}

// match (offset, n) at e.op -> records.  offset <= e.op is the caller's check (src/block/decompress.rs:398-402).
template <class Sink>
PLAN_FN void emit_match(Emit& e, uint32_t offset, uint32_t n, Sink& sink) {
    uint32_t per = offset;      // effective offset: a multiple of `offset`, doubled while the match is shorter than it is far
    uint32_t wr = 0u;           // bytes of this match already planned
    while (n != 0u) {
        uint32_t m;
        if (e.tail) {
            m = min_u32(n, PIECE);
            put_tail(e, K_FAR, m, offset, sink);
            n -= m;
            continue;
        }
        const uint32_t cut = per >= PIECE ? PIECE : (per >= LANE_B ? (per & ~(LANE_B - 1u)) : per);
        m = min_u32(min_u32(n, cut), W - (e.op & MASK));
        const bool near = per <= NEAR_MAX;
        const uint32_t src = e.op - per;
        if (near) m = min_u32(m, W - (src & MASK));
        put_piece(e, near ? K_NEAR : K_FAR, m, src, sink);
        n -= m; wr += m;
        // bytes [match start - offset, e.op) now repeat with period `offset`: 2 * per reaches back to e.op - 2 * per, which must
        // not lie before match start - offset
        if (per < PIECE && wr + offset >= 2u * per) per *= 2u;
    }
}
// the block's last piece is in: the open step, then nothing but K_END
template <class Sink>
PLAN_FN void emit_end(Emit& e, Sink& sink) { close_step(e, sink); }

}  // namespace plan
}  // namespace lz4flex_dev
