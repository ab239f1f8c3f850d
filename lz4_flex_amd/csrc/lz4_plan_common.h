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
// positions by prefix sums, cutting every copy into pieces a group of lanes can execute in one step) and leaves a flat
// array of 4-byte records; the REPLAY kernel walks that array, four lanes per block, ~30 instructions per record, and does
// nothing else.  The serial part of a block is then as short as it can be made.
//
// Record (u32):  [31:25] n (1..64 bytes; 0 in K_END)   [24:23] kind   [22:0] field
//   K_NEAR  match piece whose source is still in the block's LDS ring: field = ring address of the source (11 bits)
//   K_LIT   literal piece: field = position of the bytes in the compressed block
//   K_FAR   match piece whose source has left the ring: field = absolute output position of the source
//   K_END   end of the plan (padding behind it is K_END too)
// A piece never reads bytes it writes itself (n <= effective offset; a short-period match is a sequence of pieces with a
// doubling effective offset), never crosses the end of the LDS ring (source or destination), and -- lanes move 16 bytes
// whatever the piece's length -- a literal piece is only in the plan if its 16-byte reads stay inside the compressed
// block; the pieces behind the first one that does not (the last few bytes of a block) form the TAIL, executed byte by
// byte at the end.  (Writes go to the ring, whose slack takes a lane's surplus bytes; memory only sees whole 64-byte
// lines of final bytes, and the bytes behind the last full line byte by byte.)
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
constexpr uint32_t PIECE = 64u;          // bytes per record: 4 lanes x 16 bytes
constexpr uint32_t LANE_B = 16u;
// a near source must still be in the ring when the piece executes: the ring holds the last W bytes, minus the 15 bytes the
// previous piece's last lane may have written past its end, minus slack
constexpr uint32_t NEAR_MAX = W - 64u;
// a far source is read LOOKAHEAD records before the piece executes: it must have been stored by then.  Worst case the
// records in between write LOOKAHEAD * PIECE bytes: NEAR_MAX - 64 must exceed that.  LOOKAHEAD is as deep as that allows:
// the loads of a wavefront return in order, so every load has to be LOOKAHEAD steps' worth of time away from its use or the
// slowest one (a far source that left the L2: one to two microseconds) sets the pace of all of them
constexpr uint32_t LOOKAHEAD = 24u;
constexpr uint32_t FLUSH_EVERY = 4u;     // steps between the write-backs of a block's complete 64-byte lines
// (+ what FLUSH_EVERY steps may leave in the ring + the line that is not complete yet + the 16-byte granules of the read)
static_assert((LOOKAHEAD - 1u + FLUSH_EVERY) * PIECE + 63u + 64u <= NEAR_MAX, "far sources must be stored before they are requested");
constexpr uint32_t LINE_WORDS = 24u;     // records per 96-byte line of the plan (one line per LOOKAHEAD steps and group: 24 bytes per lane)
constexpr uint32_t END_LINES = 3u;       // lines of K_END behind the last record (the replay kernel fetches two lines ahead)
constexpr uint32_t MAX_FIELD = (1u << 23) - 1u;   // positions a record can name: blocks (compressed and decoded) below 8 MiB

constexpr uint32_t K_NEAR = 0u, K_LIT = 1u, K_FAR = 2u, K_END = 3u;
constexpr uint32_t N_SHIFT = 25u, KIND_SHIFT = 23u, KIND_MASK = 3u << KIND_SHIFT;
PLAN_FN uint32_t rec(uint32_t kind, uint32_t n, uint32_t field) { return (n << N_SHIFT) | (kind << KIND_SHIFT) | field; }
PLAN_FN uint32_t rec_kind(uint32_t r) { return (r >> KIND_SHIFT) & 3u; }
PLAN_FN uint32_t rec_n(uint32_t r) { return r >> N_SHIFT; }
PLAN_FN uint32_t rec_field(uint32_t r) { return r & MAX_FIELD; }
// tail records (executed byte by byte, in global memory): K_LIT as above, K_FAR with field = the match OFFSET
constexpr uint32_t END_REC = (K_END << KIND_SHIFT);   // n = 0: no lane moves a byte

// Per-block header the plan kernel leaves for the replay kernel (32 bytes).
struct BlockPlan {
    uint64_t in_off;        // of the compressed block in the batch's input buffer
    uint64_t out_off;       // of the block's sink in the batch's output buffer
    uint32_t first_word;    // index of the plan's first record in the plan array (a multiple of LINE_WORDS)
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
    uint32_t E;         // the block's decoded length
    uint32_t in_len;    // compressed length
    uint32_t tail;      // 1 once a piece had to go to the tail (everything behind it follows)
};

PLAN_FN uint32_t min_u32(uint32_t a, uint32_t b) { return a < b ? a : b; }

// literals [src, src + n) of the compressed block -> records.  Sink: main(record) / tail(record)
template <class Sink>
PLAN_FN void emit_literals(Emit& e, uint32_t src, uint32_t n, Sink& sink) {
    while (n != 0u) {
        uint32_t m;
        if (e.tail) {
            m = min_u32(n, PIECE);
            sink.tail(rec(K_LIT, m, src));
        } else {
            m = min_u32(min_u32(n, PIECE), W - (e.op & MASK));
            const uint32_t r16 = (m + 15u) & ~15u;             // bytes the piece's lanes move
            if (src + r16 > e.in_len) {
                m &= ~15u;                                      // whole lanes only; the rest opens the tail
                e.tail = 1u;
                if (m == 0u) continue;
            }
            sink.main(rec(K_LIT, m, src));
        }
        e.op += m; src += m; n -= m;
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
            sink.tail(rec(K_FAR, m, offset));
            e.op += m; n -= m;
            continue;
        }
        const uint32_t cut = per >= PIECE ? PIECE : (per >= LANE_B ? (per & ~(LANE_B - 1u)) : per);
        m = min_u32(min_u32(n, cut), W - (e.op & MASK));
        const bool near = per <= NEAR_MAX;
        const uint32_t src = e.op - per;
        if (near) m = min_u32(m, W - (src & MASK));
        sink.main(rec(near ? K_NEAR : K_FAR, m, near ? (src & MASK) : src));
        e.op += m; n -= m; wr += m;
        // bytes [match start - offset, e.op) now repeat with period `offset`: 2 * per reaches back to e.op - 2 * per, which must
        // not lie before match start - offset
        if (per < PIECE && wr + offset >= 2u * per) per *= 2u;
    }
}

// how many main records emit_literals / emit_match produce is what the plan kernel's counting pass needs: run them with a
// counting sink.
struct CountSink {
    uint32_t n_main = 0u, n_tail = 0u;
    PLAN_MEM void main(uint32_t) { n_main++; }
    PLAN_MEM void tail(uint32_t) { n_tail++; }
};

}  // namespace plan
}  // namespace lz4flex_dev
