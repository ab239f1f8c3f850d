// lz4_pcd_common.h -- constants and the per-lane sequence walker of the PARALLEL-CHAIN decoder (lz4_decompress_pcd.hip).
// Per-lane scalar code without any wave-level operation: the same source compiles for the host with -DLZ4FLEX_HOST_SIM,
// where tests/sim/pcd_model.cpp (test infrastructure) runs the decoder's algorithm against the oracle.
//
// The decoder this belongs to handles FEW, LARGE blocks (and small batches): one workgroup per block.  Reference semantics:
// src/block/decompress.rs:201-449.  It diagnoses nothing: a block with ANY irregularity (every DecompressError, a sink that
// is too small, a chain that does not converge) is left marked for the reference-order kernel (lz4_decompress.hip), which
// decodes it again and names the exact error -- as lz4_decompress_wave.hip does.
#pragma once
#include <stdint.h>

#ifdef LZ4FLEX_HOST_SIM
#define PCD_FN static inline
#else
#define PCD_FN __device__ __forceinline__
#endif

namespace lz4flex_dev {
namespace pcd {

// ---- geometry ---------------------------------------------------------------------------------------------------------
// A block's compressed stream is consumed in TILES of CT bytes that start at a true token position.  A tile is cut into NP
// PARTS of P bytes; lane k walks part k's token chain from an assumed entry position (the part's first byte, for part 0 the
// tile's true start), marking the token positions it visits in a bitmap and noting where its chain leaves the part (its
// EXIT, a position in a later part).  A chain that starts at the wrong byte falls into step with the true chain after a few
// sequences (median 15-25 bytes), and two chains that share a position are identical from there on.  So: follow
// the exits from part 0 -- exit of part 0 = true entry of the part it lands in, and so on --, re-walk every part whose
// entry changed, and repeat until nothing changes (three rounds on real data).  Parts a sequence jumps over hold no token: dead.
// Then the set bits of the live parts ARE the tile's sequences, in order.
constexpr uint32_t CT = 32768u;           // compressed bytes per tile
constexpr uint32_t CM = 1024u;            // bytes behind the tile that are staged with it (a walk's last sequence reads past the tile)
#ifndef LZ4P_PART
#define LZ4P_PART 128
#endif
constexpr uint32_t P = LZ4P_PART;         // bytes per part
constexpr uint32_t NP = CT / P;           // 256 parts
// every round of walks + exit following makes at least one more live part final (the first one whose entry was wrong), so NP + 1
// rounds always settle a tile; real data needs two or three (a chain in step with the true one after a few sequences), data
// on which chains rarely meet (random bytes over a two-letter alphabet: every byte is a plausible token) degrades to one part
// per round -- slow, never wrong
constexpr uint32_t MAX_ITERS = NP + 2u;
constexpr uint32_t MAXSEQ = CT / 3u + 2u; // a sequence with a match is at least 3 bytes (token, offset)
// ---- copy side --------------------------------------------------------------------------------------------------------
// The tile's sequences are executed in BATCHES of up to BATCH consecutive sequences (THREADS lanes, SEQ_PER_LANE each) on an LDS
// WINDOW of the output: HIST bytes of history before the batch + at most WNEW bytes the batch produces.
constexpr uint32_t THREADS = 1024u;
constexpr uint32_t SEQ_PER_LANE = 2u;
constexpr uint32_t BATCH = THREADS * SEQ_PER_LANE;
#ifndef LZ4P_HIST
#define LZ4P_HIST 26624
#endif
#ifndef LZ4P_WNEW
#define LZ4P_WNEW 49152
#endif
constexpr uint32_t HIST = LZ4P_HIST;
constexpr uint32_t WNEW = LZ4P_WNEW;
constexpr uint32_t WIN = HIST + WNEW;
constexpr uint32_t X_END = 0xFFFFFFFFu;   // exit: the block's last sequence ended exactly at the block's end
constexpr uint32_t X_ERR = 0xFFFFFFFEu;   // exit: this chain cannot be a real one (ran past the end, offset 0, ...)

struct Seq {
    uint32_t lit_src;   // position of the first literal byte
    uint32_t lit;       // literal length
    uint32_t ml;        // match length, 0 for the block's last sequence
    uint32_t off;       // match offset
};

// One sequence whose token is at p (p < ilen); rd(pos) returns the compressed byte at pos < ilen.  Returns the position of
// (rd.u32(pos): the four bytes at pos, pos + 4 <= ilen) the next token, X_END, or X_ERR.  Checks: everything src/block/decompress.rs:244-443 checks WITHOUT knowing the output
// position (the copy side checks offset <= position and the sink's capacity).
// CHECK_OFF = false is the WALK's notion of "next token": the same positions, but an offset of zero does not end the chain --
// a walk only needs where the sequence ends, and the kernel's walk never reads the offset (one LDS round trip per hop instead
// of two).  The copy side decodes every sequence of the true chain again with CHECK_OFF = true, so the error is found there.
template <class R, bool CHECK_OFF = true>
PCD_FN uint32_t parse_seq(const R& rd, uint32_t ilen, uint32_t p, Seq& s) {
    const uint32_t t = rd(p);
    uint32_t q = p + 1u;
    uint32_t lit = t >> 4;
    if (lit == 15u) {                                   // read_integer_ptr :126-157
        if (ilen - q >= 4u && rd.u32(q) == 0xFFFFFFFFu) {
            // (a long run of length bytes -- a 4 MiB literal run has 16 K of them, a 48 KiB match 190: 64 at a time, sixteen independent
            // reads per round trip, then 4 at a time)
            while (ilen - q >= 64u && lit <= 0x7FFF0000u) {
                uint32_t a = 0xFFFFFFFFu;
                for (uint32_t j = 0; j < 16u; ++j) a &= rd.u32(q + 4u * j);
                if (a != 0xFFFFFFFFu) break;
                lit += 16320u;
                q += 64u;
            }
            while (ilen - q >= 4u && rd.u32(q) == 0xFFFFFFFFu) {
                lit += 1020u;
                q += 4u;
                if (lit > 0x7FFFFFFFu) return X_ERR;
            }
        }
        for (;;) {
            if (q >= ilen) return X_ERR;
            const uint32_t b = rd(q);
            q += 1u;
            lit += b;
            if (lit > 0x7FFFFFFFu) return X_ERR;        // (usize in the reference; no real block gets here)
            if (b != 255u) break;
        }
    }
    s.lit_src = q;
    s.lit = lit;
    s.ml = 0u;
    s.off = 0u;
    if (lit > ilen - q) return X_ERR;                   // :346-348
    q += lit;
    if (q == ilen) return X_END;                        // :366-368 the last sequence: literals only
    if (ilen - q < 2u) return X_ERR;                    // :373-375
    const uint32_t off = rd(q) | (rd(q + 1u) << 8);
    q += 2u;
    if (CHECK_OFF && off == 0u) return X_ERR;           // :168-173
    uint32_t ml = 4u + (t & 15u);
    if (ml == 19u) {
        if (ilen - q >= 4u && rd.u32(q) == 0xFFFFFFFFu) {
            while (ilen - q >= 64u && ml <= 0x7FFF0000u) {
                uint32_t a = 0xFFFFFFFFu;
                for (uint32_t j = 0; j < 16u; ++j) a &= rd.u32(q + 4u * j);
                if (a != 0xFFFFFFFFu) break;
                ml += 16320u;
                q += 64u;
            }
            while (ilen - q >= 4u && rd.u32(q) == 0xFFFFFFFFu) {
                ml += 1020u;
                q += 4u;
                if (ml > 0x7FFFFFFFu) return X_ERR;
            }
        }
        for (;;) {
            if (q >= ilen) return X_ERR;
            const uint32_t b = rd(q);
            q += 1u;
            ml += b;
            if (ml > 0x7FFFFFFFu) return X_ERR;
            if (b != 255u) break;
        }
    }
    if (q >= ilen) return X_ERR;                        // :439-443 a block never ends with a match
    s.ml = ml;
    s.off = off;
    return q;
}

}  // namespace pcd
}  // namespace lz4flex_dev
