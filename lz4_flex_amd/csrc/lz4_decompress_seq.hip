// lz4_decompress_seq.hip -- batched LZ4 block decoder, one block per wavefront, ONE LANE PER SEQUENCE ("sequence decoder"), gfx950.
//
// What it replaces: lz4_flex::block::decompress_into / decompress_internal (src/block/decompress.rs:201-449) for many independent
// blocks.  Output bytes are the reference's; an irregular block (every error of src/block/mod.rs:82-98, a sink too small, an offset
// behind the output) is NOT diagnosed here: it is marked and the reference-order decoder of lz4_decompress.hip decodes it again and
// reports the exact error variant and detail (the scheme of lz4_decompress_pcd.hip).
//
// Round 6.  Every other batch decoder in this tree walks a block's token chain with ONE lane and copies its pieces with one
// group of four lanes: the kernel's time is one block's chain (3 380 sequences x ~630 dependent wave-instructions).
// Here nothing is done a sequence at a time:
//   * WALK (the reference's `ip` chain, decompress.rs:244-332, positions only).  The compressed stream is consumed in tiles of
//     3 840 bytes staged in LDS; a tile is cut into 64 parts of 60 bytes (15 dwords: lane k reading part k hits its own bank) and
//     lane k walks part k's chain from an ASSUMED entry, two LDS round trips per hop at most (token + first length byte; the match
//     length byte of a 15-nibble), marking token positions in a 64-bit register mask.  A chain started at a wrong byte falls into
//     step with the true chain after a few sequences; the exits are followed from the tile's true entry (every part leaving into
//     the next one: a DPP move; else pointer jumping with ds_bpermute), parts whose entry was wrong walk again until they meet
//     their first walk's marks (the parallel-chain parse of lz4_decompress_pcd.hip / lz4_decompress_plan.hip, masks only).
//     The set bits of the live parts, compacted into a u16 list in LDS, are the tile's sequences in order.
//   * CHUNKS of 64 consecutive sequences, lane = sequence: token, lengths and offset from two aligned dword-pair reads of the
//     tile (decompress.rs:249-258, 284, 373-391), a DPP prefix sum of literal + match lengths places all 64 in the output at
//     once, every reference check that needs the position (offset <= position :286-289 / :398-402, capacity :346-356) is one
//     ballot.  Literals (<= 64 bytes) are copied by their lane, 16 bytes per access, exact length.  Matches: a source older than
//     the LDS window comes from the written-back output (loads issued at set-up, a chunk ahead of their use); a source in
//     the window copies in ROUNDS -- a lane is ready when every sequence that starts before its source's end is done (the done
//     PREFIX: one v_readlane + one compare per round), all ready lanes copy at once.  A round costs its instructions whatever
//     the number of ready lanes; JSON tiles need ~12 rounds per chunk, text ~5 (tools/chunk_study.py).
//   * the output lives in a linear LDS WINDOW (3.5 KiB) that slides by copying its last 1 280 bytes down; what leaves it has been
//     written back 16 bytes per lane.  Sequences that do not fit a lane (literal runs > 64, matches > 273 bytes or overlapping
//     their source, far matches > 64, anything with more than one length byte) are executed ALONE by the whole wavefront
//     (exact_seq: decompress.rs:334-443 for one sequence of any shape) and cut the chunk in front of them.
// LDS per wavefront: token list 2 560 + tile 4 080 + window 3 584 + 16 = 10 240 bytes: 16 wavefronts per CU (4 096 blocks are one round of
// wavefronts; 8 KiB windows -- 11 per CU, a fifth of the far matches -- were a quarter slower: DESIGN.md 5.2, profiles/r06_seq_decoder.txt).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lz4_device.h"
#include "lz4_pcd_common.h"

namespace lz4flex_dev {
namespace sq {

typedef __attribute__((address_space(3))) uint8_t lds_u8;
typedef __attribute__((address_space(3))) uint16_t lds_u16;
typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef __attribute__((address_space(1))) uint8_t g_u8;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
using pcd::X_END;
using pcd::X_ERR;

#ifndef LZ4S_PB
#define LZ4S_PB 60
#endif
#ifndef LZ4S_R
#define LZ4S_R 3584
#endif
#ifndef LZ4S_KEEP
#define LZ4S_KEEP 1280
#endif
constexpr uint32_t PB = LZ4S_PB;             // bytes per part (an odd number of dwords: lane k reading part k hits its own bank)
constexpr uint32_t NPART = 64u;              // parts per tile = lanes
constexpr uint32_t PT = PB * NPART;          // 3 840 compressed bytes per tile
constexpr uint32_t TPAD = 16u;               // bytes in front of the tile (a lane reads the 16 bytes that END with its literals)
constexpr uint32_t TMARGIN = 224u;           // bytes behind the tile staged with it
constexpr uint32_t TILE_LDS = TPAD + PT + TMARGIN;
constexpr uint32_t POSCAP = PT / 3u;         // sequences per tile: a sequence with a match is at least 3 bytes
constexpr uint32_t POS_LDS = (2u * POSCAP + 15u) & ~15u;
constexpr uint32_t LITMAX = 64u;             // literal run a lane copies itself
constexpr uint32_t FARMAX = 64u;             // match from the written-back output a lane copies itself
constexpr uint32_t POS_LIMIT = 0xFFFF0000u;   // positions in either stream stay below this: `pos + a KiB` never wraps (a block beyond it is the reference-order kernel's)
constexpr uint32_t BIGRUN = 1024u;           // literal runs / matches from here on go memory to memory (exact_seq)
constexpr uint32_t WALK_LITMAX = 200u;       // literal run a hop steps over without the generic walker (its end stays inside the staged bytes)
// LDS: [token list | tile | window | scratch 16]  (the tile is not first: a lane may read up to 16 bytes in front of it)
constexpr uint32_t LDS_POS = 0u, LDS_TILE = LDS_POS + POS_LDS, LDS_WIN = LDS_TILE + TILE_LDS;
static_assert(TILE_LDS % 16u == 0u && PT % 16u == 0u && POS_LDS % 16u == 0u && PB % 4u == 0u && (PB / 4u) % 2u == 1u && PB <= 64u, "geometry");
static_assert(PT - 1u + 4u + 15u + WALK_LITMAX + 4u < PT + TMARGIN, "a hop's length byte lies inside the staged bytes");
static_assert(PT + 1u + LITMAX + 16u <= PT + TMARGIN && PT + 1u + LITMAX + 8u <= PT + TMARGIN, "a lane's literals and offset lie inside the staged bytes");

template <uint32_t R_, uint32_t KEEP_>
struct Geo {
    static constexpr uint32_t R = R_;                              // window bytes
    static constexpr uint32_t KEEP = KEEP_;                        // history a slide keeps
    static constexpr uint32_t BUDGET = (R_ - KEEP_ - 64u) / 2u;    // output bytes of one chunk (two chunks are in flight) / of one cooperative piece
    static constexpr uint32_t SCRATCH = LDS_WIN + R_;              // 16 bytes behind the window: where a lane's 16-byte source read may end, and where the
    static constexpr uint32_t LDS = LDS_WIN + R_ + 16u;            // later rounds' writes of the wrong size class go (29 x 512 bytes: 11 wavefronts per CU)
    static_assert(R_ >= 1024u + 1040u + 16u, "the window holds a long run's extended period");
};

#ifdef LZ4S_PROF      // tools: cycles of every wavefront per phase -> g_sq_prof[0..15], event counts in [16..31]
__device__ unsigned long long g_sq_prof[32];
struct Prof { uint64_t c[32]; uint64_t t; };
#define SQ_PROF_ARG , Prof& P
#define SQ_PROF_PASS , P
#define SQ_TICK(i) { const uint64_t t_ = __builtin_readcyclecounter(); P.c[i] += t_ - P.t; P.t = t_; }
#define SQ_COUNT(i, n) { P.c[i] += (uint64_t)(n); }
#else
#define SQ_PROF_ARG
#define SQ_PROF_PASS
#define SQ_TICK(i)
#define SQ_COUNT(i, n)
#endif

// Behind a region only some lanes execute: an (empty) instruction of its own.  Without it the compiler lets the region end in the
// block where uniform paths (an early return, a loop's exit) meet as well, and then takes every value merged there -- the
// function's result, the loop's state -- for divergent: masks and counters move to vector registers, uniform branches become
// exec-mask loops (the first build of this file ran its whole main loop that way).
#define SQ_JOIN() asm volatile("; join")
#define LZ4S_DPP(v, ctrl, rmask) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), (ctrl), (rmask), 0xf, false))
__device__ __forceinline__ uint32_t wave_incl_add(uint32_t v) {
    v += LZ4S_DPP(v, 0x111, 0xf);     // row_shr:1
    v += LZ4S_DPP(v, 0x112, 0xf);     // row_shr:2
    v += LZ4S_DPP(v, 0x114, 0xf);     // row_shr:4
    v += LZ4S_DPP(v, 0x118, 0xf);     // row_shr:8
    v += LZ4S_DPP(v, 0x142, 0xa);     // row_bcast:15 -> rows 1, 3
    v += LZ4S_DPP(v, 0x143, 0xc);     // row_bcast:31 -> rows 2, 3
    return v;
}
__device__ __forceinline__ uint32_t rdlane(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint64_t ballot(bool b) { return __builtin_amdgcn_ballot_w64(b); }
__device__ __forceinline__ bool lanes(uint64_t m) { return __builtin_amdgcn_inverse_ballot_w64(m); }      // this lane's bit of a wave-uniform mask
__device__ __forceinline__ uint32_t ctz64(uint64_t x) { return (uint32_t)__builtin_ctzll(x); }
__device__ __forceinline__ uint64_t low_mask(uint32_t n) { return n >= 64u ? ~0ull : (1ull << n) - 1ull; }
__device__ __forceinline__ uint32_t bperm(uint32_t lane_src, uint32_t v) { return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(lane_src * 4u), (int)v); }

// LDS by byte address (the dynamic segment starts at 0)
__device__ __forceinline__ lds_u8* L8(uint32_t a) { return (lds_u8*)(uintptr_t)a; }
__device__ __forceinline__ u32x4 lds_rd16(uint32_t a) { u32x4 v; __builtin_memcpy(&v, (const void*)L8(a), 16); return v; }
__device__ __forceinline__ void lds_wr16(uint32_t a, const u32x4& v) { __builtin_memcpy((void*)L8(a), &v, 16); }
__device__ __forceinline__ void lds_wr8(uint32_t a, uint32_t lo, uint32_t hi) { const u32x2 v = {lo, hi}; __builtin_memcpy((void*)L8(a), &v, 8); }
__device__ __forceinline__ void lds_wr4(uint32_t a, uint32_t v) { __builtin_memcpy((void*)L8(a), &v, 4); }
__device__ __forceinline__ void lds_wr2(uint32_t a, uint32_t v) { const uint16_t t = (uint16_t)v; __builtin_memcpy((void*)L8(a), &t, 2); }
__device__ __forceinline__ void lds_wr1(uint32_t a, uint32_t v) { *L8(a) = (uint8_t)v; }
// the four bytes at LDS address a (any alignment) out of two aligned dwords
__device__ __forceinline__ uint32_t lds_rd4u(uint32_t a) {
    const lds_u32* q = (const lds_u32*)(uintptr_t)(a & ~3u);
    const uint32_t d0 = q[0], d1 = q[1];
    return __builtin_amdgcn_alignbyte(d1, d0, a & 3u);
}

// n (1..16) bytes of v to LDS, exactly (lz4_decompress_wave.hip write_exact16)
__device__ __forceinline__ void write_exact16(uint32_t dst, const u32x4& v, uint32_t n) {
    if (n >= 16u) { lds_wr16(dst, v); return; }
    const bool n8 = (n & 8u) != 0u, n4 = (n & 4u) != 0u, n2 = (n & 2u) != 0u;
    const uint32_t w4 = n8 ? v.z : v.x;                               // the dword at byte offset (n & 8)
    const uint32_t wq = n8 ? (n4 ? v.w : v.z) : (n4 ? v.y : v.x);     // the dword at byte offset (n & 12)
    if (n8) lds_wr8(dst, v.x, v.y);
    if (n4) lds_wr4(dst + (n & 8u), w4);
    if (n2) lds_wr2(dst + (n & 12u), wq);
    if (n & 1u) lds_wr1(dst + (n & 14u), wq >> (n2 ? 16 : 0));
}

// the compressed bytes for the generic sequence walker (lz4_pcd_common.h parse_seq): the staged tile from LDS, else memory
struct Reader {
    const g_u8* g;
    uint32_t t0;
    __device__ __forceinline__ uint32_t operator()(uint32_t pos) const {
        const uint32_t r = pos - t0;
        return r < PT + TMARGIN ? (uint32_t)*L8(LDS_TILE + TPAD + r) : (uint32_t)g[pos];
    }
    __device__ __forceinline__ uint32_t u32(uint32_t pos) const { return (*this)(pos) | ((*this)(pos + 1u) << 8) | ((*this)(pos + 2u) << 16) | ((*this)(pos + 3u) << 24); }
};
// where the sequence at position p ends (the next token; ilen: the block ends there), exactly, byte by byte -- the walk's rare path, kept
// out of its loop.  X_ERR: this chain cannot be a real one.
__device__ __noinline__ uint32_t slow_next(const g_u8* g, uint32_t t0, uint32_t ilen, uint32_t p) {
    Reader rd;
    rd.g = g; rd.t0 = t0;
    pcd::Seq q;
    const uint32_t nx = pcd::parse_seq<Reader, false>(rd, ilen, p, q);
    return nx == X_END ? ilen : nx;
}

struct Part {           // positions relative to the tile's first byte t0
    uint64_t marks;      // token positions of the standing walk, relative to the part's first byte
    uint32_t from;       // where the standing walk began (X_ERR: none)
    uint32_t exit;       // where its chain leaves the part: a position >= the part's end (>= ilen - t0: the block ends or fails there), or X_ERR
};

// One walk of a part from r (decompress.rs:244-258, 366-391: positions only).  FIRST: every position is marked.  Else: until the walk
// lands on a position the standing walk marked (its marks stand from there, and its exit) or leaves the part.  A hop is two LDS
// round trips at most: token + first length byte, and the match length byte of a 15-nibble.
template <bool FIRST>
__device__ __forceinline__ void walk_part(const g_u8* g, uint32_t t0, uint32_t ilen, uint32_t r, uint32_t p0, uint32_t pend, Part& s) {
    const uint32_t entry = r;
    const uint32_t tb = LDS_TILE + TPAD;
    uint64_t m2 = 0ull;
    uint32_t exit_ = X_ERR;
    bool merged = false;
    for (;;) {
        bool slow = false;
        uint64_t bit = 0ull;
        for (;;) {
            if (r >= pend) { exit_ = r; break; }
            bit = 1ull << (r - p0);
            if (!FIRST && (s.marks & bit) != 0ull) { merged = true; break; }
            const uint32_t w = lds_rd4u(tb + r);
            const uint32_t L = (w >> 4) & 15u, M = w & 15u, e1 = (w >> 8) & 0xFFu;
            const bool l15 = L == 15u;
            uint32_t nx = r + (l15 ? 15u + e1 + 4u : L + 3u);
            slow = l15 && e1 > WALK_LITMAX - 15u;
            if (M == 15u && !slow) {
                const uint32_t e2 = *L8(tb + nx);
                nx += 1u;
                slow = e2 == 255u;
            }
            if (slow) break;
            m2 |= bit;
            r = nx;
        }
        if (!slow) break;
        const uint32_t nx = slow_next(g, t0, ilen, t0 + r);     // (r < pend: a position of the block)
        if (nx == X_ERR) break;
        m2 |= bit;
        r = nx - t0;
    }
    if (merged) { s.marks = m2 | (s.marks & ~((1ull << (r - p0)) - 1ull)); s.from = entry; }
    else { s.marks = m2; s.from = entry; s.exit = exit_; }
}

template <class G>
struct Dec {
    const g_u8* in;
    g_u8* out;
    uint32_t ilen, cap, lane;
    uint32_t OP;         // bytes produced
    uint32_t W0;         // the window holds [W0, OP), W0 a multiple of 16
    uint32_t F;          // bytes written back, a multiple of 16 (W0 <= F)

    // window -> output, whole 16-byte units of [F, op)
    __device__ __forceinline__ void write_back(uint32_t op) {
        const uint32_t lim = op & ~15u;
        for (uint32_t q0 = F; q0 < lim; q0 += 1024u) {         // (uniform trip counts, the lanes' share inside: a loop that lanes leave at different
            const uint32_t q = q0 + 16u * lane;                 //  times makes the compiler take every value merged behind it for divergent)
            if (q < lim) {
                const u32x4 v = lds_rd16(LDS_WIN + (q - W0));
                __builtin_memcpy((void*)(out + q), &v, 16);
            }
            SQ_JOIN();
        }
        F = lim;
    }
    // make room for `need` (<= 2 BUDGET) more bytes behind op: write back, move the last KEEP bytes to the window's start
    __device__ __forceinline__ void ensure(uint32_t op, uint32_t need) {
        if (op - W0 + need <= G::R) return;
        write_back(op);
        const uint32_t nw = (op - G::KEEP) & ~15u;           // (op - W0 > KEEP here: need <= R - KEEP - 64)
        const uint32_t S = nw - W0, n = op - nw;
        for (uint32_t i0 = 0u; i0 < n; i0 += 1024u) {        // a lower iteration never writes what a higher one reads (S >= 0); within one, reads precede writes
            const uint32_t i = i0 + 16u * lane;
            if (i < n) {
                const u32x4 v = lds_rd16(LDS_WIN + S + i);
                lds_wr16(LDS_WIN + i, v);
            }
            SQ_JOIN();
        }
        W0 = nw;
    }
    __device__ __forceinline__ void finish() {
        write_back(OP);
        if (F + lane < OP) out[F + lane] = *L8(LDS_WIN + F + lane - W0);
        SQ_JOIN();
        F = OP;
    }
    // ---- runs of BIGRUN bytes or more go memory to memory, 16 bytes per lane, and the window is read back behind them ----------------
    // everything produced so far, the last odd bytes too
    __device__ __forceinline__ void flush_all() {
        write_back(OP);
        if (F + lane < OP) out[F + lane] = *L8(LDS_WIN + F + lane - W0);
        SQ_JOIN();
    }
    // the window = the last KEEP bytes of the output, from memory (this wavefront's own stores: one CU, one L1 -- coherent in program order)
    __device__ __forceinline__ void reload_window() {
        W0 = OP > G::KEEP ? (OP - G::KEEP) & ~15u : 0u;
        const uint32_t n = OP - W0;
        for (uint32_t i0 = 0u; i0 < n; i0 += 1024u) {
            const uint32_t i = i0 + 16u * lane;
            if (i < n) {
                u32x4 v = {0u, 0u, 0u, 0u};
                if (W0 + i + 16u <= OP) __builtin_memcpy(&v, (const void*)(out + W0 + i), 16);
                else {
                    uint32_t wv[4] = {0u, 0u, 0u, 0u};
                    for (uint32_t k = 0u; k < 16u; ++k) if (W0 + i + k < OP) wv[k >> 2] |= (uint32_t)out[W0 + i + k] << (8u * (k & 3u));
                    v = u32x4{wv[0], wv[1], wv[2], wv[3]};
                }
                lds_wr16(LDS_WIN + i, v);
            }
            SQ_JOIN();
        }
        F = OP & ~15u;
    }
    // n bytes from memory at `from` to the output at OP, 1 KiB per step in order (a step's source may be an earlier step's destination)
    __device__ __forceinline__ void mem_copy(const g_u8* from, uint32_t n) {
        for (uint32_t i0 = 0u; i0 < n; i0 += 1024u) {
            const uint32_t i = i0 + 16u * lane;
            if (i + 16u <= n) {
                u32x4 v;
                __builtin_memcpy(&v, (const void*)(from + i), 16);
                __builtin_memcpy((void*)(out + OP + i), &v, 16);
            } else if (i < n) {
                for (uint32_t k = i; k < n; ++k) out[OP + k] = from[k];
            }
            SQ_JOIN();
        }
    }
    __device__ __forceinline__ uint32_t mod_small(uint32_t i, uint32_t m, float rcp) {      // i mod m, i < 2^22, rcp = 1 / m
        const uint32_t q = (uint32_t)((float)i * rcp);
        uint32_t r = i - q * m;                                    // q is off by at most one either way
        r = (int32_t)r < 0 ? r + m : r;
        return r >= m ? r - m : r;
    }
    __device__ void big_literals(uint32_t src, uint32_t n) {
        flush_all();
        mem_copy(in + src, n);
        OP += n;
        reload_window();
    }
    // a long match.  offset >= 1024: a step's source lies in front of the step: memory to memory.  Else the periodic form
    // out[d + i] = out[d - offset + i mod offset] (decompress.rs:57-82, decompress_safe.rs:301-318; offset 1 = a run of one byte, :311-313):
    // the period, repeated to 1 KiB + 16 + offset bytes in the (flushed) window's place, is what every step stores a kibibyte of
    __device__ void big_match(uint32_t offset, uint32_t n) {
        flush_all();
        if (offset >= 1024u) {
            mem_copy(out + OP - offset, n);
        } else {
            const float rcp = 1.0f / (float)offset;
            const uint32_t el = offset + 1040u;
            for (uint32_t k0 = 0u; k0 < el; k0 += 64u) {
                const uint32_t k = k0 + lane;
                if (k < el) *L8(LDS_WIN + k) = out[OP - offset + mod_small(k, offset, rcp)];
                SQ_JOIN();
            }
            uint32_t ph = 0u;                                      // i0 mod offset
            const uint32_t adv = mod_small(1024u, offset, rcp);
            for (uint32_t i0 = 0u; i0 < n; i0 += 1024u) {
                const uint32_t i = i0 + 16u * lane;
                if (i < n) {
                    const u32x4 v = lds_rd16(LDS_WIN + ph + 16u * lane);
                    if (i + 16u <= n) __builtin_memcpy((void*)(out + OP + i), &v, 16);
                    else {
                        const uint32_t wv[4] = {v.x, v.y, v.z, v.w};
                        for (uint32_t k = 0u; i + k < n; ++k) out[OP + i + k] = (uint8_t)(wv[k >> 2] >> (8u * (k & 3u)));
                    }
                }
                SQ_JOIN();
                ph += adv;
                ph = ph >= offset ? ph - offset : ph;
            }
        }
        OP += n;
        reload_window();
    }
    // literals of any length from the compressed stream, whole wavefront, <= BUDGET bytes per piece
    __device__ __forceinline__ void coop_literals(uint32_t src, uint32_t n) {
        if (n >= BIGRUN) { big_literals(src, n); return; }
        for (uint32_t c = 0u; c < n; c += G::BUDGET) {
            const uint32_t m = n - c < G::BUDGET ? n - c : G::BUDGET;
            ensure(OP, m);
            for (uint32_t i0 = 0u; i0 < m; i0 += 64u) {
                const uint32_t i = i0 + lane;
                if (i < m) *L8(LDS_WIN + OP - W0 + i) = in[src + c + i];
                SQ_JOIN();
            }
            OP += m;
        }
    }
    // a match of any offset / length, whole wavefront, 64 bytes per step.  Source bytes from the window when it holds them, else
    // from the output written back earlier.  offset < 64: the periodic form out[d + i] = out[d - offset + i mod offset]
    // (decompress.rs:57-82, decompress_safe.rs:301-318), which only reads bytes in front of the match.
    __device__ __forceinline__ void coop_match(uint32_t offset, uint32_t n) {
        if (n >= BIGRUN) { big_match(offset, n); return; }
        const float rcp = offset < 64u ? 1.0f / (float)offset : 0.0f;
        for (uint32_t c = 0u; c < n; c += G::BUDGET) {
            const uint32_t m = n - c < G::BUDGET ? n - c : G::BUDGET;
            ensure(OP, m);
            const uint32_t d = OP, src = d - offset;
            for (uint32_t s0 = 0u; s0 < m; s0 += 64u) {
                const uint32_t i = s0 + lane;
                if (i < m) {
                    uint32_t si = i;
                    if (offset < 64u) {
                        const uint32_t q = (uint32_t)((float)i * rcp);
                        uint32_t r = i - q * offset;                  // q is off by at most one either way
                        r = (int32_t)r < 0 ? r + offset : r;
                        r = r >= offset ? r - offset : r;
                        si = r;
                    }
                    const uint32_t ps = src + si;
                    const uint8_t byte = ps >= W0 ? *L8(LDS_WIN + ps - W0) : out[ps];
                    *L8(LDS_WIN + d - W0 + i) = byte;
                }
                SQ_JOIN();
            }
            OP += m;
        }
    }
    // One sequence of any shape at `ip`, by the whole wavefront, with the reference's checks (decompress.rs:334-443; any violation ->
    // false: the reference-order kernel decodes the block again and names the error).  done: the block ended here.
    __device__ bool exact_seq(uint32_t ip, bool& done) {
        const uint32_t t = uni(in[ip]);
        ip += 1u;
        uint32_t lit = t >> 4;
        if (lit == 15u) {
            for (;;) {
                if (ip >= ilen) return false;
                const uint32_t b = uni(in[ip]);
                ip += 1u;
                lit += b;
                if (lit > 0x7FFFFFFFu) return false;      // (a 32-bit sum must not wrap: the reference counts in usize)
                if (b != 255u) break;
            }
        }
        if (lit > ilen - ip || lit > cap - OP || OP + lit > POS_LIMIT) return false;
        coop_literals(ip, lit);
        ip += lit;
        if (ip >= ilen) { done = true; return true; }
        if (ilen - ip < 2u) return false;
        const uint32_t offset = uni((uint32_t)in[ip] | ((uint32_t)in[ip + 1u] << 8));
        ip += 2u;
        if (offset == 0u) return false;
        uint32_t ml = 4u + (t & 15u);
        if (ml == 19u) {
            for (;;) {
                if (ip >= ilen) return false;
                const uint32_t b = uni(in[ip]);
                ip += 1u;
                ml += b;
                if (ml > 0x7FFFFFFFu) return false;
                if (b != 255u) break;
            }
        }
        if (offset > OP || ml > cap - OP || OP + ml > POS_LIMIT) return false;
        coop_match(offset, ml);
        if (ip >= ilen) return false;              // a match is always followed by another token (decompress.rs:439-443)
        return true;
    }
};

// A chunk: up to 64 consecutive sequences of the tile's token list, lane = sequence, placed but not yet copied
struct Chunk {
    uint32_t lit, ml, dst, src, lsr;    // per lane: lengths, output position of the literals, source position of the match, literals' position relative to t0
    u32x4 f0, f1, f2, f3;               // a far match's source bytes (requested at set-up, used a chunk later)
    uint64_t act, haslit, near, far;    // lanes: in the chunk; with literals; match from the window; match from the written-back output
    uint32_t nact, T, tp0;              // sequences, output bytes; nact == 0: the sequence at tp0 (relative to t0) is executed alone by the wavefront
    bool last;                          // the block's last sequence is the chunk's last
};

// Set-up of the chunk that starts at sequence sidx with `op` bytes in front of it -- of which the last `pending` are still being
// produced by the chunk before (room for both is made here: a slide never moves a placed chunk).  false: the block is irregular.
template <class G>
__device__ __forceinline__ bool setup_chunk(Dec<G>& D, uint32_t t0, uint32_t n_tile, uint32_t sidx, uint32_t op, uint32_t pending, Chunk& C) {
    const uint32_t lane = D.lane;
    const uint32_t ilr = D.ilen - t0;                          // the block's end, relative to t0
    const uint32_t tb = LDS_TILE + TPAD;
    const uint32_t nrem = n_tile - sidx;
    const uint32_t idx = sidx + (lane < nrem ? lane : nrem - 1u);
    const uint32_t tpr = (uint32_t)*(const lds_u16*)(uintptr_t)(LDS_POS + 2u * idx);
    // ---- token, lengths, offset (decompress.rs:249-258, 284, 373-391) ----------------------------------------------------
    const uint32_t w = lds_rd4u(tb + tpr);
    const uint32_t L = (w >> 4) & 15u, M = w & 15u, e1 = (w >> 8) & 0xFFu;
    const bool l15 = L == 15u;
    const uint32_t lit = l15 ? 15u + e1 : L;
    const uint32_t lsr = tpr + (l15 ? 2u : 1u);
    const uint64_t biglit = ballot(lit > LITMAX);              // (covers a length byte of 255) not a lane's work: exact_seq
    const uint32_t lend = lit > LITMAX ? 0u : lsr + lit;
    const uint32_t w1 = lds_rd4u(tb + lend);
    const uint32_t off = w1 & 0xFFFFu, e2 = (w1 >> 16) & 0xFFu;
    const bool m15 = M == 15u;
    const uint32_t mlx = 4u + M + (m15 ? e2 : 0u);
    const uint32_t nxt = lend + (m15 ? 3u : 2u);               // the next token
    const uint64_t lastm = ballot(lend >= ilr);                // the block's last sequence: literals only (:366-368) -- or an error
    const uint64_t errm = ballot(lend > ilr) |                                                           // :346-348
                          (~lastm & (ballot(lend + 2u > ilr) | ballot(nxt >= ilr) | ballot(off == 0u))); // :373-375, :439-443, :168-173
    const uint64_t bigm = biglit | (~lastm & (ballot(m15 && e2 == 255u) | ballot(off < mlx)));           // more length bytes; a match that reads its own output
    const uint32_t ml = lanes(lastm) ? 0u : mlx;
    // ---- the chunk: up to the first sequence that is not a lane's work, and at most BUDGET bytes ------------------------------
    uint32_t nact = nrem < 64u ? nrem : 64u;
    {
        const uint64_t bm = bigm & low_mask(nact);
        if (bm != 0ull) nact = ctz64(bm);
    }
    const uint32_t u = lanes(low_mask(nact)) ? lit + ml : 0u;
    const uint32_t incl = wave_incl_add(u);
    {
        const uint64_t over = ballot(incl > G::BUDGET) & low_mask(nact);
        if (over != 0ull) nact = ctz64(over);
    }
    uint32_t T = nact != 0u ? rdlane(incl, nact - 1u) : 0u;
    C.tp0 = rdlane(tpr, 0u);
    if (nact != 0u) {
        if ((errm & low_mask(nact)) != 0ull) return false;
        if (T > D.cap - op || op + T > POS_LIMIT) return false;    // OutputTooSmall (:349-356, :403-408): named by the reference-order kernel
        D.write_back(op - pending);                            // (every chunk: a source that leaves the window has to be in memory)
        D.ensure(op - pending, pending + T);
    }
    const uint32_t dst = op + incl - u, dm = dst + lit;
    const uint32_t src = dm - off;
    uint64_t am = low_mask(nact);
    const uint64_t hasm = ~lastm;
    if ((ballot(off > dm) & hasm & am) != 0ull) return false;  // OffsetOutOfBounds (:286-289, :398-402)
    // a source in front of the window comes from the written-back output: all of it has to be there, and short enough for a lane.
    // "The window" is what the NEXT chunk's set-up will have left of it when this chunk is copied: a slide keeps KEEP bytes in front
    // of the chunk that is pending then -- this one
    const uint32_t near_lo = op > G::KEEP && op - G::KEEP > D.W0 ? op - G::KEEP : D.W0;
    const uint64_t farm = ballot(src < near_lo) & hasm;
    {
        const uint64_t fbig = farm & am & (ballot(src + ml > D.F) | ballot(ml > FARMAX));
        if (fbig != 0ull) { nact = ctz64(fbig); am = low_mask(nact); T = nact != 0u ? rdlane(incl, nact - 1u) : 0u; }
    }
    C.lit = lit; C.ml = ml; C.dst = dst; C.src = src; C.lsr = lsr;
    C.nact = nact; C.T = T;
    C.act = am;
    C.haslit = ballot(lit != 0u) & am;
    C.far = farm & am;
    C.near = hasm & am & ~farm;
    C.last = nact != 0u && ((lastm >> (nact - 1u)) & 1ull) != 0ull;
    // ---- far sources: requested now, used a chunk later --------------------------------------------------------------------------
    C.f0 = u32x4{0u, 0u, 0u, 0u}; C.f1 = C.f0; C.f2 = C.f0; C.f3 = C.f0;
    if (lanes(C.far)) {
        __builtin_memcpy(&C.f0, (const void*)(D.out + src), 16);
        if (ml > 16u) __builtin_memcpy(&C.f1, (const void*)(D.out + src + ml - 16u), 16);
        if (ml > 32u) __builtin_memcpy(&C.f2, (const void*)(D.out + src + 16u), 16);
        if (ml > 48u) __builtin_memcpy(&C.f3, (const void*)(D.out + src + 32u), 16);
    }
    SQ_JOIN();
    return true;
}

// What a lane needs to copy its n bytes (1 <= n) from LDS address s to LDS address d: the addresses of the LAST 16 bytes on both sides
// and the lanes of every size class.  One 16-byte read (16 bytes and more: a second one, of the LAST 16 bytes) and two overlapping
// writes of the size class; more than 32 bytes: the pieces between them.  Source and destination do not
// overlap.  Computed once per chunk; the rounds only AND the classes with their ready lanes.
struct CopyPlan {
    uint32_t s, d, s2, d2, n;
    uint64_t c16, c8, c4, c2, c1, gt32;       // n >= 16; 8..15; 4..7; 2..3; 1; n > 32
};
template <bool SMALL>
__device__ __forceinline__ CopyPlan plan_copy(uint32_t s, uint32_t d, uint32_t n) {
    CopyPlan P;
    P.s = s; P.d = d; P.n = n; P.s2 = s + n - 16u; P.d2 = d + n - 16u;
    const uint64_t ge16 = ballot(n >= 16u), ge8 = ballot(n >= 8u);
    P.c16 = ge16; P.c8 = ge8 & ~ge16; P.gt32 = ballot(n > 32u);
    if (SMALL) {
        const uint64_t ge4 = ballot(n >= 4u), ge2 = ballot(n >= 2u);
        P.c4 = ge4 & ~ge8; P.c2 = ge2 & ~ge4; P.c1 = ~ge2;
    } else { P.c4 = ~ge8; P.c2 = 0ull; P.c1 = 0ull; }
    return P;
}
template <bool SMALL>
__device__ __forceinline__ void lane_copy(const CopyPlan& P, uint64_t m) {
    u32x4 r1;
    if (lanes(m)) r1 = lds_rd16(P.s);
    // 16 bytes and more: the last 16 bytes are a second read; less: they are bytes of the first (an LDS instruction costs the CU's one LDS
    // pipe whatever the number of lanes, a handful of vector instructions costs one of four SIMDs)
    if (lanes(m & P.c16)) { const u32x4 r2 = lds_rd16(P.s2); lds_wr16(P.d, r1); lds_wr16(P.d2, r2); }
    if (lanes(m & P.c8)) {                                   // n = 8 .. 15: bytes [n - 8, n) of r1
        const uint32_t k = P.n - 8u;
        const bool lo4 = k < 4u;
        const uint32_t a = lo4 ? r1.x : r1.y, b = lo4 ? r1.y : r1.z, c = lo4 ? r1.z : r1.w;
        lds_wr8(P.d, r1.x, r1.y);
        lds_wr8(P.d2 + 8u, __builtin_amdgcn_alignbyte(b, a, k & 3u), __builtin_amdgcn_alignbyte(c, b, k & 3u));
    }
    if (lanes(m & P.c4)) { lds_wr4(P.d, r1.x); lds_wr4(P.d2 + 12u, __builtin_amdgcn_alignbyte(r1.y, r1.x, P.n & 3u)); }     // n = 4 .. 7: bytes [n - 4, n)
    if (SMALL) {
        if (lanes(m & P.c2)) { lds_wr2(P.d, r1.x); lds_wr2(P.d2 + 14u, r1.x >> (8u * (P.n - 2u))); }                          // n = 2, 3
        if (lanes(m & P.c1)) lds_wr1(P.d, r1.x);
    }
    SQ_JOIN();
    uint64_t more = m & P.gt32;
    for (uint32_t p = 16u; more != 0ull; p += 16u) {
        if (lanes(more)) { const u32x4 r = lds_rd16(P.s + p); lds_wr16(P.d + p, r); }
        SQ_JOIN();
        more &= ballot(p + 32u < P.n);                        // the next piece, at p + 16, begins in front of the last one's start: p + 16 < n - 16
    }
}

// The copies of a placed chunk (decompress.rs:276-280, 314-325, 357-361, 410-437).
template <class G>
__device__ __forceinline__ void exec_chunk(Dec<G>& D, const Chunk& C SQ_PROF_ARG) {
    const uint32_t wb = LDS_WIN - D.W0;                        // window address of output position 0
    const uint32_t wl = wb + C.dst, wm = wl + C.lit;
    // ---- literals: 16 bytes per access, exact length ---------------------------------------------------------------------------
    if (C.haslit != 0ull) { const CopyPlan L = plan_copy<true>(LDS_TILE + TPAD + C.lsr, wl, C.lit); lane_copy<true>(L, C.haslit); }
    SQ_TICK(6)
    // ---- far matches: the bytes requested at set-up -------------------------------------------------------------------------------
    if (C.far != 0ull) {
        if (lanes(C.far)) {
            if (C.ml <= 16u) write_exact16(wm, C.f0, C.ml);
            else {
                lds_wr16(wm, C.f0);
                if (C.ml > 32u) lds_wr16(wm + 16u, C.f2);
                if (C.ml > 48u) lds_wr16(wm + 32u, C.f3);
                lds_wr16(wm + C.ml - 16u, C.f1);
            }
        }
        SQ_JOIN();
    }
    SQ_TICK(7) SQ_COUNT(17, 1) SQ_COUNT(20, C.nact) SQ_COUNT(23, __builtin_popcountll(C.far)) SQ_COUNT(24, __builtin_popcountll(C.near))
    // ---- matches inside the window, in rounds: ready = every sequence that starts before the source's end is done ----------------
    uint64_t todo = C.near;
    const CopyPlan M = plan_copy<false>(wb + C.src, wm, C.ml);
    const uint32_t s1 = C.src + C.ml;
    if (todo != 0ull) {
        // the first round: most of the chunk's matches (every source in front of the chunk)
        const uint32_t dp = ctz64(todo);                      // every lane below dp is done: all bytes in front of its sequence are final
        const uint32_t S = rdlane(C.dst, dp);
        const uint64_t rm = todo & (ballot(s1 <= S) | (1ull << dp));
        lane_copy<false>(M, rm);
        todo &= ~rm;
        SQ_COUNT(18, 1)
#ifdef LZ4S_EXP_NOROUNDS      // timing experiments only (wrong bytes): the first round alone
        todo = 0ull;
#endif
    }
    while (todo != 0ull) {
        const uint32_t dp = ctz64(todo);
        const uint32_t S = rdlane(C.dst, dp);
        const uint64_t rm = todo & (ballot(s1 <= S) | (1ull << dp));
        lane_copy<false>(M, rm);
        todo &= ~rm;
        SQ_COUNT(18, 1)
#ifdef LZ4S_EXP_NOROUNDS      // timing experiments only (wrong bytes): the first round alone
        todo = 0ull;
#endif
    }
    SQ_TICK(8)
}

// The chunks of one tile: sequences [0, n_tile) of the token list.  Two chunks are in flight: the next one is set up (token decode,
// placement, checks, room in the window, requests for far sources) before the current one is copied.  false: the block is irregular.
template <class G>
__device__ __forceinline__ bool run_chunks(Dec<G>& D, uint32_t t0, uint32_t n_tile, bool& done SQ_PROF_ARG) {
    uint32_t sidx = 0u;
    Chunk C;
    if (!setup_chunk<G>(D, t0, n_tile, 0u, D.OP, 0u, C)) return false;
    SQ_TICK(4)
    for (;;) {
        if (C.nact == 0u) {                                    // the sequence alone, by the whole wavefront
            if (!D.exact_seq(t0 + C.tp0, done)) return false;
            sidx += 1u;
            SQ_TICK(9) SQ_COUNT(19, 1)
            if (done) return sidx == n_tile;
            if (sidx >= n_tile) return true;
            if (!setup_chunk<G>(D, t0, n_tile, sidx, D.OP, 0u, C)) return false;
            SQ_TICK(4)
            continue;
        }
        const uint32_t nsidx = sidx + C.nact, nop = D.OP + C.T;
        const bool have_next = nsidx < n_tile && !C.last;
        Chunk N;
        N.nact = 0u; N.T = 0u; N.tp0 = 0u; N.last = false; N.act = 0ull; N.haslit = 0ull; N.near = 0ull; N.far = 0ull;
        N.lit = 0u; N.ml = 0u; N.dst = 0u; N.src = 0u; N.lsr = 0u;
        N.f0 = u32x4{0u, 0u, 0u, 0u}; N.f1 = N.f0; N.f2 = N.f0; N.f3 = N.f0;
        if (have_next) { if (!setup_chunk<G>(D, t0, n_tile, nsidx, nop, C.T, N)) return false; }
        SQ_TICK(4)
#ifndef LZ4S_EXP_NOEXEC         // timing experiments only (wrong bytes): chunks are placed, nothing is copied
        exec_chunk<G>(D, C SQ_PROF_PASS);
#endif
        D.OP = nop;
        sidx = nsidx;
        if (C.last) { done = true; return sidx == n_tile; }
        if (!have_next) return true;
        C = N;
    }
}

#ifdef LZ4S_WAVES      // tools: wavefronts per SIMD the register allocation aims at (the kernel's 116 VGPRs allow 4: 16 per CU)
#define LZ4S_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(LZ4S_WAVES, LZ4S_WAVES)))
#else
#define LZ4S_WAVES_ATTR
#endif
template <class G>
__global__ void __launch_bounds__(64) LZ4S_WAVES_ATTR lz4_decompress_seq_kernel(DecompressArgs a, int32_t redo_code) {
    extern __shared__ __attribute__((aligned(16))) uint8_t seq_lds[];
    const uint32_t lane = threadIdx.x;
    const uint32_t b = blockIdx.x;
    if (b >= a.n) return;
    // (every LDS access below goes by byte address from 0: the dynamic segment is the kernel's only LDS)
    if ((uint32_t)(uintptr_t)(lds_u8*)seq_lds != 0u) { if (lane == 0u) { a.status[b] = redo_code; a.out_len[b] = 0u; } return; }
    Dec<G> D;
    D.in = (const g_u8*)(a.in_base + a.in_off[b]);
    D.out = (g_u8*)(a.out_base + a.out_off[b]);
    D.ilen = a.in_len[b];
    D.cap = a.out_cap[b];
    D.lane = lane;
    D.OP = 0u; D.W0 = 0u; D.F = 0u;
    // PREFIX mode (out_pos, round 6; decompress_into_with_prefix-like: a Linked frame's block, src/frame/decompress.rs:195-222,280-305): the
    // sink already holds [0, PFX) of the stream, matches may reach into it, and the block's bytes follow at PFX -- for this decoder a source in
    // front of its window is a read of written-back output anyway; the prefix is the same read.  The window starts as the last KEEP bytes
    // of the prefix.  (Not for CHAINED batches: the bytes must be in memory when the launch starts -- a level of chains per launch.)
    const uint32_t PFX = a.out_pos != nullptr ? uni(a.out_pos[b]) : 0u;
    const uint32_t ilen = D.ilen;
    bool ok = ilen != 0u && ilen <= POS_LIMIT && PFX <= POS_LIMIT / 2u && D.cap >= PFX, done = false;     // (an empty block: decompress.rs:207-209, the reference-order kernel reports it)
    if (PFX != 0u && ok) { D.OP = PFX; D.reload_window(); }
    uint32_t entry = 0u;
#ifdef LZ4S_PROF
    Prof P;
    for (int i = 0; i < 32; ++i) P.c[i] = 0ull;
    P.t = __builtin_readcyclecounter();
#endif
    if (lane < 4u) lds_wr4(LDS_TILE + 4u * lane, 0u);     // the bytes in front of the first tile
    for (;;) {
        // (loop-carried scalars, said to be uniform: a value merged behind a loop that lanes leave at different times -- the walks -- is
        // divergent to the compiler, and everything computed from it lands in vector registers)
        entry = uni(entry); D.OP = uni(D.OP); D.W0 = uni(D.W0); D.F = uni(D.F);
        if (uni((ok && !done) ? 1u : 0u) == 0u) break;
        const uint32_t t0 = entry & ~15u;
        SQ_TICK(15)
        // ---- stage the tile: [t0, t0 + PT + TMARGIN), zeros behind the block --------------------------------------------------------
        for (uint32_t o0 = 0u; o0 < PT + TMARGIN; o0 += 1024u) {
            const uint32_t o = o0 + 16u * lane;
            if (o < PT + TMARGIN) {
                u32x4 v = {0u, 0u, 0u, 0u};
                const uint32_t g = t0 + o;
                if (g + 16u <= ilen) __builtin_memcpy(&v, (const void*)(D.in + g), 16);
                else if (g < ilen) {
                    uint32_t wv[4] = {0u, 0u, 0u, 0u};
#pragma unroll
                    for (uint32_t k = 0u; k < 16u; ++k) if (g + k < ilen) wv[k >> 2] |= (uint32_t)D.in[g + k] << (8u * (k & 3u));
                    v = u32x4{wv[0], wv[1], wv[2], wv[3]};
                }
                lds_wr16(LDS_TILE + TPAD + o, v);
            }
            SQ_JOIN();
        }
        SQ_TICK(0) SQ_COUNT(16, 1)
        // ---- 1. first walks: lane 0 from the tile's entry, the others from their part's first byte (positions relative to t0) --------
        const uint32_t ilr = ilen - t0;
        const uint32_t p0 = PB * lane;
        const uint32_t pend = p0 + PB < ilr ? p0 + PB : ilr;
        const uint32_t entry_r = entry - t0;
        Part s;
        s.marks = 0ull; s.from = X_ERR; s.exit = X_ERR;
        if (p0 < ilr) walk_part<true>(D.in, t0, ilen, lane == 0u ? entry_r : p0, p0, pend, s);
        SQ_JOIN();
        SQ_TICK(1)
        // ---- 2. / 3. which parts does the true chain visit, and where does it enter them?  (lz4_decompress_plan.hip) ----------------
        uint32_t my_entry = X_ERR;
        uint64_t path = 1ull;
        bool settled = false;
        const uint32_t nparts = ilr < PT ? (ilr + PB - 1u) / PB : NPART;       // parts that hold bytes of the block
        for (uint32_t round = 0u; round < NPART + 2u; ++round) {
            const bool inside = s.exit != X_ERR && s.exit < PT && s.exit < ilr;
            const uint32_t nxt = inside ? s.exit / PB : 64u;
            // the usual tile: every part's chain leaves into the NEXT part (no sequence is longer than a part) -- the path is all parts and
            // a part's entry is its left neighbour's exit, one DPP move; else the general form, pointer jumping over the exits
            const uint64_t chain_ok = ballot(nxt == lane + 1u || (lane + 1u >= nparts && nxt == 64u)) | ~low_mask(nparts);
            if (chain_ok == ~0ull) {
                path = low_mask(nparts);
                const uint32_t left = (uint32_t)__builtin_amdgcn_update_dpp((int)X_ERR, (int)s.exit, 0x138, 0xf, 0xf, false);   // wave_shr:1
                my_entry = lane == 0u ? entry_r : (lane < nparts ? left : X_ERR);
            } else {
                uint64_t reach = 1ull << lane;
                uint32_t jump = nxt;
#pragma unroll
                for (uint32_t i = 0u; i < 6u; ++i) {
                    const uint32_t sl = jump < 64u ? jump : lane;
                    const uint32_t rlo = bperm(sl, (uint32_t)reach), rhi = bperm(sl, (uint32_t)(reach >> 32)), j2 = bperm(sl, jump);
                    if (jump < 64u) { reach |= ((uint64_t)rhi << 32) | rlo; jump = j2; }
                }
                path = ((uint64_t)rdlane((uint32_t)(reach >> 32), 0u) << 32) | rdlane((uint32_t)reach, 0u);
                const uint64_t before = path & ((1ull << lane) - 1ull);
                const uint32_t pred = before != 0ull ? 63u - (uint32_t)__builtin_clzll(before) : lane;
                const uint32_t pulled = bperm(pred, s.exit);
                my_entry = lane == 0u ? entry_r : (lanes(path) && before != 0ull ? pulled : X_ERR);
            }
            SQ_JOIN();
            path = ((uint64_t)uni((uint32_t)(path >> 32)) << 32) | uni((uint32_t)path);
            // an entry the standing walk passed through needs no walk: its marks stand from there
            if (my_entry != X_ERR && s.from != my_entry && my_entry - p0 < 64u && ((s.marks >> (my_entry - p0)) & 1ull) != 0ull) {
                s.marks &= ~((1ull << (my_entry - p0)) - 1ull);
                s.from = my_entry;
            }
            SQ_JOIN();
            const uint64_t needm = ballot(my_entry != X_ERR && s.from != my_entry);
            if (needm == 0ull) { settled = true; break; }
            if (lanes(needm)) walk_part<false>(D.in, t0, ilen, my_entry, p0, pend, s);
            SQ_JOIN();
            SQ_COUNT(21, 1)
        }
        // (the compiler folds the loop's uniform exit into the divergent re-walk branch and then takes everything behind it for
        // divergent: say what is uniform)
        settled = uni(settled ? 1u : 0u) != 0u;
        path = ((uint64_t)uni((uint32_t)(path >> 32)) << 32) | uni((uint32_t)path);
        // the last part on the path says where the chain leaves the tile
        const uint32_t tile_exit = settled ? rdlane(s.exit, 63u - (uint32_t)__builtin_clzll(path)) : X_ERR;
        SQ_TICK(2)
        if (tile_exit == X_ERR) { ok = false; break; }
        // ---- 4. the token list ---------------------------------------------------------------------------------------------------------
        uint64_t m = my_entry != X_ERR ? s.marks : 0ull;
        const uint32_t cnt = (uint32_t)__builtin_popcountll(m);
        const uint32_t cincl = wave_incl_add(cnt);
        const uint32_t n_tile = rdlane(cincl, 63u);
        if (n_tile > POSCAP || n_tile == 0u) { ok = false; break; }
        {
            uint32_t at = LDS_POS + 2u * (cincl - cnt);
            while (ballot(m != 0ull) != 0ull) {
                if (m != 0ull) {
                    lds_wr2(at, p0 + ctz64(m));
                    at += 2u;
                    m &= m - 1ull;
                }
                SQ_JOIN();
            }
        }
        SQ_TICK(3)
        // ---- 5. the chunks ----------------------------------------------------------------------------------------------------------------
        bool tdone = false;
#ifdef LZ4S_EXP_NOCHUNKS       // timing experiments only (no output): the walks and the token list alone
        if (tile_exit >= ilr) { done = true; break; }
        entry = t0 + tile_exit;
        continue;
#endif
        if (!run_chunks<G>(D, t0, n_tile, tdone SQ_PROF_PASS)) { ok = false; break; }
        if (tdone) { done = true; break; }
        if (tile_exit >= ilr) { ok = false; break; }        // the chain ran out without a last sequence
        entry = t0 + tile_exit;
        // (a long literal run or match length run jumps over tiles: the next tile starts where the chain goes on)
    }
    if (ok && done) {
        SQ_TICK(15)
        D.finish();
        SQ_TICK(10)
#ifdef LZ4S_PROF
        if (lane == 0u) for (int i = 0; i < 32; ++i) if (P.c[i] != 0ull) atomicAdd(&g_sq_prof[i], (unsigned long long)P.c[i]);
#endif
        if (lane == 0u) {
            a.status[b] = 0;
            a.out_len[b] = D.OP - PFX;
            if (a.detail) { a.detail[2u * b] = 0u; a.detail[2u * b + 1u] = 0u; }
        }
    } else if (lane == 0u) {
        a.status[b] = redo_code;          // decoded again, with the reference's check order, by lz4_decompress_blocks_kernel
        a.out_len[b] = 0u;
    }
}

}  // namespace sq

// Blocks without dictionary (a prefix in the sink is fine, round 6).  Irregular blocks get status `redo_code`; the caller runs launch_decompress with
// only_status = redo_code behind this launch.
hipError_t launch_decompress_seq(const DecompressArgs& a, int32_t redo_code, hipStream_t s) {
    if (a.n == 0u) return hipSuccess;
    if (a.dict_base != nullptr || a.chain_done != nullptr) return hipErrorInvalidValue;       // (a prefix -- out_pos -- is fine: see the kernel)
    typedef sq::Geo<LZ4S_R, LZ4S_KEEP> G;
    static_assert(G::LDS <= 65536u, "the default limit of dynamic LDS: no function attribute to set per device");
    hipLaunchKernelGGL(sq::lz4_decompress_seq_kernel<G>, dim3(a.n), dim3(64), G::LDS, s, a, redo_code);
    return hipGetLastError();
}

}  // namespace lz4flex_dev

#ifdef LZ4S_PROF
extern "C" int lz4flex_debug_seq_prof(unsigned long long* vals, int reset) {
    if (reset) {
        unsigned long long z[32] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(lz4flex_dev::sq::g_sq_prof), z, sizeof z);
        return 0;
    }
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(vals, HIP_SYMBOL(lz4flex_dev::sq::g_sq_prof), 256);
    return 0;
}
#endif
