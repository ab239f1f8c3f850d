// lz4_decompress_seq.hip -- batched LZ4 block decoder, one block per wavefront, ONE LANE PER SEQUENCE ("sequence decoder"), gfx950.
//
// What it replaces: lz4_flex::block::decompress_into / decompress_internal (src/block/decompress.rs:201-449) for many independent
// blocks.  Output bytes are the reference's; an irregular block (every error of src/block/mod.rs:82-98, a sink too small, an offset
// behind the output) is NOT diagnosed here: it is marked and the reference-order decoder of lz4_decompress.hip decodes it again and
// reports the exact error variant and detail (the scheme of lz4_decompress_wave.hip / lz4_decompress_pcd.hip).
//
// Round 6.  Every other batch decoder in this tree walks a block's token chain with ONE lane and copies its pieces with one
// group of four lanes: the kernel's time is one block's chain (3 380 sequences x ~630 dependent wave-instructions).
// Here nothing is done a sequence at a time:
//   * WALK (the reference's `ip` chain, decompress.rs:244-332, positions only).  The compressed stream is consumed in tiles of
//     3 840 bytes staged in LDS; a tile is cut into 64 parts of 60 bytes (15 dwords: lane k reading part k hits its own bank) and
//     lane k walks part k's chain from an ASSUMED entry, one LDS round trip per hop (token + first length byte; the match length
//     byte of a 15-nibble lies right before the next token and is checked by the next hop), marking token positions in a 64-bit
//     register mask.  A chain started at a wrong byte falls into step with the true chain after a few sequences; the exits are
//     followed from the tile's true entry by pointer jumping (ds_bpermute), parts whose entry was wrong walk again until they
//     meet their first walk's marks (the parallel-chain parse of lz4_decompress_pcd.hip / lz4_decompress_plan.hip, masks only).
//     The set bits of the live parts, compacted into a u16 list in LDS, are the tile's sequences in order.
//   * CHUNKS of 64 consecutive sequences, lane = sequence: token, lengths and offset from two aligned dword-pair reads of the
//     tile (decompress.rs:249-258, 284, 373-391), a DPP prefix sum of literal + match lengths places all 64 in the output at
//     once, every reference check that needs the position (offset <= position :286-289 / :398-402, capacity :346-356) is one
//     ballot.  Literals (<= 64 bytes) are copied by their lane, 16 bytes per access, exact length.  Matches: a source older than
//     the LDS window comes from the written-back output (loads issued before the literal copies, used behind them); a source in
//     the window copies in ROUNDS -- a lane is ready when every sequence that starts before its source's end is done (the done
//     PREFIX: one v_readlane + one compare per round), all ready lanes copy at once.  A round costs its instructions whatever
//     the number of ready lanes; JSON tiles need ~12 rounds per chunk, text ~5 (tools/chunk_study.py).
//   * the output lives in a linear LDS WINDOW (8 KiB) that slides by copying its upper half down; what leaves it has been
//     written back 16 bytes per lane.  Sequences that do not fit a lane (literal runs > 64, matches > 273 bytes or overlapping
//     their source, far matches > 64, anything with more than one length byte) are executed ALONE by the whole wavefront
//     (exact_seq: decompress.rs:334-443 for one sequence of any shape) and cut the chunk in front of them.
// LDS per wavefront: tile 4 096 + token list 2 576 + window 8 192 = 14 864 bytes: 11 wavefronts per CU.
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

constexpr uint32_t PB = 60u;                 // bytes per part
constexpr uint32_t NPART = 64u;              // parts per tile = lanes
constexpr uint32_t PT = PB * NPART;          // 3 840 compressed bytes per tile
constexpr uint32_t TPAD = 16u;               // bytes in front of the tile (a hop reads the byte before its token)
constexpr uint32_t TMARGIN = 240u;           // bytes behind the tile staged with it (a lane's literals + offset: <= 2 + 64 + 3 + 8)
constexpr uint32_t TILE_LDS = TPAD + PT + TMARGIN;
constexpr uint32_t POSCAP = PT / 3u + 8u;    // sequences per tile: a sequence with a match is at least 3 bytes
constexpr uint32_t POS_LDS = (2u * POSCAP + 15u) & ~15u;
constexpr uint32_t LITMAX = 64u;             // literal run a lane copies itself
constexpr uint32_t FARMAX = 64u;             // match from the written-back output a lane copies itself
static_assert(TILE_LDS % 16u == 0u && PT % 16u == 0u, "geometry");

template <uint32_t R_>
struct Geo {
    static constexpr uint32_t R = R_;                    // window bytes
    static constexpr uint32_t KEEP = R_ / 2u;            // history a slide keeps
    static constexpr uint32_t BUDGET = R_ - KEEP - 64u;  // output bytes of one chunk / one cooperative piece
    static constexpr uint32_t LDS = TILE_LDS + POS_LDS + R_ + 16u;   // (+ 16: a lane's 16-byte source read may end behind the window)
};

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
__device__ __forceinline__ uint32_t ctz64(uint64_t x) { return (uint32_t)__builtin_ctzll(x); }
__device__ __forceinline__ uint32_t bperm(uint32_t lane_src, uint32_t v) { return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(lane_src * 4u), (int)v); }

__device__ __forceinline__ u32x4 lds_rd16(const lds_u8* p) { u32x4 v; __builtin_memcpy(&v, (const void*)p, 16); return v; }
__device__ __forceinline__ void lds_wr16(lds_u8* p, const u32x4& v) { __builtin_memcpy((void*)p, &v, 16); }
// the four bytes at LDS address a (any alignment) out of two aligned dwords
__device__ __forceinline__ uint32_t lds_rd4u(const lds_u8* base, uint32_t a) {
    const lds_u32* q = (const lds_u32*)(base + (a & ~3u));
    const uint32_t d0 = q[0], d1 = q[1];
    return __builtin_amdgcn_alignbyte(d1, d0, a & 3u);
}

// n (1..16) bytes of v to LDS, exactly (lz4_decompress_wave.hip write_exact16)
__device__ __forceinline__ void write_exact16(lds_u8* dst, const u32x4& v, uint32_t n) {
    if (n >= 16u) { __builtin_memcpy((void*)dst, &v, 16); return; }
    const bool n8 = (n & 8u) != 0u, n4 = (n & 4u) != 0u, n2 = (n & 2u) != 0u;
    const uint32_t w4 = n8 ? v.z : v.x;                               // the dword at byte offset (n & 8)
    const uint32_t wq = n8 ? (n4 ? v.w : v.z) : (n4 ? v.y : v.x);     // the dword at byte offset (n & 12)
    if (n8) { const uint64_t t = (uint64_t)v.x | ((uint64_t)v.y << 32); __builtin_memcpy((void*)dst, &t, 8); }
    if (n4) __builtin_memcpy((void*)(dst + (n & 8u)), &w4, 4);
    if (n2) { const uint16_t t = (uint16_t)wq; __builtin_memcpy((void*)(dst + (n & 12u)), &t, 2); }
    if (n & 1u) dst[n & 14u] = (uint8_t)(wq >> (n2 ? 16 : 0));
}

// the compressed bytes for the generic sequence walker (lz4_pcd_common.h parse_seq): the staged tile from LDS, else memory
struct Reader {
    const lds_u8* tile;      // LDS copy of [t0 - TPAD, t0 + PT + TMARGIN)
    const g_u8* g;
    uint32_t t0;
    __device__ __forceinline__ uint32_t operator()(uint32_t pos) const {
        const uint32_t r = pos - t0;
        return r < PT + TMARGIN ? (uint32_t)tile[TPAD + r] : (uint32_t)g[pos];
    }
    __device__ __forceinline__ uint32_t u32(uint32_t pos) const { return (*this)(pos) | ((*this)(pos + 1u) << 8) | ((*this)(pos + 2u) << 16) | ((*this)(pos + 3u) << 24); }
};

struct Part {
    uint64_t marks;      // token positions of the standing walk, relative to the part's first byte
    uint32_t from;       // where the standing walk began (X_ERR: none)
    uint32_t exit;       // where its chain leaves the part: a position >= the part's end (>= ilen: the block ends or fails there), or X_ERR
};

// the generic walker's verdict as an exit position
__device__ __forceinline__ uint32_t as_pos(uint32_t nx, uint32_t ilen) { return nx == X_END ? ilen : nx; }

// One walk of a part from p.  FIRST: every position is marked.  Else: until the walk lands on a position the standing walk marked
// (its marks stand from there, and its exit) or leaves the part.
template <bool FIRST>
__device__ __forceinline__ void walk_part(const Reader& rd, uint32_t ilen, uint32_t p, uint32_t part0, uint32_t part_end, Part& s) {
    const uint32_t entry = p;
    uint64_t m2 = 0ull;
    uint32_t exit_ = X_ERR;
    bool merged = false;
    uint32_t prev = p;
    bool pm15 = false;                    // the previous sequence's match nibble was 15 and its length byte has not been looked at
    const lds_u8* tb = rd.tile + TPAD - rd.t0;     // tb + position (wraps; only ever indexed with positions of the staged range)
    for (;;) {
        if (p >= part_end) { exit_ = p; break; }
        const uint64_t bit = 1ull << (p - part0);
        if (!FIRST && (s.marks & bit) != 0ull) { merged = true; break; }
        // the bytes p - 1 .. p + 2 in one round trip
        const uint32_t w = lds_rd4u(tb, p - 1u);
        const uint32_t pb = w & 0xFFu, t = (w >> 8) & 0xFFu, e1 = (w >> 16) & 0xFFu;
        if (pm15 && pb == 255u) {         // the previous match length goes on: that sequence again, byte by byte
            pcd::Seq q;
            const uint32_t nx = pcd::parse_seq<Reader, false>(rd, ilen, prev, q);
            if (nx == X_ERR) break;
            p = as_pos(nx, ilen);
            pm15 = false;
            continue;
        }
        const uint32_t L = t >> 4, M = t & 15u;
        uint32_t nx;
        bool m15 = M == 15u;
        if (L == 15u && e1 == 255u) {     // more than one literal length byte
            pcd::Seq q;
            nx = pcd::parse_seq<Reader, false>(rd, ilen, p, q);
            if (nx == X_ERR) break;
            nx = as_pos(nx, ilen);
            m15 = false;
        } else {
            const uint32_t lit = L == 15u ? 15u + e1 : L;
            nx = p + (L == 15u ? 2u : 1u) + lit + 2u + (m15 ? 1u : 0u);
        }
        // a 15-nibble's length byte is checked by the next hop -- unless there is none: the walk leaves the part or meets the standing walk
        if (m15 && nx < ilen) {
            const bool leaving = nx >= part_end;
            const bool meeting = !FIRST && !leaving && ((s.marks >> (nx - part0)) & 1ull) != 0ull;
            if (leaving || meeting) {
                if (rd(nx - 1u) == 255u) {
                    pcd::Seq q;
                    nx = pcd::parse_seq<Reader, false>(rd, ilen, p, q);
                    if (nx == X_ERR) break;
                    nx = as_pos(nx, ilen);
                }
                m15 = false;
            }
        }
        m2 |= bit;
        prev = p;
        pm15 = m15;
        p = nx;
    }
    if (merged) { s.marks = m2 | (s.marks & ~((1ull << (p - part0)) - 1ull)); s.from = entry; }
    else { s.marks = m2; s.from = entry; s.exit = exit_; }
}

template <class G>
struct Dec {
    const g_u8* in;
    g_u8* out;
    lds_u8* tile;
    lds_u16* pos;
    lds_u8* win;
    uint32_t ilen, cap, lane;
    uint32_t OP;         // bytes produced
    uint32_t W0;         // the window holds [W0, OP), W0 a multiple of 16
    uint32_t F;          // bytes written back, a multiple of 16 (W0 <= F unless nothing slid yet)

    // ring -> output, whole 16-byte units
    __device__ __forceinline__ void write_back() {
        const uint32_t lim = OP & ~15u;
        for (uint32_t q = F + 16u * lane; q < lim; q += 1024u) {
            const u32x4 v = lds_rd16(win + (q - W0));
            __builtin_memcpy((void*)(out + q), &v, 16);
        }
        F = lim;
    }
    // make room for `need` (<= BUDGET) more bytes: write back, move the last KEEP bytes to the window's start
    __device__ __forceinline__ void ensure(uint32_t need) {
        if (OP - W0 + need <= G::R) return;
        write_back();
        const uint32_t nw = (OP - G::KEEP) & ~15u;          // (OP - W0 > KEEP here: need <= BUDGET)
        const uint32_t S = nw - W0, n = OP - nw;
        for (uint32_t i = 16u * lane; i < n; i += 1024u) {   // a lower iteration never writes what a higher one reads (S >= 0); within one, reads precede writes
            const u32x4 v = lds_rd16(win + S + i);
            lds_wr16(win + i, v);
        }
        W0 = nw;
    }
    __device__ __forceinline__ void finish() {
        write_back();
        if (F + lane < OP) out[F + lane] = win[F + lane - W0];
        F = OP;
    }
    // literals of any length from the compressed stream, whole wavefront, <= BUDGET bytes per piece
    __device__ __forceinline__ void coop_literals(uint32_t src, uint32_t n) {
        for (uint32_t c = 0u; c < n; c += G::BUDGET) {
            const uint32_t m = n - c < G::BUDGET ? n - c : G::BUDGET;
            ensure(m);
            for (uint32_t i = lane; i < m; i += 64u) win[OP - W0 + i] = in[src + c + i];
            OP += m;
        }
    }
    // a match of any offset / length, whole wavefront, 64 bytes per step.  Source bytes from the window when it holds them, else
    // from the output written back earlier.  offset < 64: the periodic form out[d + i] = out[d - offset + i mod offset]
    // (decompress.rs:57-82, decompress_safe.rs:301-318), which only reads bytes in front of the match.
    __device__ __forceinline__ void coop_match(uint32_t offset, uint32_t n) {
        const float rcp = offset < 64u ? 1.0f / (float)offset : 0.0f;
        for (uint32_t c = 0u; c < n; c += G::BUDGET) {
            const uint32_t m = n - c < G::BUDGET ? n - c : G::BUDGET;
            ensure(m);
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
                    const uint8_t byte = ps >= W0 ? win[ps - W0] : out[ps];
                    win[d - W0 + i] = byte;
                }
            }
            OP += m;
        }
    }
    // One sequence of any shape at `ip`, by the whole wavefront, with the reference's checks (decompress.rs:334-443; any violation ->
    // false: the reference-order kernel decodes the block again and names the error).  done: the block ended here.
    __device__ bool exact_seq(uint32_t ip, bool& done) {
        const uint32_t t = in[ip];
        ip += 1u;
        uint32_t lit = t >> 4;
        if (lit == 15u) {
            for (;;) {
                if (ip >= ilen) return false;
                const uint32_t b = in[ip];
                ip += 1u;
                lit += b;
                if (lit > 0x7FFFFFFFu) return false;      // (a 32-bit sum must not wrap: the reference counts in usize)
                if (b != 255u) break;
            }
        }
        if (lit > ilen - ip || lit > cap - OP) return false;
        coop_literals(ip, lit);
        ip += lit;
        if (ip >= ilen) { done = true; return true; }
        if (ilen - ip < 2u) return false;
        const uint32_t offset = (uint32_t)in[ip] | ((uint32_t)in[ip + 1u] << 8);
        ip += 2u;
        if (offset == 0u) return false;
        uint32_t ml = 4u + (t & 15u);
        if (ml == 19u) {
            for (;;) {
                if (ip >= ilen) return false;
                const uint32_t b = in[ip];
                ip += 1u;
                ml += b;
                if (ml > 0x7FFFFFFFu) return false;
                if (b != 255u) break;
            }
        }
        if (offset > OP || ml > cap - OP) return false;
        coop_match(offset, ml);
        if (ip >= ilen) return false;              // a match is always followed by another token (decompress.rs:439-443)
        return true;
    }
};

// The chunks of one tile: sequences [0, n_tile) of the token list (positions relative to t0).  false: the block is irregular.
template <class G>
__device__ __forceinline__ bool run_chunks(Dec<G>& D, uint32_t t0, uint32_t n_tile, bool& done) {
    const uint32_t lane = D.lane, ilen = D.ilen;
    const lds_u8* tb = D.tile + TPAD;            // tb + tile-relative position
    uint32_t sidx = 0u;
    while (sidx < n_tile) {
        const uint32_t idx = sidx + lane;
        bool act = idx < n_tile;
        const uint32_t tpr = act ? (uint32_t)D.pos[idx] : 0u;
        // ---- token, lengths, offset (decompress.rs:249-258, 284, 373-391) ----------------------------------------------------
        const uint32_t w = lds_rd4u(tb, tpr);
        const uint32_t t = w & 0xFFu, e1 = (w >> 8) & 0xFFu;
        const uint32_t L = t >> 4, M = t & 15u;
        const uint32_t lit = L == 15u ? 15u + e1 : L;
        const uint32_t lsr = tpr + (L == 15u ? 2u : 1u);          // literals, relative to t0
        bool big = (L == 15u && e1 == 255u) || lit > LITMAX;      // not a lane's work: exact_seq
        const uint32_t lend = big ? 0u : lsr + lit;
        const uint32_t w1 = lds_rd4u(tb, lend);
        const uint32_t off = w1 & 0xFFFFu, e2 = (w1 >> 16) & 0xFFu;
        const uint32_t mlx = 4u + M + (M == 15u ? e2 : 0u);
        big |= (M == 15u && e2 == 255u);
        const uint32_t nxt = lend + 2u + (M == 15u ? 1u : 0u);    // the next token, relative to t0
        const bool last = t0 + lend >= ilen;                      // the block's last sequence: literals only (:366-368) -- or an error
        const bool has_m = !last;
        bool err = t0 + lend > ilen;                              // :346-348
        err |= has_m && (t0 + lend + 2u > ilen || t0 + nxt >= ilen || off == 0u);      // :373-375, :439-443, :168-173
        big |= has_m && off < mlx;                                // a match that reads its own output: the periodic form
        const uint32_t ml = has_m ? mlx : 0u;
        // ---- the chunk: up to the first sequence that is not a lane's work, and at most BUDGET bytes ------------------------------
        const uint64_t bigm = ballot(act && big);
        uint32_t nact = n_tile - sidx < 64u ? n_tile - sidx : 64u;
        if (bigm != 0ull) { const uint32_t k = ctz64(bigm); nact = k < nact ? k : nact; }
        act = lane < nact;
        const uint32_t u = act ? lit + ml : 0u;
        const uint32_t incl = wave_incl_add(u);
        {
            const uint64_t over = ballot(act && incl > G::BUDGET);
            if (over != 0ull) { const uint32_t k = ctz64(over); nact = k < nact ? k : nact; act = lane < nact; }
        }
        uint32_t T = nact != 0u ? rdlane(incl, nact - 1u) : 0u;
        if (nact != 0u) {
            if (ballot(act && err) != 0ull) return false;
            if (T > D.cap - D.OP) return false;                    // OutputTooSmall (:349-356, :403-408): named by the reference-order kernel
            D.ensure(T);
        }
        const uint32_t dst = D.OP + incl - u, dm = dst + lit;
        const uint32_t src = dm - off;
        if (ballot(act && has_m && off > dm) != 0ull) return false;    // OffsetOutOfBounds (:286-289, :398-402)
        // a source in front of the window comes from the written-back output: all of it has to be there, and short enough for a lane
        const bool farl = act && has_m && src < D.W0;
        {
            const uint64_t fbig = ballot(farl && (src + ml > D.F || ml > FARMAX));
            if (fbig != 0ull) { const uint32_t k = ctz64(fbig); nact = k < nact ? k : nact; act = lane < nact; T = nact != 0u ? rdlane(incl, nact - 1u) : 0u; }
        }
        if (nact == 0u) {                                         // the first sequence alone, by the whole wavefront
            const uint32_t tp0 = t0 + rdlane(tpr, 0u);
            if (!D.exact_seq(tp0, done)) return false;
            sidx += 1u;
            if (done) return sidx == n_tile;
            continue;
        }
        const bool is_far = farl && act;
        const bool is_near = act && has_m && !is_far;
        // ---- far sources: requested now, used behind the literals ------------------------------------------------------------------
        u32x4 f0 = {0u, 0u, 0u, 0u}, f1 = f0, f2 = f0, f3 = f0;
        if (is_far) {
            __builtin_memcpy(&f0, (const void*)(D.out + src), 16);
            if (ml > 16u) __builtin_memcpy(&f1, (const void*)(D.out + src + ml - 16u), 16);
            if (ml > 32u) __builtin_memcpy(&f2, (const void*)(D.out + src + 16u), 16);
            if (ml > 48u) __builtin_memcpy(&f3, (const void*)(D.out + src + 32u), 16);
        }
        // ---- literals (decompress.rs:276-280, :357-361): 16 bytes per access, exact length ----------------------------------------
        lds_u8* wl = D.win + (dst - D.W0);
        const bool lp = act && lit != 0u;
        if (ballot(lp) != 0ull) {
            const lds_u8* la = tb + lsr;
            if (lp) {
                const u32x4 r1 = lds_rd16(la);
                if (lit <= 16u) write_exact16(wl, r1, lit);
                else {
                    const u32x4 r2 = lds_rd16(la + lit - 16u);
                    lds_wr16(wl, r1);
                    lds_wr16(wl + lit - 16u, r2);
                }
            }
            if (ballot(lp && lit > 32u) != 0ull) {
                for (uint32_t p = 16u; p < LITMAX - 16u; p += 16u)
                    if (lp && p + 16u < lit) { const u32x4 r = lds_rd16(la + p); lds_wr16(wl + p, r); }
            }
        }
        // ---- far matches ----------------------------------------------------------------------------------------------------------------
        lds_u8* wm = D.win + (dm - D.W0);
        if (is_far) {
            if (ml <= 16u) write_exact16(wm, f0, ml);
            else {
                lds_wr16(wm, f0);
                if (ml > 32u) lds_wr16(wm + 16u, f2);
                if (ml > 48u) lds_wr16(wm + 32u, f3);
                lds_wr16(wm + ml - 16u, f1);
            }
        }
        // ---- matches inside the window, in rounds: ready = every sequence that starts before the source's end is done ----------------
        uint64_t todo = ballot(is_near);
        const lds_u8* sa = D.win + (src - D.W0);
        const uint32_t s1 = src + ml;
        while (todo != 0ull) {
            const uint32_t dp = ctz64(todo);                      // every lane below dp is done: all bytes in front of its sequence are final
            const uint32_t S = rdlane(dst, dp);
            const bool ready = is_near && ((todo >> lane) & 1ull) != 0ull && (s1 <= S || lane == dp);
            if (ready) {
                const u32x4 r1 = lds_rd16(sa);
                if (ml <= 16u) write_exact16(wm, r1, ml);
                else {
                    const u32x4 r2 = lds_rd16(sa + ml - 16u);
                    lds_wr16(wm, r1);
                    lds_wr16(wm + ml - 16u, r2);
                }
            }
            const uint64_t rm = ballot(ready);
            if (ballot(ready && ml > 32u) != 0ull) {
                for (uint32_t p = 16u; ; p += 16u) {
                    const bool more = ready && p + 16u < ml;
                    if (ballot(more) == 0ull) break;
                    if (more) { const u32x4 r = lds_rd16(sa + p); lds_wr16(wm + p, r); }
                }
            }
            todo &= ~rm;
        }
        D.OP += T;
        sidx += nact;
        if (ballot(act && last) != 0ull) { done = true; return sidx == n_tile; }
    }
    return true;
}

template <class G>
__global__ void __launch_bounds__(64) lz4_decompress_seq_kernel(DecompressArgs a, int32_t redo_code) {
    extern __shared__ __attribute__((aligned(16))) uint8_t seq_lds[];
    const uint32_t lane = threadIdx.x;
    const uint32_t b = blockIdx.x;
    if (b >= a.n) return;
    Dec<G> D;
    D.in = (const g_u8*)(a.in_base + a.in_off[b]);
    D.out = (g_u8*)(a.out_base + a.out_off[b]);
    D.tile = (lds_u8*)seq_lds;
    D.pos = (lds_u16*)((lds_u8*)seq_lds + TILE_LDS);
    D.win = (lds_u8*)seq_lds + TILE_LDS + POS_LDS;
    D.ilen = a.in_len[b];
    D.cap = a.out_cap[b];
    D.lane = lane;
    D.OP = 0u; D.W0 = 0u; D.F = 0u;
    const uint32_t ilen = D.ilen;
    bool ok = ilen != 0u, done = false;          // (an empty block: decompress.rs:207-209, the reference-order kernel reports it)
    uint32_t entry = 0u;
    if (lane < 4u) ((lds_u32*)D.tile)[lane] = 0u;     // the bytes in front of the first tile
    while (ok && !done) {
        const uint32_t t0 = entry & ~15u;
        // ---- stage the tile: [t0, t0 + PT + TMARGIN), zeros behind the block --------------------------------------------------------
        __builtin_amdgcn_s_barrier();                      // (one wavefront: orders the LDS accesses of the previous tile)
        for (uint32_t o = 16u * lane; o < PT + TMARGIN; o += 1024u) {
            u32x4 v = {0u, 0u, 0u, 0u};
            const uint32_t g = t0 + o;
            if (g + 16u <= ilen) __builtin_memcpy(&v, (const void*)(D.in + g), 16);
            else if (g < ilen) {
                uint32_t wv[4] = {0u, 0u, 0u, 0u};
                for (uint32_t k = 0u; k < 16u; ++k) if (g + k < ilen) wv[k >> 2] |= (uint32_t)D.in[g + k] << (8u * (k & 3u));
                v = u32x4{wv[0], wv[1], wv[2], wv[3]};
            }
            lds_wr16(D.tile + TPAD + o, v);
        }
        __builtin_amdgcn_s_barrier();
        Reader rd;
        rd.tile = D.tile; rd.g = D.in; rd.t0 = t0;
        // ---- 1. first walks: lane 0 from the tile's entry, the others from their part's first byte ---------------------------------
        const uint32_t part0 = t0 + PB * lane;
        const uint32_t part_end = part0 + PB < ilen ? part0 + PB : ilen;
        const bool has_part = part0 < ilen;
        Part s;
        s.marks = 0ull; s.from = X_ERR; s.exit = X_ERR;
        if (has_part) walk_part<true>(rd, ilen, lane == 0u ? entry : part0, part0, part_end, s);
        // ---- 2. / 3. which parts does the true chain visit, and where does it enter them?  (lz4_decompress_plan.hip) ----------------
        uint32_t my_entry = X_ERR, tile_exit = X_ERR;
        for (uint32_t round = 0u; round < NPART + 2u; ++round) {
            const bool inside = s.exit != X_ERR && s.exit < t0 + PT && s.exit < ilen;
            const uint32_t nxt = inside ? ((s.exit - t0) * 2185u) >> 17 : 64u;          // / 60, exact below 4 096
            uint64_t reach = 1ull << lane;
            uint32_t jump = nxt;
#pragma unroll
            for (uint32_t i = 0u; i < 6u; ++i) {
                const uint32_t sl = jump < 64u ? jump : lane;
                const uint32_t rlo = bperm(sl, (uint32_t)reach), rhi = bperm(sl, (uint32_t)(reach >> 32)), j2 = bperm(sl, jump);
                if (jump < 64u) { reach |= ((uint64_t)rhi << 32) | rlo; jump = j2; }
            }
            const uint64_t path = ((uint64_t)rdlane((uint32_t)(reach >> 32), 0u) << 32) | rdlane((uint32_t)reach, 0u);
            const bool on_path = ((path >> lane) & 1ull) != 0ull;
            const uint64_t before = path & ((1ull << lane) - 1ull);
            const uint32_t pred = before != 0ull ? 63u - (uint32_t)__builtin_clzll(before) : lane;
            const uint32_t pulled = bperm(pred, s.exit);
            my_entry = lane == 0u ? entry : (on_path && before != 0ull ? pulled : X_ERR);
            const uint32_t lastp = 63u - (uint32_t)__builtin_clzll(path);
            tile_exit = rdlane(s.exit, lastp);
            const bool need = my_entry != X_ERR && s.from != my_entry;
            if (ballot(need) == 0ull) break;
            if (need) walk_part<false>(rd, ilen, my_entry, part0, part_end, s);
            tile_exit = X_ERR;                                       // (not final: the next pass says)
        }
        if (tile_exit == X_ERR) { ok = false; break; }
        // ---- 4. the token list ---------------------------------------------------------------------------------------------------------
        const bool live = my_entry != X_ERR;
        uint64_t m = live ? s.marks : 0ull;
        const uint32_t cnt = (uint32_t)__builtin_popcountll(m);
        const uint32_t cincl = wave_incl_add(cnt);
        const uint32_t n_tile = rdlane(cincl, 63u);
        if (n_tile > POSCAP || n_tile == 0u) { ok = false; break; }
        {
            uint32_t at = cincl - cnt;
            const uint32_t rel0 = PB * lane;
            while (ballot(m != 0ull) != 0ull) {
                if (m != 0ull) {
                    D.pos[at] = (uint16_t)(rel0 + ctz64(m));
                    at += 1u;
                    m &= m - 1ull;
                }
            }
        }
        __builtin_amdgcn_s_barrier();
        // ---- 5. the chunks ----------------------------------------------------------------------------------------------------------------
        bool tdone = false;
        if (!run_chunks<G>(D, t0, n_tile, tdone)) { ok = false; break; }
        if (tdone) { done = true; break; }
        if (tile_exit >= ilen) { ok = false; break; }       // the chain ran out without a last sequence
        entry = tile_exit;
        // (a long literal run or match length run jumps over tiles: the next tile starts where the chain goes on)
    }
    if (ok && done) {
        D.finish();
        if (lane == 0u) {
            a.status[b] = 0;
            a.out_len[b] = D.OP;
            if (a.detail) { a.detail[2u * b] = 0u; a.detail[2u * b + 1u] = 0u; }
        }
    } else if (lane == 0u) {
        a.status[b] = redo_code;          // decoded again, with the reference's check order, by lz4_decompress_blocks_kernel
        a.out_len[b] = 0u;
    }
}

}  // namespace sq

// Blocks without dictionary / prefix.  Irregular blocks get status `redo_code`; the caller runs launch_decompress with
// only_status = redo_code behind this launch.
hipError_t launch_decompress_seq(const DecompressArgs& a, int32_t redo_code, hipStream_t s) {
    if (a.n == 0u) return hipSuccess;
    if (a.dict_base != nullptr || a.out_pos != nullptr) return hipErrorInvalidValue;
    typedef sq::Geo<8192u> G;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)sq::lz4_decompress_seq_kernel<G>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS);
        attr_done = true;
    }
    hipLaunchKernelGGL(sq::lz4_decompress_seq_kernel<G>, dim3(a.n), dim3(64), G::LDS, s, a, redo_code);
    return hipGetLastError();
}

}  // namespace lz4flex_dev
