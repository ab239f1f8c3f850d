// lz4_compress_wave.hip -- the throughput LZ4 block encoder ("wave encoder") for gfx950.
//
// What it replaces: lz4_flex::block::compress_into (src/block/compress.rs:318-489) applied to many blocks.
// BASELINE.json north_star asks the compress side for "a valid LZ4 stream that the reference decodes to the
// identical input (ratio reported)", not for the reference's bytes, so this encoder has its own parse (the
// reference-exact encoder stays in lz4_compress.hip).  Block format rules honoured: MINMATCH 4, the last match
// starts >= 12 bytes before the end, the last 5 bytes are literals (src/block/mod.rs:37-61); token / length
// bytes as compress.rs:237-247,463-486.  tests/sim/wave_encoder_model.c is the scalar model of this file: the
// GPU tests require identical bytes.
//
// A CPU LZ4 encoder is one serial chain per block (probe -> verify -> extend -> next probe).  Round 1 kept that
// chain and ran 16 of them per CU; a chain step was three dependent global round trips.  Here nothing serial
// touches memory:
//   * a persistent workgroup (9 wavefronts) owns one 64 KiB window at a time, the window lives in LDS;
//   * wavefront 8 ("indexer") walks the NEXT window 64 positions per step through the 4096-entry hash table
//     (LDS, u16) and stores, for every position, the distance to the most recent earlier position with the same
//     hash (cand[], 2 B per position, in an L2-resident slot of the workgroup's workspace);
//   * wavefronts 0..7 ("workers") each own an 8 KiB segment of the CURRENT window.  Per step of 64 positions:
//     the lanes whose candidate distance differs from their predecessor's ("heads") count their true match
//     length against the LDS window (16 B per iteration, all heads of the step at once); a DPP prefix maximum
//     gives every position the match that reaches furthest; one-step lazy evaluation is a lane compare; the
//     greedy walk over the step is scalar (ballot, s_ff1, v_readlane) and touches no memory; the selected
//     sequences are encoded lane-parallel into an LDS staging buffer and flushed 16 B per lane;
//   * segments are independent parses (a match never crosses a segment end) that may reference the whole window;
//     after a barrier every worker places its segment's bytes: the literals left over at a segment's end are
//     carried into the first sequence of the next segment (its token is written at that point).
// HBM traffic: the input once, the output once; cand[] and the segment bodies stay in L2 / Infinity Cache
// (329 728 B of workspace per workgroup, 512 workgroups).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lz4_device.h"

namespace lz4flex_dev {
namespace wave {

typedef __attribute__((address_space(3))) uint8_t lds_u8;
typedef __attribute__((address_space(3))) uint16_t lds_u16;
typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr uint32_t WINDOW = 65536u;
constexpr uint32_t SEG = 8192u;
constexpr uint32_t WORKERS = 8u;          // WINDOW / SEG
constexpr uint32_t CAP = 1024u;           // longest match a head counts
constexpr uint32_t SKIPD = 64u;           // a position buried this deep in a running match is not evaluated
constexpr uint32_t HBITS = 12u;
constexpr uint32_t THREADS = 64u * (WORKERS + 1u);
constexpr uint32_t STG_BYTES = 768u;      // per worker: encoded sequences waiting for a 16 B-per-lane flush
constexpr uint32_t FLUSH_AT = 384u;
// LDS layout (80 192 B: two workgroups per CU)
constexpr uint32_t L_WIN = 0u;                              // the window + 64 B of slack for the 16-byte compares
constexpr uint32_t L_TAB = WINDOW + 64u;                    // the indexer's table, 4096 x u16
constexpr uint32_t L_STG = L_TAB + (2u << HBITS);
constexpr uint32_t L_META = L_STG + WORKERS * STG_BYTES;
constexpr uint32_t LDS_BYTES = L_META + 256u;
// workspace per workgroup
constexpr uint32_t SLOT_BYTES = 2u * WINDOW;                // cand[] of one window
constexpr uint32_t BODY_STRIDE = SEG + 256u;                // a segment's encoded bytes never exceed SEG + SEG/255 + 16
constexpr uint32_t WS_BYTES = 2u * SLOT_BYTES + WORKERS * BODY_STRIDE;
static_assert(WS_BYTES % 256u == 0u, "workspace slots stay 256 B aligned");

struct SegMeta {          // LDS, one per worker, valid between the two barriers of a window
    uint32_t has;         // the segment holds at least one match
    uint32_t first_lit;   // literals from the segment start to its first match
    uint32_t first_ml;    // that match's length (its token is written when the segments are placed)
    uint32_t trail;       // literals after the last match
    uint32_t body_len;    // bytes in the body: [offset, ml-ext] of the first sequence, then whole sequences
};
struct BlkCarry { uint32_t out_pos, pend; };

__device__ __forceinline__ uint32_t dpp_wave_shr1(uint32_t v, uint32_t lane0) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)lane0, (int)v, 0x138, 0xf, 0xf, false);
}
__device__ __forceinline__ uint32_t dpp_wave_shl1(uint32_t v, uint32_t lane63) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)lane63, (int)v, 0x130, 0xf, 0xf, false);
}
#define LZ4W_DPP(v, ctrl, rmask) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), (ctrl), (rmask), 0xf, false))
__device__ __forceinline__ uint32_t wave_incl_max(uint32_t v) {
    uint32_t t;
    t = LZ4W_DPP(v, 0x111, 0xf); v = v > t ? v : t;     // row_shr:1
    t = LZ4W_DPP(v, 0x112, 0xf); v = v > t ? v : t;     // row_shr:2
    t = LZ4W_DPP(v, 0x114, 0xf); v = v > t ? v : t;     // row_shr:4
    t = LZ4W_DPP(v, 0x118, 0xf); v = v > t ? v : t;     // row_shr:8
    t = LZ4W_DPP(v, 0x142, 0xa); v = v > t ? v : t;     // row_bcast:15 -> rows 1, 3
    t = LZ4W_DPP(v, 0x143, 0xc); v = v > t ? v : t;     // row_bcast:31 -> rows 2, 3
    return v;
}
__device__ __forceinline__ uint32_t wave_incl_add(uint32_t v) {
    v += LZ4W_DPP(v, 0x111, 0xf);
    v += LZ4W_DPP(v, 0x112, 0xf);
    v += LZ4W_DPP(v, 0x114, 0xf);
    v += LZ4W_DPP(v, 0x118, 0xf);
    v += LZ4W_DPP(v, 0x142, 0xa);
    v += LZ4W_DPP(v, 0x143, 0xc);
    return v;
}
__device__ __forceinline__ uint32_t rdlane(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint64_t ld64l(const lds_u8* p) { uint64_t v; __builtin_memcpy(&v, (const void*)p, 8); return v; }
__device__ __forceinline__ uint32_t ctz64(uint64_t x) { return (uint32_t)__builtin_ctzll(x); }
__device__ __forceinline__ uint32_t len_ext_bytes(uint32_t v) { return v >= 15u ? (v - 15u) / 255u + 1u : 0u; }   // compress.rs:237-247

// ---- indexer --------------------------------------------------------------------------------------------------
// cand[p] = distance from p to the most recent earlier position of the window with the same 4-byte hash (0: none).
// All 64 lookups of a step precede its 64 inserts (LDS operations of a wavefront execute in order); when lanes of a
// step share a bucket the highest one stays (checked on the device by tests/test_gpu_wave_encoder.py).
__device__ void index_window(const uint8_t* __restrict__ gwin, uint32_t wl, uint32_t act_n, uint16_t* __restrict__ slot,
                             lds_u8* lds, uint32_t lane) {
    lds_u16* tab = (lds_u16*)(lds + L_TAB);
    {
        lds_u32* t4 = (lds_u32*)(lds + L_TAB);
        for (uint32_t i = lane; i < (2u << HBITS) / 4u; i += 64u) t4[i] = 0u;
    }
    for (uint32_t b = 0; b < wl; b += 256u) {
        uint32_t v[4];
#pragma unroll
        for (uint32_t u = 0; u < 4u; ++u) {
            const uint32_t p = b + 64u * u + lane;
            uint32_t x = 0u;
            if (p < act_n) __builtin_memcpy(&x, gwin + p, 4);
            v[u] = x;
        }
#pragma unroll
        for (uint32_t u = 0; u < 4u; ++u) {
            const uint32_t p = b + 64u * u + lane;
            const bool act = p < act_n;
            const uint32_t h = (v[u] * 2654435761u) >> (32u - HBITS);
            uint32_t d = 0u;
            if (act) {
                const uint32_t e = tab[h];
                tab[h] = (uint16_t)p;
                d = (p - e) & 0xFFFFu;
            }
            if (p < wl) slot[p] = (uint16_t)d;
        }
    }
}

// ---- worker ---------------------------------------------------------------------------------------------------
struct Worker {
    lds_u8* win;
    lds_u8* stg;
    uint8_t* body;
    uint32_t lane;
    uint32_t fill, body_len;
    uint32_t has, first_lit, first_ml;

    __device__ __forceinline__ void flush(bool all) {
        const uint32_t n16 = fill & ~15u;
        for (uint32_t i = 16u * lane; i < n16; i += 1024u) {
            const u32x4 v = *reinterpret_cast<const __attribute__((address_space(3))) u32x4*>(stg + i);
            *reinterpret_cast<u32x4*>(body + body_len + i) = v;
        }
        const uint32_t rem = fill - n16;
        if (all) {
            if (lane < rem) body[body_len + n16 + lane] = stg[n16 + lane];
            body_len += fill;
            fill = 0u;
        } else {
            uint8_t t = 0;
            if (lane < rem) t = stg[n16 + lane];
            if (n16 != 0u && lane < rem) stg[lane] = t;     // n16 >= 16 > rem: source and destination do not overlap
            body_len += n16;
            fill = rem;
        }
    }
    __device__ __forceinline__ void room(uint32_t n) {       // n <= 256
        if (fill + n > STG_BYTES) flush(false);
    }
    // one sequence through the generic path: any literal count, any match length
    __device__ void emit_generic(uint32_t lit_src, uint32_t lit, uint32_t off, uint32_t ml) {
        const uint32_t mlc = ml - 4u;
        if (!has) {
            has = 1u; first_lit = lit; first_ml = ml;
        } else {
            // token + literal length bytes
            const uint32_t ne = len_ext_bytes(lit);
            const uint32_t hdr = 1u + ne;
            for (uint32_t c0 = 0; c0 < hdr; c0 += 256u) {
                const uint32_t cn = hdr - c0 < 256u ? hdr - c0 : 256u;
                room(cn);
                for (uint32_t i = lane; i < cn; i += 64u) {
                    const uint32_t j = c0 + i;
                    uint32_t byte;
                    if (j == 0u) byte = ((lit < 15u ? lit : 15u) << 4) | (mlc < 15u ? mlc : 15u);
                    else byte = (j < ne) ? 255u : (lit - 15u) % 255u;
                    stg[fill + i] = (uint8_t)byte;
                }
                fill += cn;
            }
            for (uint32_t c0 = 0; c0 < lit; c0 += 256u) {
                const uint32_t cn = lit - c0 < 256u ? lit - c0 : 256u;
                room(cn);
                for (uint32_t i = lane; i < cn; i += 64u) stg[fill + i] = win[lit_src + c0 + i];
                fill += cn;
            }
        }
        const uint32_t me = len_ext_bytes(mlc);
        room(2u + me);                                      // CAP 1024: me <= 4
        if (lane < 2u + me) {
            uint32_t byte;
            if (lane == 0u) byte = off & 255u;
            else if (lane == 1u) byte = off >> 8;
            else byte = (lane - 2u + 1u < me) ? 255u : (mlc - 15u) % 255u;
            stg[fill + lane] = (uint8_t)byte;
        }
        fill += 2u + me;
    }
};

__device__ __forceinline__ void copy_lit_small(lds_u8* dst, const lds_u8* src, uint32_t n) {   // n < 16, exact
    if (n & 8u) { uint64_t t; __builtin_memcpy(&t, (const void*)src, 8); __builtin_memcpy((void*)dst, &t, 8); src += 8; dst += 8; }
    if (n & 4u) { uint32_t t; __builtin_memcpy(&t, (const void*)src, 4); __builtin_memcpy((void*)dst, &t, 4); src += 4; dst += 4; }
    if (n & 2u) { uint16_t t; __builtin_memcpy(&t, (const void*)src, 2); __builtin_memcpy((void*)dst, &t, 2); src += 2; dst += 2; }
    if (n & 1u) { *dst = *src; }
}

// One segment [s0, s1) of the window in LDS.  mfl: positions p < mfl_end may start a match (p <= n - 12);
// mend: matches end here at the latest (segment end, block end - 5, 65535).
__device__ void match_segment(lds_u8* lds, const uint16_t* __restrict__ cand, uint8_t* body, uint32_t w, uint32_t lane,
                              uint32_t s0, uint32_t s1, uint32_t mfl_end, uint32_t mend) {
    Worker W;
    W.win = lds + L_WIN;
    W.stg = lds + L_STG + w * STG_BYTES;
    W.body = body;
    W.lane = lane;
    W.fill = 0u; W.body_len = 0u; W.has = 0u; W.first_lit = 0u; W.first_ml = 0u;
    uint32_t cursor = s0, anchor = s0, carry = 0u, dlast = 0u;
    uint32_t d_next = 0u;
    if (s0 + lane < s1) d_next = cand[s0 + lane];
    for (uint32_t b = s0; b < s1; b += 64u) {
        const uint32_t p = b + lane;
        const uint32_t d = d_next;
        d_next = 0u;
        if (p + 64u < s1) d_next = cand[p + 64u];
        // heads
        const uint32_t dprev = dpp_wave_shr1(d, dlast);
        dlast = rdlane(d, 63u);
        const uint32_t cend = carry >> 16;
        bool head = (p < s1) && (p < mfl_end) && d != 0u && d != dprev && d <= p;
        head = head && !(cend > p && cend - p >= SKIPD);
        uint32_t lim = mend > p ? mend - p : 0u;
        lim = lim < CAP ? lim : CAP;
        uint32_t k = 0u;
        bool act = head && lim >= 4u;
        while (__ballot(act) != 0ull) {
            if (act) {
                const lds_u8* a = W.win + p + k;
                const lds_u8* c = a - d;
                const uint64_t x0 = ld64l(a) ^ ld64l(c);
                const uint64_t x1 = ld64l(a + 8) ^ ld64l(c + 8);
                if (x0 != 0ull) { k += ctz64(x0) >> 3; act = false; }
                else if (x1 != 0ull) { k += 8u + (ctz64(x1) >> 3); act = false; }
                else { k += 16u; act = k < lim; }
            }
        }
        k = k < lim ? k : lim;
        const uint32_t own = (head && k >= 4u) ? (((p + k) << 16) | d) : 0u;
        // the match that reaches furthest, from any head at or before this position
        uint32_t best = wave_incl_max(own);
        best = best > carry ? best : carry;
        carry = rdlane(best, 63u);
        const uint32_t e = best >> 16;
        const uint32_t e_next = dpp_wave_shl1(e, 0u);
        bool elig = (p < s1) && (p < mfl_end) && (e >= p + 4u);
        elig = elig && !(lane < 63u && p + 1u < s1 && e_next > e + 1u);
        const uint64_t em = __ballot(elig);
        // greedy walk (scalar)
        uint64_t sel = 0ull;
        const uint32_t anchor_in = anchor;
        while (cursor < b + 64u) {
            const uint32_t c = cursor > b ? cursor - b : 0u;
            const uint64_t m = em & (~0ull << c);
            if (m == 0ull) break;
            const uint32_t q = ctz64(m);
            sel |= 1ull << q;
            cursor = anchor = rdlane(e, q);
        }
        if (sel == 0ull) continue;
        // encode the selected sequences
        const bool issel = (sel >> lane) & 1ull;
        const uint32_t len = e - p;
        const uint32_t off = best & 0xFFFFu;
        uint32_t pe = wave_incl_max(issel ? e : 0u);
        pe = dpp_wave_shr1(pe, 0u);
        pe = pe > anchor_in ? pe : anchor_in;             // end of the previous sequence
        const uint32_t lit = p - pe;
        const uint32_t mlc = len - 4u;
        const bool simple = !issel || (lit < 15u && mlc < 270u);
        if (__ballot(!simple) == 0ull) {
            const uint32_t fl = ctz64(sel);
            const bool first = !W.has && lane == fl;
            const uint32_t size = issel ? ((first ? 2u : 3u + lit) + (mlc >= 15u ? 1u : 0u)) : 0u;
            const uint32_t incl = wave_incl_add(size);
            const uint32_t total = rdlane(incl, 63u);
            if (issel) {
                lds_u8* o = W.stg + W.fill + incl - size;
                if (!first) {
                    o[0] = (uint8_t)((lit << 4) | (mlc < 15u ? mlc : 15u));
                    copy_lit_small(o + 1, W.win + pe, lit);
                    o += 1u + lit;
                }
                o[0] = (uint8_t)off;
                o[1] = (uint8_t)(off >> 8);
                if (mlc >= 15u) o[2] = (uint8_t)(mlc - 15u);
            }
            if (!W.has) { W.has = 1u; W.first_lit = rdlane(lit, fl); W.first_ml = rdlane(len, fl); }
            W.fill += total;
        } else {
            uint64_t m = sel;
            while (m != 0ull) {
                const uint32_t q = ctz64(m);
                m &= m - 1ull;
                W.emit_generic(rdlane(pe, q), rdlane(lit, q), rdlane(off, q), rdlane(len, q));
            }
        }
        if (W.fill >= FLUSH_AT) W.flush(false);
    }
    W.flush(true);
    if (lane == 0u) {
        SegMeta* M = (SegMeta*)nullptr;
        (void)M;
        lds_u32* mp = (lds_u32*)(lds + L_META) + 5u * w;
        mp[0] = W.has; mp[1] = W.first_lit; mp[2] = W.first_ml; mp[3] = s1 - anchor; mp[4] = W.body_len;
    }
}

// bytes [0, n) with a per-byte generator; 64 lanes
template <typename F>
__device__ __forceinline__ void put_bytes(uint8_t* dst, uint32_t n, uint32_t lane, F f) {
    for (uint32_t i = lane; i < n; i += 64u) dst[i] = (uint8_t)f(i);
}
__device__ __forceinline__ void put_len_header(uint8_t* dst, uint32_t lit, uint32_t ml_nibble, uint32_t lane) {
    const uint32_t ne = len_ext_bytes(lit);
    put_bytes(dst, 1u + ne, lane, [&](uint32_t j) -> uint32_t {
        if (j == 0u) return ((lit < 15u ? lit : 15u) << 4) | ml_nibble;
        return (j < ne) ? 255u : (lit - 15u) % 255u;
    });
}

// Place segment w of the current window (after the barrier: every worker's SegMeta is final).
__device__ void place_segment(lds_u8* lds, const uint8_t* __restrict__ gin, uint32_t blk_len, uint32_t win_idx, bool last_win,
                              uint32_t wl, const uint8_t* body, uint8_t* gout, uint32_t carry_slot, uint32_t w, uint32_t lane,
                              uint32_t* out_len, int32_t* status) {
    const lds_u32* mp = (const lds_u32*)(lds + L_META);
    lds_u32* cp = (lds_u32*)(lds + L_META) + 5u * WORKERS;          // BlkCarry[2]
    uint32_t out_pos = 0u, pend = 0u;
    if (win_idx != 0u) { out_pos = cp[2u * (carry_slot ^ 1u)]; pend = cp[2u * (carry_slot ^ 1u) + 1u]; }
    for (uint32_t j = 0; j < w; ++j) {
        const uint32_t sl = wl > j * SEG ? (wl - j * SEG < SEG ? wl - j * SEG : SEG) : 0u;
        if (mp[5u * j] != 0u) {
            const uint32_t L = pend + mp[5u * j + 1u];
            out_pos += 1u + len_ext_bytes(L) + L + mp[5u * j + 4u];
            pend = mp[5u * j + 3u];
        } else {
            pend += sl;
        }
    }
    const uint32_t sl = wl > w * SEG ? (wl - w * SEG < SEG ? wl - w * SEG : SEG) : 0u;
    const uint32_t abs0 = win_idx * WINDOW + w * SEG;              // block-relative start of this segment
    if (mp[5u * w] != 0u) {
        const uint32_t fl = mp[5u * w + 1u], L = pend + fl, ml = mp[5u * w + 2u] - 4u, bl = mp[5u * w + 4u];
        put_len_header(gout + out_pos, L, ml < 15u ? ml : 15u, lane);
        out_pos += 1u + len_ext_bytes(L);
        const uint8_t* src = gin + (abs0 + fl - L);
        for (uint32_t i = lane; i < L; i += 64u) gout[out_pos + i] = src[i];
        out_pos += L;
        for (uint32_t i = lane; i < bl; i += 64u) gout[out_pos + i] = body[i];
        out_pos += bl;
        pend = mp[5u * w + 3u];
    } else {
        pend += sl;
    }
    if (w == WORKERS - 1u) {
        if (last_win) {
            // the block's last literals (compress.rs handle_last_literals): token, length bytes, bytes; no offset
            put_len_header(gout + out_pos, pend, 0u, lane);
            out_pos += 1u + len_ext_bytes(pend);
            const uint8_t* src = gin + (blk_len - pend);
            for (uint32_t i = lane; i < pend; i += 64u) gout[out_pos + i] = src[i];
            out_pos += pend;
            if (lane == 0u) { *out_len = out_pos; *status = 0; }
        } else if (lane == 0u) {
            cp[2u * carry_slot] = out_pos;
            cp[2u * carry_slot + 1u] = pend;
        }
    }
}

struct Item {
    uint32_t blk, win, nwin, len, skip;
    uint64_t in_off;
};

__device__ __forceinline__ void item_load(const CompressArgs& a, Item& it) {
    // first window of block it.blk (or invalid)
    it.win = 0u; it.nwin = 0u; it.len = 0u; it.skip = 0u; it.in_off = 0ull;
    if (it.blk >= a.n) return;
    const uint32_t len = a.in_len[it.blk];
    const uint32_t cap = a.out_cap[it.blk];
    it.len = len;
    it.in_off = a.in_off[it.blk];
    it.nwin = len == 0u ? 1u : (uint32_t)(((uint64_t)len + WINDOW - 1u) / WINDOW);
    const uint64_t need = 20ull + (uint64_t)len * 110ull / 100ull;   // get_maximum_output_size, compress.rs:588-590
    if ((uint64_t)cap < need) { it.skip = 1u; it.nwin = 1u; }
}
__device__ __forceinline__ void item_next(const CompressArgs& a, Item& it) {
    if (it.win + 1u < it.nwin) { it.win += 1u; return; }
    it.blk += gridDim.x;
    item_load(a, it);
}

__global__ void __launch_bounds__(THREADS) lz4_compress_wave_kernel(const CompressArgs a, uint8_t* __restrict__ ws) {
    extern __shared__ __attribute__((aligned(16))) uint8_t dyn_lds[];
    lds_u8* lds = (lds_u8*)dyn_lds;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t w = uni(threadIdx.x >> 6);
    uint8_t* my_ws = ws + (size_t)blockIdx.x * WS_BYTES;
    uint16_t* slots = (uint16_t*)my_ws;                           // two cand[] slots
    uint8_t* bodies = my_ws + 2u * SLOT_BYTES;

    Item it;
    it.blk = blockIdx.x;
    item_load(a, it);
    if (it.blk >= a.n) return;
    Item ix = it;                                                 // the indexer runs one window ahead
    uint32_t k = 0u;

    auto win_len = [](const Item& t) -> uint32_t {
        const uint32_t base = t.win * WINDOW;
        return t.len > base ? (t.len - base < WINDOW ? t.len - base : WINDOW) : 0u;
    };
    auto do_index = [&](const Item& t, uint32_t slot) {
        if (t.skip) return;
        const uint32_t base = t.win * WINDOW, wl = win_len(t);
        const uint32_t act_abs = t.len >= 12u ? t.len - 11u : 0u;            // positions p < act_abs start 4 readable bytes and may match
        const uint32_t act_n = act_abs > base ? (act_abs - base < wl ? act_abs - base : wl) : 0u;
        index_window(a.in_base + t.in_off + base, wl, act_n, slots + (size_t)slot * WINDOW, lds, lane);
    };
    auto do_load = [&](const Item& t) {
        if (t.skip) return;
        const uint32_t wl = win_len(t);
        const uint8_t* g = a.in_base + t.in_off + (size_t)t.win * WINDOW;
        const uint32_t tid = threadIdx.x;                                       // 512 worker threads
        const uint32_t mis = (uint32_t)((16u - ((uintptr_t)g & 15u)) & 15u);    // bytes up to the first 16 B boundary
        const uint32_t head = mis < wl ? mis : wl;
        if (tid < head) lds[L_WIN + tid] = g[tid];
        const uint32_t nvec = (wl - head) / 16u;
        for (uint32_t i = tid; i < nvec; i += 512u) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(g + head + 16u * i);
            __builtin_memcpy((void*)(lds + L_WIN + head + 16u * i), &v, 16);
        }
        const uint32_t done = head + 16u * nvec;
        if (tid < wl - done) lds[L_WIN + done + tid] = g[done + tid];
        if (tid < 64u) lds[L_WIN + wl + tid] = 0;                               // slack read by the 16-byte compares
    };

    // prologue: cand[] of the first window, the first window into LDS
    if (w == WORKERS) { do_index(ix, 0u); item_next(a, ix); }
    else do_load(it);
    __syncthreads();
    for (;;) {
        const uint32_t wl = win_len(it);
        const bool last_win = it.win + 1u == it.nwin;
        if (w == WORKERS) {
            if (ix.blk < a.n) { do_index(ix, (k + 1u) & 1u); item_next(a, ix); }
        } else if (!it.skip) {
            const uint32_t base = it.win * WINDOW;
            const uint32_t s0 = w * SEG < wl ? w * SEG : wl;
            const uint32_t s1 = (w + 1u) * SEG < wl ? (w + 1u) * SEG : wl;
            const uint32_t act_abs = it.len >= 12u ? it.len - 11u : 0u;
            const uint32_t mfl_end = act_abs > base ? act_abs - base : 0u;      // window-relative, may exceed wl
            uint32_t mend = it.len >= 5u ? it.len - 5u : 0u;                    // block-relative
            mend = mend > base ? mend - base : 0u;
            mend = mend < s1 ? mend : s1;
            mend = mend < 65535u ? mend : 65535u;
            match_segment(lds, slots + (size_t)(k & 1u) * WINDOW, bodies + (size_t)w * BODY_STRIDE, w, lane, s0, s1, mfl_end, mend);
        }
        __syncthreads();
        if (w != WORKERS) {
            if (it.skip) {
                if (threadIdx.x == 0u) { a.out_len[it.blk] = 0u; a.status[it.blk] = LZ4FLEX_DEV_E_OUTPUT_TOO_SMALL; }
            } else {
                place_segment(lds, a.in_base + it.in_off, it.len, it.win, last_win, wl, bodies + (size_t)w * BODY_STRIDE,
                              a.out_base + a.out_off[it.blk], k & 1u, w, lane, a.out_len + it.blk, a.status + it.blk);
            }
        }
        item_next(a, it);
        k += 1u;
        if (it.blk >= a.n) break;
        if (w != WORKERS) do_load(it);
        __syncthreads();
    }
}

}  // namespace wave

size_t compress_wave_workspace_bytes(int n_workgroups) { return (size_t)n_workgroups * wave::WS_BYTES; }

hipError_t launch_compress_wave(const CompressArgs& a, void* workspace, int n_workgroups, hipStream_t s) {
    if (a.n == 0u) return hipSuccess;
    if (!workspace || n_workgroups <= 0) return hipErrorInvalidValue;
    static unsigned long long have = 0ull;
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(have & bit)) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wave::lz4_compress_wave_kernel),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)wave::LDS_BYTES);
        if (e != hipSuccess) return e;
        have |= bit;
    }
    const uint32_t grid = a.n < (uint32_t)n_workgroups ? a.n : (uint32_t)n_workgroups;
    hipLaunchKernelGGL(wave::lz4_compress_wave_kernel, dim3(grid), dim3(wave::THREADS), wave::LDS_BYTES, s, a, (uint8_t*)workspace);
    return hipGetLastError();
}

}  // namespace lz4flex_dev
