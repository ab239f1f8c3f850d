// lz4_decompress_wave.hip -- batched LZ4 block decoder, one block per wavefront ("wave decoder"), gfx950.
//
// What it replaces: lz4_flex::block::decompress_into / decompress_internal (src/block/decompress.rs:201-449) for many
// independent blocks.  Output bytes are the reference's; every irregular block (any error of src/block/mod.rs:82-98, a
// sink too small, an offset behind the output) is NOT diagnosed here: the block is marked and the reference-order
// decoder of lz4_decompress.hip decodes it again and reports the exact error variant and detail.
//
// Round 1 ran the reference's token loop as one serial chain per block (a lane per block parsing, eight lanes copying):
// 64 chains per CU, ~1 350 cycles per sequence, and a 4 MiB block took as long as 64 small ones.  Here a block is
// decoded by a whole wavefront and the 64 lanes are the parallelism INSIDE the block:
//   * parse: a window of 64 compressed bytes at a time, lane i assuming a token at byte i (two small loads: token +
//     first length byte, then the offset / match-length byte behind its literals); the real token chain through the
//     window is a scalar hop over v_readlane (~14 sequences per window, ~5 instructions each);
//   * a DPP prefix sum of literal + match lengths places every sequence of the window in the output at once;
//   * copies: the window's literals (<= 16 bytes each) and every match whose source lies before the window's own
//     output are written lane-parallel (lane = sequence: 16-byte unaligned LDS reads, exact-length writes), matches
//     whose source is older than the LDS ring come from the already written output (requested before the other copies,
//     consumed after them); only matches that read bytes produced inside the window, ring wrap-arounds and long runs go
//     through a serial loop in which the whole wavefront copies one sequence, 64 bytes per step;
//   * the output lives in an 8 KiB LDS ring per wavefront (20 wavefronts per CU), written back 16 bytes per lane.
// Anything the speculative parse cannot express (length bytes beyond one, runs longer than 2 KiB) is decoded by
// exact_token(): the same work done by the wavefront for ONE sequence of any length.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lz4_device.h"

namespace lz4flex_dev {
namespace wdec {

typedef __attribute__((address_space(3))) uint8_t lds_u8;
typedef __attribute__((address_space(1))) uint8_t g_u8;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) u32x4 g_u32x4;

constexpr uint32_t RB = 8192u, RM = RB - 1u;      // LDS ring bytes per wavefront
constexpr uint32_t IB = 1024u;                    // LDS copy of the compressed stream around the parse position (a window reads < 352 bytes ahead)
constexpr uint32_t IB_PAD = 32u;                  // bytes readable behind it (16-byte literal reads)
constexpr uint32_t WAVE_LDS = RB + IB + IB_PAD;
constexpr uint32_t WPB = 4u;                      // wavefronts (= blocks) per workgroup
constexpr uint32_t TMAX = 2048u;                  // output bytes of one window
constexpr uint32_t FLUSH_AT = 512u;               // write back when this much is pending

#ifdef LZ4D_PROF      // tools: cycles of wavefront 0 of every workgroup per part of a window -> g_wdec_prof[0..7], windows in [8]
__device__ unsigned long long g_wdec_prof[16];
#define LZ4D_T0 uint64_t pt_ = __builtin_readcyclecounter();
#define LZ4D_TICK(i) { const uint64_t t_ = __builtin_readcyclecounter(); if (D.lane == 0u) atomicAdd(&g_wdec_prof[i], (unsigned long long)(t_ - pt_)); pt_ = __builtin_readcyclecounter(); }
#else
#define LZ4D_T0
#define LZ4D_TICK(i)
#endif

#define LZ4D_DPP(v, ctrl, rmask) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), (ctrl), (rmask), 0xf, false))
__device__ __forceinline__ uint32_t wave_incl_add(uint32_t v) {
    v += LZ4D_DPP(v, 0x111, 0xf);
    v += LZ4D_DPP(v, 0x112, 0xf);
    v += LZ4D_DPP(v, 0x114, 0xf);
    v += LZ4D_DPP(v, 0x118, 0xf);
    v += LZ4D_DPP(v, 0x142, 0xa);
    v += LZ4D_DPP(v, 0x143, 0xc);
    return v;
}
__device__ __forceinline__ uint32_t rdlane(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint32_t ctz64(uint64_t x) { return (uint32_t)__builtin_ctzll(x); }

// n (1..16) bytes of v to LDS, exactly
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

struct Dec {
    const g_u8* in;
    g_u8* out;
    lds_u8* ring;
    lds_u8* ibuf;            // compressed bytes [ib0, ib0 + IB) (zero beyond the block)
    uint32_t ilen, cap, lane;
    uint32_t op, F;          // bytes produced; bytes written back (a multiple of 16)
    uint32_t ib0;            // a multiple of 16

    // make [pos, pos + IB / 2) readable from ibuf: one coalesced reload of the whole buffer when pos has moved past its
    // first half (the compressed stream is read about twice from L2, never byte-wise from the parse chain)
    __device__ __forceinline__ void input_at(uint32_t pos) {
        if (pos - ib0 < IB / 2u) return;
        ib0 = pos & ~15u;
        for (uint32_t i = 16u * lane; i < IB + IB_PAD; i += 1024u) {
            u32x4 v = {0u, 0u, 0u, 0u};
            const uint32_t g = ib0 + i;
            if (g + 16u <= ilen) __builtin_memcpy(&v, (const void*)(in + g), 16);
            else if (g < ilen) {
                uint32_t w[4] = {0u, 0u, 0u, 0u};
                for (uint32_t k = 0; k < 16u; ++k) if (g + k < ilen) w[k >> 2] |= (uint32_t)in[g + k] << (8u * (k & 3u));
                v = u32x4{w[0], w[1], w[2], w[3]};
            }
            __builtin_memcpy((void*)(ibuf + i), &v, 16);
        }
    }
    __device__ __forceinline__ bool in_ibuf(uint32_t pos, uint32_t n) const { return pos >= ib0 && pos + n <= ib0 + IB + IB_PAD; }

    __device__ __forceinline__ void flush() {            // ring -> output, whole 16-byte units
        const uint32_t lim = op & ~15u;
        for (uint32_t pos = F + 16u * lane; pos < lim; pos += 1024u) {
            u32x4 v;
            __builtin_memcpy(&v, (const void*)(ring + (pos & RM)), 16);
            __builtin_memcpy((void*)(out + pos), &v, 16);
        }
        F = lim;
    }
    __device__ __forceinline__ void finish() {
        flush();
        if (F + lane < op) out[F + lane] = ring[(F + lane) & RM];
        F = op;
    }
    // literals of any length from the compressed stream, whole wavefront; the caller bounds n so that the ring keeps
    // everything not written back yet
    __device__ __forceinline__ void coop_literals(uint32_t src, uint32_t dst, uint32_t n) {
        for (uint32_t c = 0; c < n; c += 64u) {
            const uint32_t i = c + lane;
            if (i < n) ring[(dst + i) & RM] = in[src + i];
        }
    }
    // a match of any offset / length, whole wavefront, 64 bytes per step.  Source bytes come from the ring when it still
    // holds them (position >= near_lo), else from the output written back earlier.  offset < 64: the periodic form
    // out[d + i] = out[d - offset + i mod offset], which only reads bytes that precede the match.
    __device__ __forceinline__ void coop_match(uint32_t dst, uint32_t offset, uint32_t n, uint32_t near_lo) {
        const uint32_t src = dst - offset;
        const float rcp = offset < 64u ? 1.0f / (float)offset : 0.0f;
        for (uint32_t c = 0; c < n; c += 64u) {
            const uint32_t i = c + lane;
            if (i < n) {
                uint32_t si = i;
                if (offset < 64u) {
                    uint32_t q = (uint32_t)((float)i * rcp);
                    uint32_t r = i - q * offset;                  // q is off by at most one either way
                    r = (int32_t)r < 0 ? r + offset : r;
                    r = r >= offset ? r - offset : r;
                    si = r;
                }
                const uint32_t pos = src + si;
                uint8_t byte;
                if (pos >= near_lo) byte = ring[pos & RM];
                else byte = out[pos];
                ring[(dst + i) & RM] = byte;
            }
        }
    }
};

// One sequence of any shape at `ip`, decoded by the whole wavefront with the reference's checks (any violation -> false:
// the block is decoded again by the reference-order kernel, which reports the error).  *done: the block ended here.
__device__ bool exact_token(Dec& D, uint32_t& ip, bool& done) {
    const g_u8* in = D.in;
    const uint32_t ilen = D.ilen;
    uint32_t t = in[ip];
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
    if (lit > ilen - ip || lit > D.cap - D.op) return false;
    for (uint32_t c = 0; c < lit; c += 1024u) {
        const uint32_t n = lit - c < 1024u ? lit - c : 1024u;
        D.coop_literals(ip + c, D.op, n);
        D.op += n;
        if (D.op - D.F >= FLUSH_AT) D.flush();
    }
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
    if (offset > D.op || ml > D.cap - D.op) return false;
    for (uint32_t c = 0; c < ml; c += 1024u) {
        const uint32_t n = ml - c < 1024u ? ml - c : 1024u;
        const uint32_t wend = D.op + n;
        D.coop_match(D.op, offset, n, wend > RB ? wend - RB : 0u);
        D.op += n;
        if (D.op - D.F >= FLUSH_AT) D.flush();
    }
    if (ip >= ilen) return false;              // a match is always followed by another token (decompress.rs:439-443)
    return true;
}

// A parsed window: what its (up to 21) sequences are and where they go.  Everything that needs memory outside LDS -- the
// literal bytes, the sources of matches older than the ring -- is requested while the window is parsed, i.e. one window
// before it is executed.
struct Win {
    uint32_t lit, ls, mlen, offs, o;      // per lane (= per byte of the window; meaningful on token lanes)
    u32x4 lv, f0, f1;                     // the 16 literal bytes; 32 source bytes of a far match
    bool tk, mt, lpl, lpf, lpn;           // token lane; has a match; literals / far match / near match written lane-parallel
    uint32_t op0, T, near_lo, adv;        // uniform: output position before / bytes produced / ring horizon / input bytes consumed
    bool ok, done, stop_cx;               // uniform: regular so far; the block's last sequence is inside; exact_token comes next
};

// FARLOADS: request the sources of far matches here (the single-wavefront kernel: the window is executed by this wavefront
// after the previous one); false: the executing wavefront requests them itself (request_far), W.lpf = "far" only
template <bool FARLOADS>
__device__ __forceinline__ Win parse_window(Dec& D, uint32_t ip, uint32_t op0) {
    Win W;
    LZ4D_T0
    const uint32_t lane = D.lane, ilen = D.ilen;
    D.input_at(ip);
    LZ4D_TICK(0)
    const lds_u8* ib = D.ibuf - D.ib0;                // ib + position
    // ---- speculative parse: lane i takes byte ip + i for a token (bytes behind the block read as 0) ---------------------
    const uint32_t p = ip + lane;
    uint16_t h;
    __builtin_memcpy(&h, (const void*)(ib + p), 2);
    const uint32_t tok = h & 0xFFu, e1 = h >> 8;
    uint32_t lit = tok >> 4;
    const uint32_t mlc = tok & 15u;
    uint32_t lhdr = 1u;
    bool cx = p >= ilen;                               // cx: not expressible here (or simply wrong): exact_token decides
    if (lit == 15u) { lhdr = 2u; cx |= (p + 1u >= ilen) | (e1 == 255u); lit = 15u + e1; }
    const uint32_t ls = p + lhdr, le = ls + lit;
    cx |= le > ilen;
    const bool fin = !cx & (le == ilen);               // the block's last sequence: literals only
    const bool seq = !cx & !fin;
    cx |= seq & (le + 2u > ilen);
    uint32_t w4;
    __builtin_memcpy(&w4, (const void*)(ib + le), 4);
    __builtin_memcpy(&W.lv, (const void*)(ib + ls), 16);
    const uint32_t offs = w4 & 0xFFFFu, m1 = (w4 >> 16) & 0xFFu;
    uint32_t mlen = 4u + mlc, nx = le + 2u;
    if (mlc == 15u) { cx |= seq & ((le + 3u > ilen) | (m1 == 255u)); mlen += m1; nx += 1u; }
    cx |= seq & (nx >= ilen);                          // a match must be followed by another token: leave it to exact_token
    if (fin) { mlen = 0u; nx = ilen; }
    LZ4D_TICK(1)
    // ---- the real token chain through the window (scalar hop) ----------------------------------------------------------
    const uint64_t cxm = __ballot(cx), finm = __ballot(fin), stopm = cxm | finm;
    const uint32_t nrel = nx - ip;                     // >= 3: the window always advances
    uint64_t tokm = 0ull;
    uint32_t cur = 0u;
    W.done = false; W.stop_cx = false;
    while (cur < 64u && ((stopm >> cur) & 1ull) == 0ull) {       // four scalar instructions and a v_readlane per sequence
        tokm |= 1ull << cur;
        cur = rdlane(nrel, cur);
    }
    if (cur < 64u) {
        if ((cxm >> cur) & 1ull) W.stop_cx = true;
        else { W.done = true; tokm |= 1ull << cur; }
    }
    LZ4D_TICK(2)
    // ---- place the window's sequences ----------------------------------------------------------------------------------
    bool tk = (tokm >> lane) & 1ull;
    uint32_t tl = tk ? lit + mlen : 0u;
    uint32_t incl = wave_incl_add(tl);
    {
        const uint64_t big = __ballot(tk & (incl > TMAX));     // keep a window's output small: cut behind the last fitting sequence
        if (big != 0ull) {
            const uint32_t cut = ctz64(big);
            tokm &= (1ull << cut) - 1ull;
            W.done = false;
            cur = cut;                                          // the sequence at `cut` starts the next window ...
            W.stop_cx = tokm == 0ull;                           // ... or is a long run by itself: exact_token
            tk = (tokm >> lane) & 1ull;
            tl = tk ? tl : 0u;
            incl = wave_incl_add(tl);
        }
    }
    W.adv = cur;
    W.T = rdlane(incl, 63u);
    W.op0 = op0;
    const uint32_t o = op0 + incl - tl;
    const bool mt = tk & !fin;
    const uint32_t dm = o + lit;
    W.ok = !(W.T > D.cap - op0 || __ballot(mt & ((offs == 0u) | (offs > dm))) != 0ull);
    const uint32_t sm = dm - offs;
    const uint32_t wend = op0 + W.T;
    W.near_lo = wend > RB ? wend - RB : 0u;                    // positions from here on are in the ring once this window is done
    auto crosses = [](uint32_t pos, uint32_t n) -> bool { return (pos & RM) + n > RB; };
    W.lpl = tk & (lit != 0u) & (lit <= 16u) & !crosses(o, 16u);
    const bool far = mt & (sm + mlen <= W.near_lo);            // source older than the ring: read it from the output
    W.lpf = FARLOADS ? (W.ok & far & (mlen <= 32u) & !crosses(dm, 32u)) : far;
    W.lpn = mt & !far & (sm >= W.near_lo) & (sm + mlen <= op0) & (mlen <= 64u) & !crosses(sm, mlen + 15u) & !crosses(dm, mlen + 15u);
    W.f0 = u32x4{0u, 0u, 0u, 0u}; W.f1 = W.f0;
    if (FARLOADS && W.lpf) {                                   // requested now, written when the window is executed
        __builtin_memcpy(&W.f0, (const void*)(D.out + sm), 16);
        __builtin_memcpy(&W.f1, (const void*)(D.out + sm + 16u), 16);
    }
    W.lit = lit; W.ls = ls; W.mlen = mlen; W.offs = offs; W.o = o; W.tk = tk; W.mt = mt;
    LZ4D_TICK(3)
    return W;
}

__device__ __forceinline__ void exec_window(Dec& D, const Win& W) {
    LZ4D_T0
#ifdef LZ4D_EXP_NOEXEC      // tools: the parser's share of a window (the output is wrong)
    asm volatile("" :: "v"(W.lv), "v"(W.f0), "v"(W.f1), "v"(W.o), "v"(W.lit), "v"(W.mlen), "v"(W.offs));
    D.op = W.op0 + W.T;
    if (D.op - D.F >= FLUSH_AT) D.flush();
    return;
#endif
    const uint32_t dm = W.o + W.lit, sm = dm - W.offs;
    // ---- phase A: lane = sequence --------------------------------------------------------------------------------------
    if (W.lpl) write_exact16(D.ring + (W.o & RM), W.lv, W.lit);
    if (__ballot(W.lpn) != 0ull) {
#pragma unroll 1
        for (uint32_t k = 0u; k < 64u; k += 16u) {
            const bool act = W.lpn & (k < W.mlen);
            if (__ballot(act) == 0ull) break;
            if (act) {
                u32x4 v;
                __builtin_memcpy(&v, (const void*)(D.ring + ((sm + k) & RM)), 16);
                write_exact16(D.ring + ((dm + k) & RM), v, W.mlen - k);
            }
        }
    }
    if (W.lpf) {
        write_exact16(D.ring + (dm & RM), W.f0, W.mlen);
        if (W.mlen > 16u) write_exact16(D.ring + ((dm + 16u) & RM), W.f1, W.mlen - 16u);
    }
    LZ4D_TICK(4)
    // ---- phase B: what is left, one sequence at a time, in order -------------------------------------------------------
    const bool litB = W.tk & (W.lit != 0u) & !W.lpl;
    const bool matB = W.mt & !W.lpf & !W.lpn;
    // the usual member of this loop: a match that reads bytes written earlier in this window -- up to 64 bytes, source in
    // the ring, not overlapping itself: one LDS read and one write by the wavefront
    const bool easy = matB & !litB & (W.mlen <= 64u) & (sm >= W.near_lo) & (W.offs >= W.mlen);
    const uint32_t pk = W.offs | (W.mlen << 16);
    uint64_t rest = __ballot(litB | matB);
    const uint64_t litBm = __ballot(litB), matBm = __ballot(matB), easym = __ballot(easy);
    while (rest != 0ull) {
        const uint32_t q = ctz64(rest);
        rest &= rest - 1ull;
        if ((easym >> q) & 1ull) {
            const uint32_t dq = rdlane(dm, q), pq = rdlane(pk, q);
            if (D.lane < (pq >> 16)) D.ring[(dq + D.lane) & RM] = D.ring[(dq - (pq & 0xFFFFu) + D.lane) & RM];
            continue;
        }
        const uint32_t oq = rdlane(W.o, q), lq = rdlane(W.lit, q);
        if ((litBm >> q) & 1ull) D.coop_literals(rdlane(W.ls, q), oq, lq);
        if ((matBm >> q) & 1ull) D.coop_match(oq + lq, rdlane(W.offs, q), rdlane(W.mlen, q), W.near_lo);
    }
    LZ4D_TICK(5)
    D.op = W.op0 + W.T;
    if (D.op - D.F >= FLUSH_AT) D.flush();
    LZ4D_TICK(6)
#ifdef LZ4D_PROF
    if (D.lane == 0u) { atomicAdd(&g_wdec_prof[8], 1ull); atomicAdd(&g_wdec_prof[9], (unsigned long long)__builtin_popcountll(__ballot(W.tk))); atomicAdd(&g_wdec_prof[10], (unsigned long long)__builtin_popcountll(__ballot(matB))); atomicAdd(&g_wdec_prof[11], (unsigned long long)__builtin_popcountll(easym)); atomicAdd(&g_wdec_prof[12], (unsigned long long)__builtin_popcountll(__ballot(W.lpf))); atomicAdd(&g_wdec_prof[13], (unsigned long long)__builtin_popcountll(__ballot(W.lpn))); }
#endif
}

__global__ void __launch_bounds__(64 * WPB) lz4_decompress_wave_kernel(DecompressArgs a, int32_t redo_code) {
    __shared__ __attribute__((aligned(16))) uint8_t lds_raw[WPB * WAVE_LDS];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wv = uni(threadIdx.x >> 6);
    const uint32_t b = blockIdx.x * WPB + wv;
    if (b >= a.n) return;
    Dec D;
    D.in = (const g_u8*)(a.in_base + a.in_off[b]);
    D.out = (g_u8*)(a.out_base + a.out_off[b]);
    D.ring = (lds_u8*)lds_raw + wv * WAVE_LDS;
    D.ibuf = D.ring + RB;
    D.ilen = a.in_len[b];
    D.cap = a.out_cap[b];
    D.lane = lane;
    D.op = 0u; D.F = 0u;
    D.ib0 = 0xFFFF0000u;                              // nothing buffered yet
    bool ok = D.ilen != 0u, done = false;
    uint32_t ip = 0u;
    if (ok) {
        Win A = parse_window<true>(D, 0u, 0u);
        for (;;) {
            if (!A.ok) { ok = false; break; }
            const uint32_t ip_next = ip + A.adv;
            const bool pipe = !A.done && !A.stop_cx;
            Win B = A;
            if (pipe) B = parse_window<true>(D, ip_next, A.op0 + A.T);   // the next window's loads are in flight while this one is executed
            exec_window(D, A);
            ip = ip_next;
            if (A.done) { done = true; break; }
            if (A.stop_cx) {
                ok = exact_token(D, ip, done);
                if (!ok || done) break;
                A = parse_window<true>(D, ip, D.op);
                continue;
            }
            A = B;
        }
    }
    if (ok && done) {
        D.finish();
        if (lane == 0u) {
            a.status[b] = 0;
            a.out_len[b] = D.op;
            if (a.detail) { a.detail[2u * b] = 0u; a.detail[2u * b + 1u] = 0u; }
        }
    } else if (lane == 0u) {
        a.status[b] = redo_code;          // decoded again, with the reference's check order, by lz4_decompress_blocks_kernel
        a.out_len[b] = 0u;
    }
}

// =====================================================================================================================
// Two wavefronts per block: a PARSER (parse_window: the token chain, placement, classification) and an EXECUTOR
// (exec_window: the copies, the write-back, exact_token), coupled by a queue of window descriptors in LDS.  A block costs the
// same per byte whatever its size (one wavefront: 80 MB/s), so for batches of few, large blocks -- fewer wavefronts than
// the chip has SIMDs -- the chain itself is what can be shortened: the two halves of a window's work overlap.
// Every wait is bounded: a wavefront that gives up marks the block for the reference-order kernel (as any irregularity).
// =====================================================================================================================
constexpr uint32_t NQ = 4u;                          // window descriptors in flight
constexpr uint32_t P_DESC = RB + IB + IB_PAD;        // per lane and window: {lit | mlen << 9 | flags << 18, ls, offs, o}, the 16 literal bytes
constexpr uint32_t P_HDR = P_DESC + NQ * 2048u;      // per window: {op0, T, near_lo, adv}, {flags, ip behind the window, -, -}
constexpr uint32_t P_CTL = P_HDR + NQ * 32u;         // tail (parser), head (executor), resume sequence / ip / op / flags, abort
constexpr uint32_t PAIR_LDS = P_CTL + 32u;
constexpr uint32_t SPIN_LIMIT = 1u << 22;            // x s_sleep(1) = 64 clocks: a quarter of a second
typedef volatile __attribute__((address_space(3))) uint32_t lds_vu32;
typedef __attribute__((address_space(3))) u32x4 lds_u32x4;

__device__ __forceinline__ void put_window(lds_u8* lds, uint32_t slot, const Win& W, uint32_t ip_next, uint32_t lane) {
    const uint32_t fl = (W.tk ? 1u : 0u) | (W.mt ? 2u : 0u) | (W.lpl ? 4u : 0u) | (W.lpn ? 8u : 0u) | (W.lpf ? 16u : 0u);
    lds_u32x4* d = (lds_u32x4*)(lds + P_DESC + slot * 2048u + lane * 32u);
    d[0] = u32x4{W.lit | (W.mlen << 9) | (fl << 18), W.ls, W.offs, W.o};
    d[1] = W.lv;
    if (lane == 0u) {
        lds_u32x4* h = (lds_u32x4*)(lds + P_HDR + slot * 32u);
        h[0] = u32x4{W.op0, W.T, W.near_lo, W.adv};
        h[1] = u32x4{(W.ok ? 1u : 0u) | (W.done ? 2u : 0u) | (W.stop_cx ? 4u : 0u), ip_next, 0u, 0u};
    }
}
__device__ __forceinline__ Win get_window(const lds_u8* lds, uint32_t slot, uint32_t lane, uint32_t& ip_next) {
    Win W;
    const lds_u32x4* d = (const lds_u32x4*)(lds + P_DESC + slot * 2048u + lane * 32u);
    const u32x4 a = d[0];
    W.lv = d[1];
    const lds_u32x4* h = (const lds_u32x4*)(lds + P_HDR + slot * 32u);
    const u32x4 h0 = h[0], h1 = h[1];
    W.lit = a.x & 511u; W.mlen = (a.x >> 9) & 511u;
    const uint32_t fl = a.x >> 18;
    W.tk = fl & 1u; W.mt = fl & 2u; W.lpl = fl & 4u; W.lpn = fl & 8u;
    W.ls = a.y; W.offs = a.z; W.o = a.w;
    W.op0 = uni(h0.x); W.T = uni(h0.y); W.near_lo = uni(h0.z); W.adv = uni(h0.w);
    const uint32_t hf = uni(h1.x);
    W.ok = hf & 1u; W.done = hf & 2u; W.stop_cx = hf & 4u;
    ip_next = uni(h1.y);
    // a far match (source older than the ring) is written lane-parallel if it is short and does not wrap; its source is
    // requested by request_far
    const uint32_t dm = W.o + W.lit;
    W.lpf = W.ok && (fl & 16u) && (W.mlen <= 32u) && ((dm & RM) + 32u <= RB);
    W.f0 = u32x4{0u, 0u, 0u, 0u}; W.f1 = W.f0;
    return W;
}
__device__ __forceinline__ void request_far(const Dec& D, Win& W) {
    if (W.lpf) {
        const uint32_t sm = W.o + W.lit - W.offs;
        __builtin_memcpy(&W.f0, (const void*)(D.out + sm), 16);
        __builtin_memcpy(&W.f1, (const void*)(D.out + sm + 16u), 16);
    }
}

__global__ void __launch_bounds__(128) lz4_decompress_wave_pair_kernel(DecompressArgs a, int32_t redo_code) {
    extern __shared__ __attribute__((aligned(16))) uint8_t pair_lds[];
    lds_u8* lds = (lds_u8*)pair_lds;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t role = uni(threadIdx.x >> 6);          // 0 parser, 1 executor
    const uint32_t b = blockIdx.x;
    if (b >= a.n) return;
    lds_vu32* ctl = (lds_vu32*)(lds + P_CTL);             // [0] tail, [1] head, [2] resume seq, [3] ip, [4] op, [5] flags (1 done, 2 failed), [6] abort
    if (threadIdx.x < 8u) ctl[threadIdx.x] = 0u;
    __syncthreads();
    Dec D;
    D.in = (const g_u8*)(a.in_base + a.in_off[b]);
    D.out = (g_u8*)(a.out_base + a.out_off[b]);
    D.ring = lds;
    D.ibuf = lds + RB;
    D.ilen = a.in_len[b];
    D.cap = a.out_cap[b];
    D.lane = lane;
    D.op = 0u; D.F = 0u;
    D.ib0 = 0xFFFF0000u;
    if (D.ilen == 0u) {                                     // decompress.rs:207-209: the reference-order kernel reports it
        if (threadIdx.x == 0u) { a.status[b] = redo_code; a.out_len[b] = 0u; }
        return;
    }
    auto fence = []() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); };
    if (role == 0u) {
        // ---- parser ------------------------------------------------------------------------------------------------
        uint32_t ip = 0u, op0 = 0u, tail = 0u, seq = 0u;
        for (;;) {
            uint32_t spins = 0u;
            while (tail - ctl[1] >= NQ && ctl[6] == 0u && ++spins < SPIN_LIMIT) __builtin_amdgcn_s_sleep(1);
            if (ctl[6] != 0u) break;
            if (spins >= SPIN_LIMIT) { if (lane == 0u) ctl[6] = 1u; break; }
            const Win W = parse_window<false>(D, ip, op0);
            put_window(lds, tail & (NQ - 1u), W, ip + W.adv, lane);
            fence();
            tail += 1u;
            if (lane == 0u) ctl[0] = tail;
            if (!W.ok || W.done) break;
            ip += W.adv;
            op0 += W.T;
            if (W.stop_cx) {                                  // the executor decodes one sequence of any shape, then tells where to go on
                seq += 1u;
                spins = 0u;
                while (ctl[2] != seq && ctl[6] == 0u && ++spins < SPIN_LIMIT) __builtin_amdgcn_s_sleep(1);
                if (ctl[6] != 0u) break;
                if (spins >= SPIN_LIMIT) { if (lane == 0u) ctl[6] = 1u; break; }
                fence();
                if (ctl[5] != 0u) break;                      // the block ended or failed inside that sequence
                ip = ctl[3];
                op0 = ctl[4];
            }
        }
        return;
    }
    // ---- executor ------------------------------------------------------------------------------------------------------
    bool ok = true, done = false, have_next = false;
    uint32_t head = 0u, seq = 0u, ipn = 0u, ipn_next = 0u;
    Win A, B;
    for (;;) {
        if (have_next) { A = B; ipn = ipn_next; have_next = false; }
        else {
            uint32_t spins = 0u;
            while (ctl[0] == head && ctl[6] == 0u && ++spins < SPIN_LIMIT) __builtin_amdgcn_s_sleep(1);
            if (ctl[6] != 0u || spins >= SPIN_LIMIT) { ok = false; break; }
            fence();
            A = get_window(lds, head & (NQ - 1u), lane, ipn);
            request_far(D, A);
        }
        if (!A.ok) { ok = false; break; }
        if (!A.done && !A.stop_cx && ctl[0] - head >= 2u) {    // the window behind it is parsed already: its far sources travel while this one is executed
            fence();
            B = get_window(lds, (head + 1u) & (NQ - 1u), lane, ipn_next);
            request_far(D, B);
            have_next = true;
        }
        exec_window(D, A);
        head += 1u;
        if (lane == 0u) ctl[1] = head;
        if (A.done) { done = true; break; }
        if (A.stop_cx) {
            uint32_t ip = ipn;
            ok = exact_token(D, ip, done);
            seq += 1u;
            if (lane == 0u) { ctl[3] = ip; ctl[4] = D.op; ctl[5] = (done ? 1u : 0u) | (ok ? 0u : 2u); }
            fence();
            if (lane == 0u) ctl[2] = seq;
            if (!ok || done) break;
        }
    }
    if (ok && done) {
        D.finish();
        if (lane == 0u) {
            a.status[b] = 0;
            a.out_len[b] = D.op;
            if (a.detail) { a.detail[2u * b] = 0u; a.detail[2u * b + 1u] = 0u; }
        }
    } else {
        if (lane == 0u) {
            ctl[6] = 1u;                      // (the parser may be waiting for a slot)
            a.status[b] = redo_code;          // decoded again, with the reference's check order, by lz4_decompress_blocks_kernel
            a.out_len[b] = 0u;
        }
    }
}

}  // namespace wdec

// Blocks without dictionary / prefix.  Irregular blocks get status `redo_code`; the caller runs launch_decompress with
// only_status = redo_code behind this launch.
hipError_t launch_decompress_wave(const DecompressArgs& a, int32_t redo_code, hipStream_t s) {
    if (a.n == 0u) return hipSuccess;
    if (a.dict_base != nullptr || a.out_pos != nullptr) return hipErrorInvalidValue;
    const uint32_t grid = (a.n + wdec::WPB - 1u) / wdec::WPB;
    hipLaunchKernelGGL(wdec::lz4_decompress_wave_kernel, dim3(grid), dim3(64u * wdec::WPB), 0, s, a, redo_code);
    return hipGetLastError();
}

// The same contract with two wavefronts per block (see above): for batches with fewer blocks than the chip has SIMDs.
hipError_t launch_decompress_wave_pair(const DecompressArgs& a, int32_t redo_code, hipStream_t s) {
    if (a.n == 0u) return hipSuccess;
    if (a.dict_base != nullptr || a.out_pos != nullptr) return hipErrorInvalidValue;
    hipLaunchKernelGGL(wdec::lz4_decompress_wave_pair_kernel, dim3(a.n), dim3(128), wdec::PAIR_LDS, s, a, redo_code);
    return hipGetLastError();
}

}  // namespace lz4flex_dev

#ifdef LZ4D_PROF
extern "C" int lz4flex_debug_wdec_prof(unsigned long long* vals, int reset) {
    if (reset) {
        unsigned long long z[16] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(lz4flex_dev::wdec::g_wdec_prof), z, sizeof z);
        return 0;
    }
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(vals, HIP_SYMBOL(lz4flex_dev::wdec::g_wdec_prof), 128);
    return 0;
}
#endif

