// lz4_decompress_lds.hip -- batched LZ4 block decoder, LDS-staged variant ("v2").
//
// Same contract as lz4_decompress.hip (reference src/block/decompress.rs:201-449: result bytes,
// byte count, error variant and OutputTooSmall{expected,actual}, unsafe-flavour check order), for
// blocks decoded WITHOUT a dictionary / prefix.  What changes is where the bytes live:
//
//   * a group of G=8 lanes owns one block; a 64-lane wavefront decodes 8 blocks; 8 wavefronts per
//     CU keep 64 blocks per CU (16 384 blocks per MI355X) in flight -- the only parallelism the
//     format offers across sequences is across blocks, so residency is sized for "all blocks of
//     the 1 GiB batch at once" and the per-block LDS footprint is held to 2 496 B (160 KiB / 64).
//   * compressed input is streamed HBM -> LDS in 16-byte-per-lane pieces (coalesced) into a
//     384-byte window; the token/offset/length parse reads only LDS (lgkmcnt), never waits on
//     global memory and in particular never on outstanding stores (gfx950 counts loads and
//     stores on one in-order vmcnt, so in the v1 kernel every token load also waited for the
//     previous match's write acknowledgements).
//   * output is produced in LDS: a linear buffer holding the last >= 1 KiB of history plus the
//     bytes not yet written back.  Matches whose source is inside that history are LDS -> LDS
//     copies; older sources are read from the already written-back output in HBM/L2.
//     Write-back is 16 bytes per lane, contiguous (full 128-byte lines per group), so HBM sees the
//     uncompressed bytes exactly once, coalesced.
//   * overlapping matches use the periodic form out[op+i] = out[op-offset + (i mod offset)].
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lz4_device.h"

namespace lz4flex_dev {
#ifdef LZ4FLEX_PROFILE_PHASES
__device__ unsigned long long gd_phase_cycles[8];
__device__ unsigned long long gd_phase_counts[8];
#define D_PHASE_DECL unsigned long long _pt = __builtin_readcyclecounter(); unsigned long long _pacc[8] = {0,0,0,0,0,0,0,0}; unsigned _pcnt[8] = {0,0,0,0,0,0,0,0};
#define D_PHASE_MARK(k) { const unsigned long long _n = __builtin_readcyclecounter(); _pacc[k] += _n - _pt; _pcnt[k]++; _pt = _n; }
#define D_PHASE_FLUSH if (threadIdx.x == 0) { for (int _k = 0; _k < 8; ++_k) { atomicAdd(&gd_phase_cycles[_k], _pacc[_k]); atomicAdd(&gd_phase_counts[_k], (unsigned long long)_pcnt[_k]); } }
#else
#define D_PHASE_DECL
#define D_PHASE_MARK(k)
#define D_PHASE_FLUSH
#endif
namespace v2 {

// Geometry of one decoder flavour.  G lanes own a block and copy WB bytes each per piece (a piece is
// G*WB bytes); the per-block LDS is IN_CAP + IN_PAD (compressed-input window) + OUT_CAP (output buffer
// holding OUT_H bytes of history after a write-back plus the bytes not yet written back).
template <uint32_t G_, uint32_t WB_, uint32_t IN_CAP_, uint32_t OUT_H_, uint32_t OUT_CAP_>
struct Geometry {
    static constexpr uint32_t G = G_;                 // lanes per block
    static constexpr uint32_t WB = WB_;               // bytes per lane and piece in the pipelined decoder (4 or 8)
    static constexpr uint32_t PIECE = G_ * WB_;
    static constexpr uint32_t IN_CAP = IN_CAP_;       // compressed-input window (bytes)
    static constexpr uint32_t IN_PAD = 32;            // readable slack behind the window (parse window + wild copies)
    static constexpr uint32_t OUT_H = OUT_H_;         // history kept in LDS after a write-back (older sources: pipelined HBM loads)
    static constexpr uint32_t OUT_SLACK = 32;         // wild-copy slack
    static constexpr uint32_t OUT_CAP = OUT_CAP_;
    static constexpr uint32_t GROUP_LDS = IN_CAP + IN_PAD + OUT_CAP;
    static constexpr uint32_t IN_SLIDE = IN_CAP / 3;                              // slide the window once this much is consumed
    static constexpr uint32_t FLUSH_AT = (OUT_CAP - OUT_SLACK - OUT_H) / 2 - 8;   // service: write back below this much space
    static constexpr uint32_t SLOW_FLUSH_AT = FLUSH_AT < 320u ? FLUSH_AT : 320u;
    static_assert(GROUP_LDS % 16 == 0 && OUT_H % 16 == 0 && IN_CAP % 16 == 0, "16-byte pieces");
    static_assert(OUT_H >= 2 * PIECE + 32, "a far piece must lie entirely in the written-back part");
    static_assert((64 / G) * GROUP_LDS <= 65536, "static LDS per workgroup");
};
// Measured on the configs[1] workload (tools/dec_geometry.py): a batch of n blocks puts n*G/64 wavefronts on 1 024
// SIMDs, and two wavefronts per SIMD are needed to cover each other's LDS / issue latency (a lone wavefront needs
// 1 760 cycles per step, two share a SIMD at 1 460 cycles per step each).  16 384 blocks: Geo8 2.93 ms, Geo4s
// 3.67 ms (one wavefront per SIMD); 32 768 blocks: Geo8 5.80 ms (two rounds), Geo4s 4.18 ms.  Tried and dropped:
// 4 lanes x 4 B (16-byte pieces: 40 % more steps, 3.88 ms), 4 x 8 B with 2 496 B of LDS (never more than one
// wavefront per SIMD: 3.63 / 7.26 ms), 8 x 8 B (64-byte pieces: 16 % fewer steps but no faster, 2.99 ms).
using Geo8 = Geometry<8, 4, 384, 512, 2080>;      // 8 blocks per wavefront, 2 496 B per block: 8 wavefronts per CU hold 64 blocks
using Geo4s = Geometry<4, 8, 192, 256, 1024>;     // 16 blocks per wavefront, 1 248 B per block: 8 wavefronts per CU hold 128 blocks
static_assert(Geo8::GROUP_LDS == 2496 && Geo8::FLUSH_AT == 760 && Geo8::IN_SLIDE == 128, "round-1 geometry");
static_assert(Geo4s::GROUP_LDS == 1248, "LDS budget per block");

__device__ __forceinline__ uint32_t ld32(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
__device__ __forceinline__ void st32(uint8_t* p, uint32_t v) { __builtin_memcpy(p, &v, 4); }
__device__ __forceinline__ uint4 ld128(const uint8_t* p) { uint4 v; __builtin_memcpy(&v, p, 16); return v; }
__device__ __forceinline__ void st128(uint8_t* p, uint4 v) { __builtin_memcpy(p, &v, 16); }
// the WB-byte word one lane moves per piece
template <uint32_t WB> struct Word;
template <> struct Word<4> { using type = uint32_t; };
template <> struct Word<8> { using type = uint2; };
template <uint32_t WB> __device__ __forceinline__ typename Word<WB>::type ldw(const uint8_t* p) {
    typename Word<WB>::type v; __builtin_memcpy(&v, p, WB); return v;
}
template <uint32_t WB> __device__ __forceinline__ void stw(uint8_t* p, typename Word<WB>::type v) { __builtin_memcpy(p, &v, WB); }

template <class GEO, bool ABLATE_FAR>   // ABLATE_FAR: timing ablation only, far sources read garbage from LDS instead of HBM (wrong bytes)
struct Dec {
    static constexpr uint32_t G = GEO::G, IN_CAP = GEO::IN_CAP, OUT_H = GEO::OUT_H, OUT_CAP = GEO::OUT_CAP,
                              OUT_SLACK = GEO::OUT_SLACK;
    const uint8_t* gin;   // compressed block (global)
    uint8_t* gout;        // output block (global)
    uint8_t* lin;         // LDS: input window
    uint8_t* lout;        // LDS: output buffer
    uint32_t g;           // lane in group
    uint32_t ilen, cap;
    uint32_t ip, op;
    uint32_t in_lo, in_end;   // window holds compressed positions [in_lo, in_end)
    uint32_t L0, F;           // LDS output holds positions [L0, op); [0, F) is written back to HBM

    // ---- input window -----------------------------------------------------------------------
    // slide the window so that it starts at (at & ~15) and fill it from HBM
    __device__ __forceinline__ void refill(uint32_t at) {
        const uint32_t new_lo = at & ~15u;
        if (new_lo > in_lo) {
            const uint32_t shift = new_lo - in_lo;
            if (new_lo < in_end) {
                const uint32_t keep = in_end - new_lo;
                for (uint32_t i = 16u * g; i < keep; i += 16u * G) {
                    const uint4 v = *reinterpret_cast<const uint4*>(lin + shift + i);
                    *reinterpret_cast<uint4*>(lin + i) = v;
                }
            } else {
                in_end = new_lo;
            }
            in_lo = new_lo;
        }
        uint32_t want = in_lo + IN_CAP;
        if (want > ilen) want = ilen;
        for (uint32_t pos = in_end + 16u * g; pos < want; pos += 16u * G) {
            if (pos + 16u <= ilen) {
                st128(lin + (pos - in_lo), ld128(gin + pos));
            } else {
                for (uint32_t k = pos; k < ilen; ++k) lin[k - in_lo] = gin[k];
            }
        }
        in_end = want;
    }
    // byte at compressed position `pos` (< ilen), refilling if it is not in the window
    __device__ __forceinline__ uint32_t peek(uint32_t pos) {
        if (pos >= in_end || pos < in_lo) refill(pos);
        return lin[pos - in_lo];
    }

    // ---- output buffer ----------------------------------------------------------------------
    // write back [F, op & ~15) and drop history older than OUT_H
    __device__ __forceinline__ void flush_slide() {
        const uint32_t fnew = op & ~15u;
        for (uint32_t p = F + 16u * g; p < fnew; p += 16u * G)
            st128(gout + p, *reinterpret_cast<const uint4*>(lout + (p - L0)));
        F = fnew;
        const uint32_t new_l0 = F > OUT_H ? F - OUT_H : 0u;   // multiple of 16
        if (new_l0 > L0) {
            const uint32_t shift = new_l0 - L0;
            const uint32_t keep = op - new_l0;
            for (uint32_t i = 16u * g; i < keep; i += 16u * G) {
                const uint4 v = *reinterpret_cast<const uint4*>(lout + shift + i);
                *reinterpret_cast<uint4*>(lout + i) = v;
            }
            L0 = new_l0;
        }
    }
    __device__ __forceinline__ uint32_t out_space() const { return L0 + OUT_CAP - OUT_SLACK - op; }
    __device__ __forceinline__ void final_flush() {
        const uint32_t fnew = op & ~15u;
        for (uint32_t p = F + 16u * g; p < fnew; p += 16u * G)
            st128(gout + p, *reinterpret_cast<const uint4*>(lout + (p - L0)));
        for (uint32_t p = fnew + g; p < op; p += G) gout[p] = lout[p - L0];
        F = op;
    }

    // ---- copies -----------------------------------------------------------------------------
    // literals: compressed positions [s, s+n) -> output positions [op, op+n); advances op
    __device__ __forceinline__ void copy_literals(uint32_t s, uint32_t n) {
        while (n != 0u) {
            if (s >= in_end || s < in_lo) refill(s);
            uint32_t space = out_space();
            if (space < 64u && space < n) { flush_slide(); space = out_space(); }
            uint32_t m = in_end - s;
            if (m > n) m = n;
            if (m > space) m = space;
            const uint8_t* src = lin + (s - in_lo);
            uint8_t* dst = lout + (op - L0);
            for (uint32_t i = 4u * g; i < m; i += 4u * G) st32(dst + i, ld32(src + i));   // wild: <= 3 bytes over, inside the pads
            s += m; op += m; n -= m;
        }
    }
    // match: offset back from op, n bytes; advances op
    __device__ __forceinline__ void copy_match(uint32_t offset, uint32_t n) {
        while (n != 0u) {
            uint32_t space = out_space();
            if (space < 64u && space < n) { flush_slide(); space = out_space(); }
            const uint32_t m = n < space ? n : space;
            uint8_t* dst = lout + (op - L0);
            const uint32_t src = op - offset;
            if (offset >= 4u * G) {
                for (uint32_t i = 4u * g; i < m; i += 4u * G) {
                    const uint32_t p = src + i;
                    // inside the LDS history, or older bytes already written back (p + 4 <= L0 + 3 < F)
                    const uint32_t v = (p >= L0) ? ld32(lout + (p - L0)) : (ABLATE_FAR ? ld32(lout + (p & 1023u)) : ld32(gout + p));
                    st32(dst + i, v);
                }
            } else if (offset == 1u) {
                const uint32_t v = (uint32_t)dst[-1] * 0x01010101u;
                for (uint32_t i = 4u * g; i < m; i += 4u * G) st32(dst + i, v);
            } else {
                // periodic: the `offset` bytes before op are in LDS (offset < 32 <= op - L0 or L0 == 0)
                const uint8_t* pat = dst - offset;
                uint32_t idx = (4u * g) % offset;
                const uint32_t step = (4u * G) % offset;
                for (uint32_t i = 4u * g; i < m; i += 4u * G) {
                    uint32_t j = idx, v = 0u;
#pragma unroll
                    for (uint32_t k = 0; k < 4u; ++k) {
                        v |= (uint32_t)pat[j] << (8u * k);
                        j = (j + 1u == offset) ? 0u : j + 1u;
                    }
                    st32(dst + i, v);
                    idx += step;
                    if (idx >= offset) idx -= offset;
                }
            }
            op += m; n -= m;
        }
    }

    // ---- generic ("slow") handling of exactly one sequence at ip (reference src/block/decompress.rs:244-444):
    // any literal/match length, periodic matches, end of block, every error.  Also performs the window
    // refill / write-back the fast path asked for.  Returns: 0 continue, 1 block finished, <0 -error code.
    __device__ __forceinline__ int32_t slow_step(uint64_t* det_expected) {
        // keep >= 24 bytes of lookahead while the input lasts; slide when 128 bytes are consumed
        if ((in_end - ip < 24u && in_end < ilen) || ip - in_lo >= GEO::IN_SLIDE) refill(ip);
        if (out_space() < GEO::SLOW_FLUSH_AT) flush_slide();
        const uint32_t avail = in_end - ip;
        const uint32_t rel = ip - in_lo;
        const uint32_t* lw = reinterpret_cast<const uint32_t*>(lin + (rel & ~3u));
        const uint32_t a = lw[0], b = lw[1], c = lw[2];
        const uint32_t sh = rel & 3u;
        const uint32_t w0 = __builtin_amdgcn_alignbyte(b, a, sh);   // compressed bytes ip .. ip+3
        const uint32_t w1 = __builtin_amdgcn_alignbyte(c, b, sh);   // ip+4 .. ip+7
        const bool fast = avail >= 16u;
        const uint32_t token = w0 & 0xFFu;
        ip += 1u;
        uint32_t lit = token >> 4;
        bool have_win = fast;     // w0/w1 still describe the bytes at (ip - 1)
        // ---- literals (:334-362)
        if (lit != 0u) {
            if (lit == 15u) {
                have_win = false;
                for (;;) {   // read_integer_ptr :126-157
                    if (ip >= ilen) return -LZ4FLEX_DEV_E_EXPECTED_ANOTHER_BYTE;
                    const uint32_t e = peek(ip);
                    ip += 1u;
                    lit += e;
                    if (e != 0xFFu) break;
                }
            }
            if (lit > ilen - ip) return -LZ4FLEX_DEV_E_LITERAL_OUT_OF_BOUNDS;
            if (lit > cap - op) { *det_expected = (uint64_t)op + lit; return -LZ4FLEX_DEV_E_OUTPUT_TOO_SMALL; }
            if (have_win && lit <= 7u && out_space() >= 8u) {
                // the literals are window bytes 1..lit: two lanes store a dword each
                if (g < 2u) {
                    const uint32_t v = g == 0u ? __builtin_amdgcn_alignbyte(w1, w0, 1u) : (w1 >> 8);
                    st32(lout + (op - L0) + 4u * g, v);
                }
                op += lit;
            } else {
                copy_literals(ip, lit);
                have_win = false;
            }
            ip += lit;
        }
        if (ip >= ilen) return 1;                                 // :366-368
        if (ilen - ip < 2u) return -LZ4FLEX_DEV_E_EXPECTED_ANOTHER_BYTE;   // :373-375
        // ---- offset, match length (:377-391)
        uint32_t offset;
        uint32_t ml = 4u + (token & 15u);
        if (have_win && lit <= 4u) {
            const uint32_t sft = 8u * (1u + lit);
            const uint64_t w = ((uint64_t)w1 << 32) | w0;
            const uint32_t t = (uint32_t)(w >> sft);
            offset = t & 0xFFFFu;
            ip += 2u;
            if (offset == 0u) return -LZ4FLEX_DEV_E_OFFSET_ZERO;
            if (ml == 19u) {
                uint32_t e = (t >> 16) & 0xFFu;     // in the window: 3 + lit <= 7
                ip += 1u;
                ml += e;
                while (e == 0xFFu) {
                    if (ip >= ilen) return -LZ4FLEX_DEV_E_EXPECTED_ANOTHER_BYTE;
                    e = peek(ip);
                    ip += 1u;
                    ml += e;
                }
            }
        } else {
            offset = peek(ip) | (peek(ip + 1u) << 8);
            ip += 2u;
            if (offset == 0u) return -LZ4FLEX_DEV_E_OFFSET_ZERO;
            if (ml == 19u) {
                for (;;) {
                    if (ip >= ilen) return -LZ4FLEX_DEV_E_EXPECTED_ANOTHER_BYTE;
                    const uint32_t e = peek(ip);
                    ip += 1u;
                    ml += e;
                    if (e != 0xFFu) break;
                }
            }
        }
        // ---- bounds (:398-408, unsafe-flavour order)
        if (offset > op) return -LZ4FLEX_DEV_E_OFFSET_OUT_OF_BOUNDS;
        if (ml > cap - op) { *det_expected = (uint64_t)op + ml; return -LZ4FLEX_DEV_E_OUTPUT_TOO_SMALL; }
        copy_match(offset, ml);
        if (ip >= ilen) return -LZ4FLEX_DEV_E_EXPECTED_ANOTHER_BYTE;   // :439-443
        return 0;
    }

    // second half of a sequence (offset, match length, match copy) for a group whose literals are already
    // emitted; `mlc` is the token's low nibble.  Same return convention as slow_step.
    __device__ __forceinline__ int32_t slow_offset_part(uint32_t mlc, uint64_t* det_expected) {
        if (ip >= ilen) return 1;
        if (ilen - ip < 2u) return -LZ4FLEX_DEV_E_EXPECTED_ANOTHER_BYTE;
        const uint32_t offset = peek(ip) | (peek(ip + 1u) << 8);
        ip += 2u;
        if (offset == 0u) return -LZ4FLEX_DEV_E_OFFSET_ZERO;
        uint32_t ml = 4u + mlc;
        if (ml == 19u) {
            for (;;) {
                if (ip >= ilen) return -LZ4FLEX_DEV_E_EXPECTED_ANOTHER_BYTE;
                const uint32_t e = peek(ip);
                ip += 1u;
                ml += e;
                if (e != 0xFFu) break;
            }
        }
        if (offset > op) return -LZ4FLEX_DEV_E_OFFSET_OUT_OF_BOUNDS;
        if (ml > cap - op) { *det_expected = (uint64_t)op + ml; return -LZ4FLEX_DEV_E_OUTPUT_TOO_SMALL; }
        copy_match(offset, ml);
        if (ip >= ilen) return -LZ4FLEX_DEV_E_EXPECTED_ANOTHER_BYTE;
        return 0;
    }

    __device__ __forceinline__ int32_t run(uint64_t* det_expected) {
        if (ilen == 0u) return LZ4FLEX_DEV_E_EXPECTED_ANOTHER_BYTE;   // :207-209
        in_lo = 0u; in_end = 0u; L0 = 0u; F = 0u; ip = 0u; op = 0u;
        refill(0u);
        for (;;) {
            const int32_t r = slow_step(det_expected);
            if (r > 0) break;
            if (r < 0) return -r;
        }
        final_flush();
        return 0;
    }
};

// ---------------------------------------------------------------------------------------------------
// Pipelined decoder ("v4"): the same LDS-staged buffers, but the per-group work is cut into uniform
// STEPS so that the 8 groups of a wavefront run one common instruction stream, and every step is
// PARSED (front end) Q-1 steps before it is EXECUTED (back end):
//   * front end, per step: read a 16-byte window at ip; a token with <= 12 literals, its offset and
//     at most one length-extension byte are decoded from registers; longer literal runs and matches
//     are emitted as 32-byte pieces over several steps (micro-states lit_rem / need_off / ml_rem).
//     A match piece whose source is older than the LDS history issues its HBM load here.
//   * back end, per step: executes the piece parsed Q-1 steps earlier: LDS->LDS literal copy,
//     LDS->LDS match copy or the store of the HBM dword that has been in flight for Q-1 steps.
//     Execution stays strictly in sequence order, so LDS read-after-write needs no tracking.
//   * anything else (errors, 255-chains, offsets < 4, block tail, window refill, write-back) blocks
//     the group; once per 4-step iteration the wave checks for blocked groups, drains the pipeline
//     and runs the generic code above for them (and window/write-back maintenance for every group).
template <class GEO, bool ABLATE_FAR>
struct PipeDec : Dec<GEO, ABLATE_FAR> {
    using B = Dec<GEO, ABLATE_FAR>;
    static constexpr uint32_t G = GEO::G, WB = GEO::WB, PIECE = GEO::PIECE;
    using word_t = typename Word<WB>::type;
    using B::gin; using B::gout; using B::lin; using B::lout; using B::g; using B::ilen; using B::cap;
    using B::ip; using B::op; using B::in_lo; using B::in_end; using B::L0; using B::F;

    struct Slot { uint32_t lit_n, lit_src, lit_dst, m_n, m_src, m_dst, far; word_t v; };
    enum : uint32_t { K_NONE = 0, K_MAINT = 1, K_RARE_TOKEN = 2, K_RARE_OFFSET = 3, K_FINISH = 4 };

    uint32_t lit_rem, ml_rem, moff, mlc_saved;
    uint32_t need_off;    // literals of the current sequence are emitted, offset not parsed yet
    uint32_t blocked;     // K_*
    uint32_t done;
    const uint8_t* dummy; // always-readable address for lanes without a far load

    // Front end, written as straight-line data flow (selects, no divergent branches): all 8 groups of the
    // wavefront run the same instructions whatever micro-state they are in.
    __device__ __forceinline__ void fe_step(Slot& s) {
        const bool active = (done | blocked) == 0u;
        const uint32_t avail = in_end - ip;
        const bool space_ok = B::out_space() >= 64u;
        const bool boundary = (lit_rem | ml_rem) == 0u;       // at a token, or at the offset after a long literal run
        const bool need = need_off != 0u;
        // ---- 16-byte window at ip (always read: the LDS address is always inside the group's buffer)
        const uint32_t rel = ip - in_lo;
        const uint32_t* lw = reinterpret_cast<const uint32_t*>(lin + (rel & ~3u));
        const uint32_t q0 = lw[0], q1 = lw[1], q2 = lw[2], q3 = lw[3], q4 = lw[4];
        const uint32_t sh = rel & 3u;
        const uint32_t W0 = __builtin_amdgcn_alignbyte(q1, q0, sh);
        const uint32_t W1 = __builtin_amdgcn_alignbyte(q2, q1, sh);
        const uint32_t W2 = __builtin_amdgcn_alignbyte(q3, q2, sh);
        const uint32_t W3 = __builtin_amdgcn_alignbyte(q4, q3, sh);
        // ---- token fields (meaningful when !need)
        const uint32_t lc = (W0 >> 4) & 15u;
        const uint32_t mlc_t = W0 & 15u;
        const uint32_t e1 = (W0 >> 8) & 0xFFu;
        const bool lc15 = lc == 15u;
        const uint32_t lit_t = lc15 ? 15u + e1 : lc;
        const uint32_t hdr = lc15 ? 2u : 1u;
        const bool rare_t = (lc15 && e1 == 0xFFu) || lit_t > ilen - ip - hdr || lit_t > cap - op;
        const bool long_t = lit_t > 12u;
        // ---- offset / extension byte at window index pos_off
        const uint32_t lit_s = need ? 0u : lit_t;
        const uint32_t pos_off = need ? 0u : 1u + lit_t;      // <= 13 whenever it is used
        const uint32_t mlc = need ? mlc_saved : mlc_t;
        const uint32_t wi = pos_off >> 2;
        const uint32_t lo = wi == 0u ? W0 : (wi == 1u ? W1 : (wi == 2u ? W2 : W3));
        const uint32_t hi = wi == 0u ? W1 : (wi == 1u ? W2 : (wi == 2u ? W3 : 0u));
        const uint32_t t = __builtin_amdgcn_alignbyte(hi, lo, pos_off & 3u);
        const uint32_t offset = t & 0xFFFFu;
        const uint32_t e = (t >> 16) & 0xFFu;
        const bool ext = mlc == 15u;
        const uint32_t ml = 4u + mlc + (ext ? e : 0u);
        const uint32_t mstart = op + lit_s;
        const bool rare_o = (ext && e == 0xFFu) || offset < WB || offset > mstart || ml > cap - mstart;
        // ---- what this group does in this step
        const bool tok = active && space_ok && boundary;
        const bool win_ok = avail >= 20u;
        const bool do_parse = tok && win_ok;
        const bool start_long = do_parse && !need && !rare_t && long_t;
        const bool do_short = do_parse && (need || (!rare_t && !long_t)) && !rare_o;
        const bool rare = (do_parse && !start_long && !do_short) || (tok && !win_ok && in_end >= ilen);
        bool maint = active && (!space_ok || (tok && !win_ok && in_end < ilen));
        // ---- commit the parse
        const uint32_t ip1 = do_short ? ip + pos_off + 2u + (ext ? 1u : 0u) : (start_long ? ip + hdr : ip);
        const uint32_t op1 = do_short ? mstart : op;
        const uint32_t lit_rem1 = start_long ? lit_t : lit_rem;
        const uint32_t ml_rem1 = do_short ? ml : ml_rem;
        moff = do_short ? offset : moff;
        mlc_saved = start_long ? mlc_t : mlc_saved;
        need_off = do_short ? 0u : (start_long ? 1u : need_off);
        const bool lit_now = do_short && !need && lit_s != 0u;      // <= 12 literals straight from the window
        // ---- one piece: a long-literal piece, else a match piece
        const bool go = active && space_ok && !rare && !maint;
        const bool lpiece = go && lit_rem1 != 0u;
        const uint32_t ln = lit_rem1 < PIECE ? lit_rem1 : PIECE;
        const bool l_in_ok = in_end - ip1 >= ln;
        const bool do_l = lpiece && l_in_ok;
        maint = maint || (lpiece && !l_in_ok);
        const bool do_m = go && lit_rem1 == 0u && ml_rem1 != 0u;
        const uint32_t pm = moff >= PIECE ? PIECE : (moff & ~(WB - 1u));
        const uint32_t mn = ml_rem1 < pm ? ml_rem1 : pm;
        const uint32_t msrc = op1 - moff;
        const bool far = do_m && msrc < L0;
        // literal piece fields (short literals and long pieces are mutually exclusive)
        s.lit_n = lit_now ? lit_s : (do_l ? ln : 0u);
        s.lit_src = lit_now ? rel + 1u : ip1 - in_lo;
        s.lit_dst = op - L0;                                        // == op1 - L0 for a long piece (op1 == op then)
        s.m_n = do_m ? mn : 0u;
        s.m_dst = op1 - L0;
        s.m_src = msrc - L0;
        s.far = far ? 1u : 0u;
        const uint8_t* ld_addr = far ? gout + msrc + WB * g : dummy;   // msrc + PIECE <= L0 + PIECE - 1 < F: written back
        // ---- advance
        ip = do_l ? ip1 + ln : ip1;
        op = do_l ? op1 + ln : (do_m ? op1 + mn : op1);
        lit_rem = do_l ? lit_rem1 - ln : lit_rem1;
        ml_rem = do_m ? ml_rem1 - mn : ml_rem1;
        const bool finish = do_l && lit_rem == 0u && ip >= ilen;   // the block's last (literal-only) sequence
        need_off = finish ? 0u : need_off;
        blocked = finish ? (uint32_t)K_FINISH
                         : (rare ? (need ? (uint32_t)K_RARE_OFFSET : (uint32_t)K_RARE_TOKEN)
                                 : (maint ? (uint32_t)K_MAINT : blocked));
        if (ABLATE_FAR) s.v = word_t{}; else s.v = ldw<WB>(ld_addr);   // exactly one HBM load per step and lane: exact vmcnt bookkeeping
    }

    __device__ __forceinline__ void be_step(const Slot& s) {
        // literal write strictly before the match read: a match may start inside the literals just written
        if (WB * g < s.lit_n) stw<WB>(lout + s.lit_dst + WB * g, ldw<WB>(lin + s.lit_src + WB * g));
        if (WB * g < s.m_n) {
            word_t x = s.v;
            if (!s.far) x = ldw<WB>(lout + s.m_src + WB * g);
            stw<WB>(lout + s.m_dst + WB * g, x);
        }
    }

    // generic handling for blocked groups + window / write-back maintenance for every live group
    __device__ __forceinline__ void service(int32_t& status, uint64_t* det_expected) {
        if (done) return;
        if ((in_end - ip < GEO::IN_SLIDE && in_end < ilen) || ip - in_lo >= GEO::IN_SLIDE) B::refill(ip);
        if (B::out_space() < GEO::FLUSH_AT) B::flush_slide();
        int32_t r = 0;
        if (blocked == K_FINISH) {
            r = 1;
        } else if (blocked == K_RARE_TOKEN) {
            r = B::slow_step(det_expected);
            while (r == 0 && in_end == ilen && ilen - ip < 24u) r = B::slow_step(det_expected);   // block tail
        } else if (blocked == K_RARE_OFFSET) {
            r = B::slow_offset_part(mlc_saved, det_expected);
            need_off = 0u;
        }
        blocked = K_NONE;
        if (r > 0) { B::final_flush(); done = 1u; }
        else if (r < 0) { status = -r; done = 1u; }
    }

    __device__ __forceinline__ int32_t run(uint64_t* det_expected) {
        if (ilen == 0u) return LZ4FLEX_DEV_E_EXPECTED_ANOTHER_BYTE;
        in_lo = 0u; in_end = 0u; L0 = 0u; F = 0u; ip = 0u; op = 0u;
        lit_rem = 0u; ml_rem = 0u; moff = 0u; mlc_saved = 0u; need_off = 0u; blocked = K_NONE; done = 0u;
        B::refill(0u);
        int32_t status = 0;
        Slot s0, s1, s2, s3;
        D_PHASE_DECL
        for (;;) {
            D_PHASE_MARK(0)
            s1.lit_n = 0u; s1.m_n = 0u; s1.far = 0u; s1.v = word_t{}; s1.lit_src = s1.lit_dst = s1.m_src = s1.m_dst = 0u;
            s2 = s1; s3 = s1;
            do {
                fe_step(s0); be_step(s1);
                fe_step(s1); be_step(s2);
                fe_step(s2); be_step(s3);
                fe_step(s3); be_step(s0);
                D_PHASE_MARK(1)   // one 4-step iteration of the steady loop
            } while (!__any(blocked != K_NONE));
            be_step(s1); be_step(s2); be_step(s3);
            D_PHASE_MARK(2)       // drain
            service(status, det_expected);
            D_PHASE_MARK(3)       // service
            if (done) break;
        }
        D_PHASE_FLUSH
        return status;
    }
};

template <class GEO, bool ABLATE_FAR>
__global__ void __launch_bounds__(64) lz4_decompress_pipe_kernel(DecompressArgs a) {
    constexpr uint32_t G = GEO::G, GROUP_LDS = GEO::GROUP_LDS, IN_CAP = GEO::IN_CAP, IN_PAD = GEO::IN_PAD;
    __shared__ __attribute__((aligned(16))) uint8_t lds[(64 / G) * GROUP_LDS];
    const uint32_t lane = threadIdx.x;
    const uint32_t b = blockIdx.x * (64u / G) + lane / G;
    if (b >= a.n) return;
    PipeDec<GEO, ABLATE_FAR> d;
    d.g = lane % G;
    d.gin = a.in_base + a.in_off[b];
    d.gout = a.out_base + a.out_off[b];
    d.lin = lds + (lane / G) * GROUP_LDS;
    d.lout = d.lin + IN_CAP + IN_PAD;
    d.ilen = a.in_len[b];
    d.cap = a.out_cap[b];
    d.dummy = reinterpret_cast<const uint8_t*>(a.in_off);   // always readable, >= 8 bytes
    uint64_t expected = 0u;
    const int32_t st = d.run(&expected);
    if (d.g == 0u) {
        a.status[b] = st;
        a.out_len[b] = st == 0 ? d.op : 0u;
        if (a.detail) {
            a.detail[2u * b] = st == LZ4FLEX_DEV_E_OUTPUT_TOO_SMALL ? expected : 0u;
            a.detail[2u * b + 1u] = st == LZ4FLEX_DEV_E_OUTPUT_TOO_SMALL ? (uint64_t)d.cap : 0u;
        }
    }
}

}  // namespace v2

template <class GEO, bool ABLATE_FAR>
static hipError_t launch_pipe_geo(const DecompressArgs& a, hipStream_t s) {
    const uint32_t per_wg = 64u / GEO::G;
    const uint32_t grid = (a.n + per_wg - 1u) / per_wg;
    hipLaunchKernelGGL((v2::lz4_decompress_pipe_kernel<GEO, ABLATE_FAR>), dim3(grid), dim3(64), 0, s, a);
    return hipGetLastError();
}

// geometry: 0 = 8 lanes x 4 B per block (Geo8), 1 = 4 lanes x 8 B with 1 248 B of LDS per block (Geo4s),
// -1 = by batch size: Geo4s once the batch is large enough to give it two wavefronts per SIMD
hipError_t launch_decompress_pipe(const DecompressArgs& a, hipStream_t s, int /*unused*/, int geometry) {
    if (a.n == 0u) return hipSuccess;
    if (a.dict_base != nullptr || a.out_pos != nullptr) return hipErrorInvalidValue;
    if (geometry < 0) geometry = a.n > 20480u ? 1 : 0;
    switch (geometry) {
        case 0: return launch_pipe_geo<v2::Geo8, false>(a, s);
        case 1: return launch_pipe_geo<v2::Geo4s, false>(a, s);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace lz4flex_dev

#ifdef LZ4FLEX_PROFILE_PHASES
extern "C" int lz4flex_debug_phase_dec(unsigned long long* cycles, unsigned long long* counts, int reset) {
    if (reset) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(lz4flex_dev::gd_phase_cycles), z, sizeof z);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(lz4flex_dev::gd_phase_counts), z, sizeof z);
        return 0;
    }
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(cycles, HIP_SYMBOL(lz4flex_dev::gd_phase_cycles), 64);
    (void)hipMemcpyFromSymbol(counts, HIP_SYMBOL(lz4flex_dev::gd_phase_counts), 64);
    return 0;
}
#endif
