// lz4_compress_lds.hip -- batched LZ4 block encoder, LDS-staged variant ("v2") for blocks <= 64 KiB.
//
// Same bytes as lz4_compress.hip / the reference encoder (src/block/compress.rs:318-489): same hash,
// table semantics, skip schedule, extension and end-of-block rules.  What changes is the memory path,
// because the encoder is a serial chain of ~3-6 k sequences per block and every dependent global round
// trip on that chain is paid in full (only 16 blocks fit per CU: the 8 KiB table is the LDS budget):
//   * the input around the cursor lives in a 512-byte LDS window (16 B/lane coalesced refills): probe
//     bytes, the cursor side of the backward/forward extension, the hash of cur-2 and the literal
//     bytes are LDS reads;
//   * each probing lane issues ONE 32-byte load around its candidate (8 bytes before, 24 after): the
//     4-byte verification, up to 8 bytes of backward and 20 bytes of forward extension come out of
//     that single round trip; only longer matches go back to memory (64 bytes per extra round trip);
//   * the compressed bytes are assembled in a 384-byte LDS stage and written back 16 bytes per lane.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lz4_device.h"

namespace lz4flex_dev {
#ifdef LZ4FLEX_PROFILE_PHASES
__device__ unsigned long long g2_phase_cycles[8];
__device__ unsigned long long g2_phase_counts[8];
#define C2_PHASE_DECL unsigned long long _pt = __builtin_readcyclecounter(); unsigned long long _pacc[8] = {0,0,0,0,0,0,0,0}; unsigned _pcnt[8] = {0,0,0,0,0,0,0,0};
#define C2_PHASE_MARK(k) { const unsigned long long _n = __builtin_readcyclecounter(); _pacc[k] += _n - _pt; _pcnt[k]++; _pt = _n; }
#define C2_PHASE_FLUSH if (threadIdx.x == 0) { for (int _k = 0; _k < 8; ++_k) { atomicAdd(&g2_phase_cycles[_k], _pacc[_k]); atomicAdd(&g2_phase_counts[_k], (unsigned long long)_pcnt[_k]); } }
#else
#define C2_PHASE_DECL
#define C2_PHASE_MARK(k)
#define C2_PHASE_FLUSH
#endif
namespace c2 {

constexpr uint32_t G = 8;
constexpr uint32_t TBL_BYTES = 8192;      // 4096 x u16
constexpr uint32_t WIN = 512;             // input window
constexpr uint32_t WIN_PAD = 32;
constexpr uint32_t STG = 384;             // output stage
constexpr uint32_t STG_PAD = 32;
constexpr uint32_t GROUP_LDS = TBL_BYTES + WIN + WIN_PAD + STG + STG_PAD;   // 9152
static_assert(GROUP_LDS % 16 == 0, "alignment");

__device__ __forceinline__ uint32_t ld32(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
__device__ __forceinline__ uint64_t ld64(const uint8_t* p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }
__device__ __forceinline__ void st32(uint8_t* p, uint32_t v) { __builtin_memcpy(p, &v, 4); }
__device__ __forceinline__ uint4 ld128(const uint8_t* p) { uint4 v; __builtin_memcpy(&v, p, 16); return v; }
__device__ __forceinline__ void st128(uint8_t* p, uint4 v) { __builtin_memcpy(p, &v, 16); }

__device__ __forceinline__ uint32_t hidx4(uint32_t x) { return ((x * 2654435761u) >> 16) >> 4; }
__device__ __forceinline__ uint32_t hidx5(uint64_t x) { return (uint32_t)(((x << 24) * 889523592379ull) >> 52); }
__device__ __forceinline__ uint64_t max_output_size(uint32_t n) { return 20ull + ((uint64_t)n * 110ull) / 100ull; }
__device__ __forceinline__ uint32_t probe_pos(uint32_t base, uint32_t i) {
    const uint32_t q = 1u + (i >> 5), r = i & 31u;
    return base + 16u * q * (q - 1u) + r * q;
}
template <int N>
__device__ __forceinline__ uint32_t row_shr(uint32_t v, uint32_t fill) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)v, 0x110 + N, 0xF, 0xF, false);
}
template <int N>
__device__ __forceinline__ uint32_t row_shl(uint32_t v, uint32_t fill) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)v, 0x100 + N, 0xF, 0xF, false);
}
template <int N>
struct FwdConflict {
    static __device__ __forceinline__ uint32_t run(uint32_t idx, uint32_t g) {
        const uint32_t far = FwdConflict<N + 1>::run(idx, g);
        const uint32_t v = row_shr<N>(idx, 0xFFFFFFFFu);
        return (g >= (uint32_t)N && v == idx) ? (uint32_t)N : far;
    }
};
template <>
struct FwdConflict<(int)G> {
    static __device__ __forceinline__ uint32_t run(uint32_t, uint32_t) { return 0u; }
};
template <int N>
struct BwdConflict {
    static __device__ __forceinline__ bool run(uint32_t idx, uint32_t g, uint32_t last) {
        const uint32_t v = row_shl<N>(idx, 0xFFFFFFFFu);
        const bool hit = (g + (uint32_t)N <= last) && (g + (uint32_t)N < G) && v == idx;
        return hit || BwdConflict<N + 1>::run(idx, g, last);
    }
};
template <>
struct BwdConflict<(int)G> {
    static __device__ __forceinline__ bool run(uint32_t, uint32_t, uint32_t) { return false; }
};

struct Enc {
    const uint8_t* in;     // block input (global)
    uint8_t* out;          // block output (global)
    uint16_t* tbl;         // LDS
    uint8_t* win;          // LDS input window
    uint8_t* stg;          // LDS output stage
    uint32_t g, shift;     // lane in group, first lane of the group in the wave
    uint32_t n;
    uint32_t wlo, wend;    // window holds input positions [wlo, wend)
    uint32_t obase, opos;  // stage holds output positions [obase, opos); [0, obase) is written back

    __device__ __forceinline__ uint32_t ballot(bool p) const { return (uint32_t)(__ballot(p) >> shift) & 0xFFu; }
    __device__ __forceinline__ uint32_t bcast(uint32_t v, uint32_t src) const { return __shfl(v, (int)(shift + src)); }

    // ---- input window ------------------------------------------------------------------------------
    // make the window hold [lo, hi) (hi - lo <= WIN - 16); slides forward only
    __device__ __forceinline__ void window(uint32_t lo, uint32_t hi) {
        if (lo >= wlo && hi <= wend) return;
        const uint32_t new_lo = lo & ~15u;
        if (new_lo > wlo) {
            if (new_lo < wend) {
                const uint32_t sh = new_lo - wlo, keep = wend - new_lo;
                for (uint32_t i = 16u * g; i < keep; i += 16u * G) {
                    const uint4 v = *reinterpret_cast<const uint4*>(win + sh + i);
                    *reinterpret_cast<uint4*>(win + i) = v;
                }
            } else {
                wend = new_lo;
            }
            wlo = new_lo;
        }
        uint32_t want = wlo + WIN;
        if (want > n) want = n;
        for (uint32_t p = wend + 16u * g; p < want; p += 16u * G) {
            if (p + 16u <= n) st128(win + (p - wlo), ld128(in + p));
            else for (uint32_t k = p; k < n; ++k) win[k - wlo] = in[k];
        }
        wend = want;
    }
    // 8 input bytes at position p (window must cover [p, p+8) up to the pad)
    __device__ __forceinline__ uint64_t win64(uint32_t p) const {
        const uint32_t rel = p - wlo;
        const uint32_t* w = reinterpret_cast<const uint32_t*>(win + (rel & ~3u));
        const uint32_t a = w[0], b = w[1], c = w[2];
        const uint32_t sh = rel & 3u;
        return ((uint64_t)__builtin_amdgcn_alignbyte(c, b, sh) << 32) | __builtin_amdgcn_alignbyte(b, a, sh);
    }

    // 8 input bytes at position p from the window when it covers them, else from memory (bytes >= n read as 0)
    __device__ __forceinline__ uint64_t rd64(uint32_t p) const {
        if (p >= wlo && p + 8u <= wend) return win64(p);
        if (p + 8u <= n) return ld64(in + p);
        uint64_t v = 0ull;
        for (uint32_t k = 0u; k < 8u && p + k < n; ++k) v |= (uint64_t)in[p + k] << (8u * k);
        return v;
    }
    // 16 input bytes at position p (bytes >= n read as 0, never touching memory past the block)
    __device__ __forceinline__ uint4 rd128(uint32_t p) const {
        if (p + 16u <= n) return ld128(in + p);
        uint32_t w[4] = {0u, 0u, 0u, 0u};
        for (uint32_t k = 0u; k < 16u && p + k < n; ++k) w[k >> 2] |= (uint32_t)in[p + k] << (8u * (k & 3u));
        return make_uint4(w[0], w[1], w[2], w[3]);
    }

    // ---- output stage ------------------------------------------------------------------------------
    __device__ __forceinline__ void stage_flush(bool final) {
        const uint32_t fnew = final ? opos : (opos & ~15u);
        const uint32_t full = fnew & ~15u;
        for (uint32_t p = obase + 16u * g; p < full; p += 16u * G)
            st128(out + p, *reinterpret_cast<const uint4*>(stg + (p - obase)));
        if (final) for (uint32_t p = full + g; p < opos; p += G) out[p] = stg[p - obase];
        // keep the partial 16-byte unit at the front of the stage
        if (!final && full > obase) {
            const uint4 v = *reinterpret_cast<const uint4*>(stg + (full - obase));
            if (g == 0u) *reinterpret_cast<uint4*>(stg) = v;
            obase = full;
        }
    }
    __device__ __forceinline__ void stage_room(uint32_t need) {
        if (opos - obase + need > STG) stage_flush(false);
    }
    // literal run [ls, ls+len) -> stage / output (after token + length bytes were staged)
    __device__ __forceinline__ void put_literals(uint32_t ls, uint32_t len) {
        if (len <= 64u && ls >= wlo && ls + len <= wend) {
            stage_room(len + 8u);
            uint8_t* d = stg + (opos - obase);
            const uint8_t* s = win + (ls - wlo);
            for (uint32_t i = 4u * g; i < len; i += 4u * G) st32(d + i, ld32(s + i));   // wild <= 3 bytes into the pad
            opos += len;
        } else {
            // long run (or literals that left the window): stream input -> stage -> output in 256-byte pieces
            while (len != 0u) {
                stage_room(256u + 8u);
                const uint32_t m = len < 256u ? len : 256u;
                uint8_t* d = stg + (opos - obase);
                for (uint32_t i = 4u * g; i < m; i += 4u * G) {
                    if (i + 4u <= m) st32(d + i, ld32(in + ls + i));
                    else for (uint32_t k = i; k < m; ++k) d[k] = in[ls + k];
                }
                opos += m; ls += m; len -= m;
            }
        }
    }
    __device__ __forceinline__ void put_length_ext(uint32_t rem) {   // write_integer, compress.rs:224-233
        const uint32_t n255 = rem / 255u;
        uint32_t left = n255;
        while (left != 0u) {
            stage_room(128u + 8u);
            const uint32_t m = left < 128u ? left : 128u;
            uint8_t* d = stg + (opos - obase);
            for (uint32_t k = g; k < m; k += G) d[k] = 0xFFu;
            opos += m; left -= m;
        }
        stage_room(8u);
        if (g == 0u) stg[opos - obase] = (uint8_t)(rem - n255 * 255u);
        opos += 1u;
    }
    // token (+ literal length extension) + literals
    __device__ __forceinline__ void emit_literals(uint32_t ls, uint32_t lit_len, uint32_t token_low) {
        stage_room(8u);
        if (g == 0u) stg[opos - obase] = (uint8_t)(((lit_len < 15u ? lit_len : 15u) << 4) | token_low);
        opos += 1u;
        if (lit_len >= 15u) put_length_ext(lit_len - 15u);
        if (lit_len != 0u) put_literals(ls, lit_len);
    }

#ifdef LZ4FLEX_PROFILE_PHASES
    unsigned long long facc[4] = {0, 0, 0, 0};
    unsigned fcnt[4] = {0, 0, 0, 0};
#endif
    // ---- per-block state ----------------------------------------------------------------------------
    uint32_t base, i0, lit_start;      // probing origin, index of the next probe, start of the pending literals
    uint32_t end_check, limit, idx0;
    uint32_t use_h5, continuation, done;
    // a match whose forward extension continues over several steps
    uint32_t ext, ecur, ecnd, e_mstart, e_offset;

    __device__ __forceinline__ uint32_t hidx(uint64_t x) const { return use_h5 ? hidx5(x) : hidx4((uint32_t)x); }

    // token, literals, offset, match-length extension of one sequence; then the cursor moves to cur_end
    __device__ __forceinline__ void finalize(uint32_t mstart, uint32_t offset, uint32_t cur_end) {
        const uint32_t dl = cur_end - (mstart + 4u);              // duplicate_length (compress.rs:456)
        const uint32_t lit_len = mstart - lit_start;
        // table: cur-2 (compress.rs:460-461); its 8 input bytes come from the window when it covers them
        const uint32_t q = cur_end - 2u;
        uint64_t qx;
        if (q >= wlo && q + 8u <= wend) {
            const uint8_t* wq = win + (q - wlo);
            qx = ((uint64_t)ld32(wq + 4) << 32) | ld32(wq);
        } else {
            qx = rd64(q);
        }
        const uint32_t room = STG - (opos - obase);
        if (lit_len <= 12u && dl < 270u && room >= 24u && lit_start >= wlo && mstart <= wend) {
            // quick emit: <= 16 bytes, written straight into the stage (each later write overwrites the wild tail)
            uint8_t* d = stg + (opos - obase);
            const uint32_t token = (lit_len << 4) | (dl < 15u ? dl : 15u);
            if (g == 0u) d[0] = (uint8_t)token;
            if (4u * g < lit_len) st32(d + 1u + 4u * g, ld32(win + (lit_start - wlo) + 4u * g));
            const uint32_t tail = offset | ((dl - 15u) << 16);   // offset LE, then the single extension byte (if any)
            if (g == 0u) st32(d + 1u + lit_len, tail);
            opos += 1u + lit_len + 2u + (dl >= 15u ? 1u : 0u);
        } else {
            emit_literals(lit_start, lit_len, dl < 15u ? dl : 15u);
            stage_room(8u);
            if (g == 0u) { stg[opos - obase] = (uint8_t)(offset & 0xFFu); stg[opos - obase + 1u] = (uint8_t)(offset >> 8); }
            opos += 2u;
            if (dl >= 15u) put_length_ext(dl - 15u);
        }
        if (g == 0u) tbl[hidx(qx)] = (uint16_t)q;
        lit_start = cur_end;
        base = cur_end;
        i0 = 0u;
        ext = 0u;
    }
    __device__ __forceinline__ void terminate() {   // handle_last_literals, compress.rs:237-247
        emit_literals(lit_start, n - lit_start, 0u);
        stage_flush(true);
        done = 1u;
    }

    // generic continuation from a verified candidate (cur, cnd): any backward / forward length, from memory
    __device__ __forceinline__ void finish_generic(uint32_t cur, uint32_t cnd) {
        const uint32_t offset = cur - cnd;
        const uint32_t m4 = cur + 4u, c4 = cnd + 4u;
        for (;;) {                                                             // backtrack, compress.rs:442-448
            const bool ok = (cnd > g) && (cur > lit_start + g) && in[cur - 1u - g] == in[cnd - 1u - g];
            const uint32_t okm = ballot(ok);
            const uint32_t k = (uint32_t)__builtin_ctz(~okm);
            cur -= k; cnd -= k;
            if (k < G) break;
        }
        uint32_t dl = 0u;
        for (;;) {                                                             // count_same_bytes, :156-216
            const uint32_t a = m4 + dl + 8u * g;
            uint32_t c = 0u;
            if (a < limit) {
                const uint32_t rem = limit - a;
                const uint32_t b = c4 + dl + 8u * g;
                if (rem >= 8u) {
                    const uint64_t diff = ld64(in + a) ^ ld64(in + b);
                    c = diff ? (uint32_t)(__builtin_ctzll(diff) >> 3) : 8u;
                } else {
                    while (c < rem && in[a + c] == in[b + c]) ++c;
                }
            }
            const uint32_t part = ballot(c != 8u);
            if (part == 0u) { dl += 8u * G; continue; }
            const uint32_t f = (uint32_t)__builtin_ctz(part);
            dl += 8u * f + bcast(c, f);
            break;
        }
        finalize(cur, offset, m4 + dl);
    }

    // one probe batch entirely from memory (block start, probing sparser than the window, tiny blocks)
    __device__ __forceinline__ void generic_step() {
        const uint32_t i = i0 + g;
        const uint32_t p = probe_pos(base, i);
        const bool valid = p <= end_check;
        uint32_t idx = 0xFFFF0000u + g;
        uint32_t cand = 0u, cur4 = 0u;
        bool cand_ok = false;
        if (valid) {
            const uint64_t x = ld64(in + p);
            idx = hidx(x);
            cur4 = (uint32_t)x;
            cand = (uint32_t)tbl[idx];
            cand_ok = !continuation || cand != 0u || idx == idx0;
        }
        const uint32_t dconf = FwdConflict<1>::run(idx, g);
        if (dconf != 0u) { cand = probe_pos(base, i - dconf); cand_ok = true; }
        bool is_match = false;
        if (valid && cand_ok) is_match = ld32(in + cand) == cur4;
        const uint32_t mm = ballot(is_match);
        const uint32_t vm = ballot(valid);
        const uint32_t last = mm ? (uint32_t)__builtin_ctz(mm) : (G - 1u);
        if (valid && g <= last && !BwdConflict<1>::run(idx, g, last)) tbl[idx] = (uint16_t)p;
        if (mm == 0u) {
            if (vm != 0xFFu) terminate(); else i0 += G;
            return;
        }
        finish_generic(bcast(p, last), bcast(cand, last));
    }

    // ---- the fast step: window-resident probe batch / extension round, ONE memory round trip ---------------
    __device__ __forceinline__ void fast_step() {
#ifdef LZ4FLEX_PROFILE_PHASES
        unsigned long long _t0 = __builtin_readcyclecounter();
#define FS_MARK(k) { const unsigned long long _n = __builtin_readcyclecounter(); facc[k - 4] += _n - _t0; fcnt[k - 4]++; _t0 = _n; }
#else
#define FS_MARK(k)
#endif
        const bool ex = ext != 0u;
        // cursor-side bytes: PROBE lane g looks at [p-8, p+40), EXT lane g at [ecur+16g, +16)
        const uint32_t i = i0 + g;
        const uint32_t p = probe_pos(base, i);
        const bool valid = !ex && p <= end_check;
        const uint32_t ea = ecur + 16u * g;
        const uint32_t rb = (ex ? ea : p - 8u) - wlo;
        const uint8_t* wp = win + rb;
        uint32_t Cc[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) Cc[k] = ld32(wp + 4 * k);
        // ---- PROBE: hash, table, same-bucket forwarding
        uint32_t idx = 0xFFFF0000u + g;
        uint32_t cand = 0u;
        bool cand_ok = false;
        if (valid) {
            idx = hidx(((uint64_t)Cc[3] << 32) | Cc[2]);
            cand = (uint32_t)tbl[idx];
            cand_ok = !continuation || cand != 0u || idx == idx0;
        }
        const uint32_t dconf = FwdConflict<1>::run(idx, g);
        if (dconf != 0u) { cand = probe_pos(base, i - dconf); cand_ok = true; }
        const bool want = valid && cand_ok;
        FS_MARK(4)   // window reads + hash + table read + forward conflict
        // later lanes of the batch that hit the same bucket (bit k-1: lane g+k)
        uint32_t later = 0u;
        {
            const uint32_t v1 = row_shl<1>(idx, 0xFFFFFFFFu), v2 = row_shl<2>(idx, 0xFFFFFFFFu), v3 = row_shl<3>(idx, 0xFFFFFFFFu),
                           v4 = row_shl<4>(idx, 0xFFFFFFFFu), v5 = row_shl<5>(idx, 0xFFFFFFFFu), v6 = row_shl<6>(idx, 0xFFFFFFFFu),
                           v7 = row_shl<7>(idx, 0xFFFFFFFFu);
            later = (v1 == idx ? 1u : 0u) | (v2 == idx ? 2u : 0u) | (v3 == idx ? 4u : 0u) | (v4 == idx ? 8u : 0u) |
                    (v5 == idx ? 16u : 0u) | (v6 == idx ? 32u : 0u) | (v7 == idx ? 64u : 0u);
            later &= (1u << (G - 1u - g)) - 1u;          // only lanes of this group
        }
        // ---- the one memory round trip: candidate bytes [cand-8, cand+40) (PROBE) / [ecnd+16g, +16) (EXT)
        const uint32_t lo8 = cand >= 8u ? cand - 8u : 0u;
        const uint32_t shc = cand - lo8;
        const uint32_t ga = ex ? ecnd + 16u * g : lo8;
        uint4 A = make_uint4(0u, 0u, 0u, 0u), Bq = A, Dq = A;
        const bool a_fast = ga + 16u <= n;       // always true for PROBE lanes of blocks >= 128 B (lo8 + 16 <= n - 5)
        if ((want || ex) && a_fast) A = ld128(in + ga);
        if (want && lo8 + 32u <= n) Bq = ld128(in + lo8 + 16u);
        if (want && lo8 + 48u <= n) Dq = ld128(in + lo8 + 32u);
        if (ex && !a_fast) A = rd128(ga);        // block tail only
        FS_MARK(5)   // backward-conflict masks + issue of the candidate loads
        // ---- PROBE: verification dword = loaded bytes [shc, shc+4)
        uint32_t c4;
        {
            const uint32_t wi = shc >> 2, sb = shc & 3u;
            const uint32_t l0 = wi == 0u ? A.x : (wi == 1u ? A.y : A.z);
            const uint32_t l1 = wi == 0u ? A.y : (wi == 1u ? A.z : A.w);
            c4 = __builtin_amdgcn_alignbyte(l1, l0, sb);
        }
        const bool is_match = want && c4 == Cc[2];
        // ---- speculative extension of THIS lane's candidate (exact when shc == 8)
        uint32_t nb, common;
        bool more, back_more;
        {
            const uint64_t ab = ((uint64_t)Cc[1] << 32) | Cc[0];
            const uint64_t cb = ((uint64_t)A.y << 32) | A.x;
            const uint64_t xd = ab ^ cb;
            uint32_t maxb = p - lit_start;
            if (maxb > cand) maxb = cand;
            nb = xd ? (uint32_t)(__builtin_clzll(xd) >> 3) : 8u;
            if (nb > maxb) nb = maxb;
            back_more = nb == 8u && maxb > 8u;
            const uint64_t x0 = (((uint64_t)Cc[4] << 32) | Cc[3]) ^ (((uint64_t)Bq.x << 32) | A.w);
            const uint64_t x1 = (((uint64_t)Cc[6] << 32) | Cc[5]) ^ (((uint64_t)Bq.z << 32) | Bq.y);
            const uint64_t x2 = (((uint64_t)Cc[8] << 32) | Cc[7]) ^ (((uint64_t)Dq.x << 32) | Bq.w);
            const uint64_t x3 = (((uint64_t)Cc[10] << 32) | Cc[9]) ^ (((uint64_t)Dq.z << 32) | Dq.y);
            const uint32_t x4 = Cc[11] ^ Dq.w;
            common = x0 ? (uint32_t)(__builtin_ctzll(x0) >> 3)
                        : (x1 ? 8u + (uint32_t)(__builtin_ctzll(x1) >> 3)
                              : (x2 ? 16u + (uint32_t)(__builtin_ctzll(x2) >> 3)
                                    : (x3 ? 24u + (uint32_t)(__builtin_ctzll(x3) >> 3)
                                          : (x4 ? 32u + (uint32_t)(__builtin_ctz(x4) >> 3) : 36u))));
            const uint32_t have_f = lo8 + 48u <= n ? 36u : (lo8 + 32u <= n ? 20u : 4u);
            const uint32_t maxlen = limit - (p + 4u);               // p <= n-12 => >= 2
            uint32_t lim = have_f < maxlen ? have_f : maxlen;
            if (common > lim) common = lim;
            more = common == lim && lim < maxlen;
        }
        // ---- EXT: equal bytes of this lane's 16
        uint32_t ec;
        {
            const uint64_t y0 = (((uint64_t)Cc[1] << 32) | Cc[0]) ^ (((uint64_t)A.y << 32) | A.x);
            const uint64_t y1 = (((uint64_t)Cc[3] << 32) | Cc[2]) ^ (((uint64_t)A.w << 32) | A.z);
            ec = y0 ? (uint32_t)(__builtin_ctzll(y0) >> 3) : (y1 ? 8u + (uint32_t)(__builtin_ctzll(y1) >> 3) : 16u);
            const uint32_t rem = ea < limit ? limit - ea : 0u;
            if (ec > rem) ec = rem;
        }
        FS_MARK(6)   // wait for the candidate bytes + verification + speculative extension
        if (ex) {
            const uint32_t part = ballot(ec != 16u);
            if (part == 0u) { ecur += 16u * G; ecnd += 16u * G; return; }
            const uint32_t f = (uint32_t)__builtin_ctz(part);
            finalize(e_mstart, e_offset, ecur + 16u * f + bcast(ec, f));
            return;
        }
        const uint32_t mm = ballot(is_match);
        const uint32_t vm = ballot(valid);
        const uint32_t last = mm ? (uint32_t)__builtin_ctz(mm) : (G - 1u);
        // table stores of the executed probes (compress.rs:393), last writer per bucket only
        {
            const uint32_t upto = last > g ? last - g : 0u;          // later lanes that execute: g+1 .. last
            const bool superseded = (later & ((1u << upto) - 1u)) != 0u;
            if (valid && g <= last && !superseded) tbl[idx] = (uint16_t)p;
        }
        if (mm == 0u) {
            if (vm != 0xFFu) terminate(); else i0 += G;
            return;
        }
        FS_MARK(7)   // ballots + table stores
        const uint32_t pw = bcast(p, last), cw = bcast(cand, last);
        const uint32_t pk = bcast(nb | (common << 8) | (more ? 0x10000u : 0u) | (back_more ? 0x20000u : 0u) |
                                  (shc == 8u ? 0x40000u : 0u), last);
        if ((pk & 0x40000u) == 0u || (pk & 0x20000u) != 0u) { finish_generic(pw, cw); return; }
        const uint32_t mstart = pw - (pk & 0xFFu);
        const uint32_t cm = (pk >> 8) & 0xFFu;
        if (pk & 0x10000u) {           // every compared byte matched: keep extending, 128 bytes per step
            ext = 1u; ecur = pw + 4u + cm; ecnd = cw + 4u + cm; e_mstart = mstart; e_offset = pw - cw;
            return;
        }
        finalize(mstart, pw - cw, pw + 4u + cm);
    }

    // ---- the block -----------------------------------------------------------------------------------
    __device__ __forceinline__ int32_t run(uint32_t cap, uint32_t flags, uint32_t* produced) {
        if ((uint64_t)cap < max_output_size(n)) return LZ4FLEX_DEV_E_OUTPUT_TOO_SMALL;   // compress.rs:338-340
        wlo = 0u; wend = 0u; obase = 0u; opos = 0u; done = 0u; ext = 0u;
        ecur = ecnd = e_mstart = e_offset = 0u;
        if (n < 13u) {   // compress.rs:343-346
            emit_literals(0u, n, 0u);
            stage_flush(true);
            *produced = opos;
            return 0;
        }
        continuation = (flags & 1u) != 0u ? 1u : 0u;
        use_h5 = ((flags & 2u) != 0u || n >= 65535u) ? 1u : 0u;   // compress.rs:559-566; FrameEncoder: always hash5
        end_check = n - 12u;
        limit = n - 6u;
        {
            uint4* t4 = reinterpret_cast<uint4*>(tbl);
            for (uint32_t k = g; k < TBL_BYTES / 16u; k += G) t4[k] = make_uint4(0u, 0u, 0u, 0u);
        }
        idx0 = hidx(ld64(in));
        lit_start = 0u;
        base = continuation ? 0u : 1u;      // compress.rs:353-359 (see lz4_compress.hip)
        i0 = continuation ? 1u : 0u;
        const bool big_enough = n >= 128u;
        C2_PHASE_DECL
        for (;;) {
            C2_PHASE_MARK(0)
            // what the next step touches
            bool slow = false, need = false;
            uint32_t lo = 0u, hi = 0u;
            if (!done) {
                if (ext) { lo = ecur - 16u; hi = ecur + 16u * G + 48u; }   // ecur >= 20; keeps 16 bytes of history in the window
                else {
                    const uint32_t p0 = probe_pos(base, i0);
                    uint32_t pl = probe_pos(base, i0 + G - 1u);
                    if (pl > end_check) pl = end_check;
                    slow = !big_enough || p0 < 16u;
                    lo = p0 - 8u;
                    if (lit_start < lo && p0 - lit_start <= 64u) lo = lit_start;
                    hi = pl + 48u;
                }
                if (hi > n) hi = n;
                if (!slow) {
                    if (hi - (lo & ~15u) > WIN - 16u) slow = !ext;     // probing sparser than the window
                    else need = lo < wlo || hi > wend || STG - (opos - obase) < 64u;
                }
            }
            // a refill / write-back costs a memory round trip for the whole wavefront: when one group needs it,
            // every group that is at least half way through its window / stage does it in the same round trip
            if (__any(need)) {
                if (!done && !slow) {
                    if (lo < wlo || hi > wend || hi + 192u > wend) window(lo, hi);
                    if (STG - (opos - obase) < 192u) stage_flush(false);
                }
            }
            C2_PHASE_MARK(1)   // window / stage maintenance
            if (!done && slow) generic_step();
            C2_PHASE_MARK(2)   // generic steps
            if (!done && !slow) fast_step();
            C2_PHASE_MARK(3)   // fast steps
            if (done) break;
        }
        C2_PHASE_FLUSH
#ifdef LZ4FLEX_PROFILE_PHASES
        if (threadIdx.x == 0) for (int _k = 0; _k < 4; ++_k) { atomicAdd(&g2_phase_cycles[4 + _k], facc[_k]); atomicAdd(&g2_phase_counts[4 + _k], (unsigned long long)fcnt[_k]); }
#endif
        *produced = opos;
        return 0;
    }
};

__global__ void __launch_bounds__(64) lz4_compress_lds_kernel(CompressArgs a) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[(64 / G) * GROUP_LDS];
    const uint32_t lane = threadIdx.x;
    const uint32_t b = blockIdx.x * (64u / G) + lane / G;
    if (b >= a.n) return;
    Enc e;
    e.g = lane % G;
    e.shift = (lane / G) * G;
    uint8_t* base = lds + (lane / G) * GROUP_LDS;
    e.tbl = reinterpret_cast<uint16_t*>(base);
    e.win = base + TBL_BYTES;
    e.stg = e.win + WIN + WIN_PAD;
    e.in = a.in_base + a.in_off[b];
    e.out = a.out_base + a.out_off[b];
    e.n = a.in_len[b];
    uint32_t produced = 0u;
    const int32_t st = e.run(a.out_cap[b], a.flags ? a.flags[b] : 0u, &produced);
    if (e.g == 0u) {
        a.status[b] = st;
        a.out_len[b] = st == 0 ? produced : 0u;
    }
}

}  // namespace c2

// blocks <= 64 KiB only (u16 table); the caller routes larger blocks to launch_compress
hipError_t launch_compress_lds(const CompressArgs& a, hipStream_t s) {
    if (a.n == 0u) return hipSuccess;
    const uint32_t per_wg = 64u / c2::G;
    const uint32_t grid = (a.n + per_wg - 1u) / per_wg;
    hipLaunchKernelGGL(c2::lz4_compress_lds_kernel, dim3(grid), dim3(64), 0, s, a);
    return hipGetLastError();
}

}  // namespace lz4flex_dev

#ifdef LZ4FLEX_PROFILE_PHASES
extern "C" int lz4flex_debug_phase2(unsigned long long* cycles, unsigned long long* counts, int reset) {
    if (reset) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(lz4flex_dev::g2_phase_cycles), z, sizeof z);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(lz4flex_dev::g2_phase_counts), z, sizeof z);
        return 0;
    }
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(cycles, HIP_SYMBOL(lz4flex_dev::g2_phase_cycles), 64);
    (void)hipMemcpyFromSymbol(counts, HIP_SYMBOL(lz4flex_dev::g2_phase_counts), 64);
    return 0;
}
#endif
