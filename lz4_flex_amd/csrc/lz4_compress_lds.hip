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

    // ---- the block -----------------------------------------------------------------------------------
    __device__ __forceinline__ int32_t run(uint32_t cap, uint32_t flags, uint32_t* produced) {
        if ((uint64_t)cap < max_output_size(n)) return LZ4FLEX_DEV_E_OUTPUT_TOO_SMALL;   // compress.rs:338-340
        wlo = 0u; wend = 0u; obase = 0u; opos = 0u;
        if (n < 13u) {   // compress.rs:343-346
            emit_literals(0u, n, 0u);
            stage_flush(true);
            *produced = opos;
            return 0;
        }
        const bool continuation = (flags & 1u) != 0u;
        const bool use_h5 = (flags & 2u) != 0u || n >= 65535u;   // compress.rs:559-566; FrameEncoder: always hash5
        const uint32_t end_check = n - 12u;
        const uint32_t limit = n - 6u;
        {
            uint4* t4 = reinterpret_cast<uint4*>(tbl);
            for (uint32_t k = g; k < TBL_BYTES / 16u; k += G) t4[k] = make_uint4(0u, 0u, 0u, 0u);
        }
        window(0u, 64u < n ? 64u : n);
        const uint32_t idx0 = use_h5 ? hidx5(win64(0u)) : hidx4((uint32_t)win64(0u));
        uint32_t lit_start = 0u;
        uint32_t base = continuation ? 0u : 1u;
        uint32_t i0 = continuation ? 1u : 0u;
        for (;;) {
            // ------------------------------------------------------------------ probe batch (compress.rs:373-439)
            const uint32_t i = i0 + g;
            const uint32_t p = probe_pos(base, i);
            const bool valid = p <= end_check;
            // window: 8 bytes of history before the first probe .. 48 bytes after the last valid probe
            {
                const uint32_t p0 = probe_pos(base, i0);
                uint32_t pl = probe_pos(base, i0 + G - 1u);
                if (pl > end_check) pl = end_check;
                uint32_t lo = p0 >= 16u ? p0 - 16u : 0u;
                if (lit_start < lo && p0 - lit_start <= 128u) lo = lit_start;
                uint32_t hi = pl + 48u;
                if (hi > n) hi = n;
                if (hi - (lo & ~15u) <= WIN - 16u) window(lo, hi);
                else window(p0 >= 16u ? p0 - 16u : 0u, (p0 + 400u) < n ? p0 + 400u : n);   // very sparse probing: see below
            }
            uint32_t idx = 0xFFFF0000u + g;
            uint32_t cand = 0u, cur4 = 0u;
            bool cand_ok = false;
            if (valid) {
                const uint64_t x = rd64(p);                                 // window, or memory when probing is sparser than the window
                if (use_h5) idx = hidx5(x); else idx = hidx4((uint32_t)x);
                cur4 = (uint32_t)x;
                cand = (uint32_t)tbl[idx];
                cand_ok = !continuation || cand != 0u || idx == idx0;
            }
            const uint32_t dconf = FwdConflict<1>::run(idx, g);
            if (dconf != 0u) { cand = probe_pos(base, i - dconf); cand_ok = true; }
            // ONE 32-byte load around the candidate: [cand-8, cand+24)
            const bool want = valid && cand_ok;          // distance <= 65535 always holds for blocks <= 64 KiB
            uint32_t lo8 = cand >= 8u ? cand - 8u : 0u;  // first loaded position
            const uint32_t shc = cand - lo8;             // candidate's byte offset inside the loaded data (0..8)
            uint4 ca = make_uint4(0u, 0u, 0u, 0u), cb = make_uint4(0u, 0u, 0u, 0u);
            if (want) {
                ca = rd128(lo8);
                if (lo8 + 32u <= n) cb = ld128(in + lo8 + 16u);
            }
            // verification dword = loaded bytes [shc, shc+4)
            uint32_t c4;
            {
                const uint32_t wi = shc >> 2, sb = shc & 3u;
                const uint32_t l0 = wi == 0u ? ca.x : (wi == 1u ? ca.y : ca.z);
                const uint32_t l1 = wi == 0u ? ca.y : (wi == 1u ? ca.z : ca.w);
                c4 = __builtin_amdgcn_alignbyte(l1, l0, sb);
            }
            const bool is_match = want && c4 == cur4;
            const uint32_t mm = ballot(is_match);
            const uint32_t vm = ballot(valid);
            const uint32_t last = mm ? (uint32_t)__builtin_ctz(mm) : (G - 1u);
            if (valid && g <= last && !BwdConflict<1>::run(idx, g, last)) tbl[idx] = (uint16_t)p;
            if (mm == 0u) {
                if (vm != 0xFFu) break;        // ran past end_check: last literals
                i0 += G;
                continue;
            }
            // ------------------------------------------------------------------ the winning probe's data, group-wide
            uint32_t cur = bcast(p, last);
            uint32_t cnd = bcast(cand, last);
            const uint32_t sc = bcast(shc, last);
            uint32_t d0 = bcast(ca.x, last), d1 = bcast(ca.y, last), d2 = bcast(ca.z, last), d3 = bcast(ca.w, last);
            uint32_t d4 = bcast(cb.x, last), d5 = bcast(cb.y, last), d6 = bcast(cb.z, last), d7 = bcast(cb.w, last);
            const uint32_t have = ((cnd >= 8u ? cnd - 8u : 0u) + 32u <= n) ? 32u : 16u;   // loaded bytes
            const uint32_t offset = cur - cnd;
            // ------------------------------------------------------------------ backtrack (compress.rs:442-448)
            {
                // candidate side: loaded bytes [0, sc) are positions cnd-sc .. cnd-1 ; cursor side from the window
                uint32_t maxb = cur - lit_start;
                if (maxb > cnd) maxb = cnd;
                uint32_t nb = 0u;
                if (maxb != 0u) {
                    const uint64_t lo64 = ((uint64_t)d1 << 32) | d0;     // loaded bytes 0..7
                    // align so that byte 7 is position cnd-1: shift left by (8 - sc) bytes
                    const uint64_t cb8 = sc == 8u ? lo64 : (sc == 0u ? 0ull : (lo64 << (8u * (8u - sc))));
                    uint64_t ab8;
                    if (cur >= 8u && cur - 8u >= wlo && cur <= wend) ab8 = win64(cur - 8u);
                    else {   // fewer than 8 bytes of input before cur, or history left the window
                        ab8 = 0ull;
                        const uint32_t k0 = cur < 8u ? cur : 8u;
                        for (uint32_t k = 1u; k <= k0; ++k) {
                            const uint32_t q = cur - k;
                            const uint32_t byte = (q >= wlo && q < wend) ? win[q - wlo] : in[q];
                            ab8 |= (uint64_t)byte << (8u * (8u - k));
                        }
                    }
                    const uint64_t xd = ab8 ^ cb8;
                    nb = xd == 0ull ? 8u : (uint32_t)(__builtin_clzll(xd) >> 3);
                    const uint32_t cap8 = maxb < 8u ? maxb : 8u;
                    if (nb > cap8) nb = cap8;
                    if (nb > sc) nb = sc;
                    cur -= nb; cnd -= nb;
                    if (nb == 8u && maxb > 8u) {
                        // rare: more than 8 bytes of backward extension, continue byte-wise from memory
                        for (;;) {
                            const bool ok = (cnd > g) && (cur > lit_start + g) && in[cur - 1u - g] == in[cnd - 1u - g];
                            const uint32_t okm = ballot(ok);
                            const uint32_t k = (uint32_t)__builtin_ctz(~okm);
                            cur -= k; cnd -= k;
                            if (k < G) break;
                        }
                    }
                }
            }
            const uint32_t lit_len = cur - lit_start;
            const uint32_t mstart = cur;                              // match start after backtracking
            // ------------------------------------------------------------------ forward (count_same_bytes :156-216)
            // match start (after backtracking) + 4; the loaded candidate bytes from index sc+4 on are cnd0+4 ...
            const uint32_t m4 = bcast(p, last) + 4u;                  // un-backtracked cursor + 4
            uint32_t dl = 0u;
            {
                const uint32_t maxlen = limit > m4 ? limit - m4 : 0u;
                const uint32_t fwd_have = have - (sc + 4u);          // candidate bytes available after the 4 verified ones
                // candidate bytes [sc+4, ...) as up to three 64-bit pieces
                const uint32_t wi = (sc + 4u) >> 2, sb = (sc + 4u) & 3u;     // wi in 1..3
                const uint32_t e0 = wi == 1u ? d1 : (wi == 2u ? d2 : d3);
                const uint32_t e1 = wi == 1u ? d2 : (wi == 2u ? d3 : d4);
                const uint32_t e2 = wi == 1u ? d3 : (wi == 2u ? d4 : d5);
                const uint32_t e3 = wi == 1u ? d4 : (wi == 2u ? d5 : d6);
                const uint32_t e4 = wi == 1u ? d5 : (wi == 2u ? d6 : d7);
                const uint32_t e5 = wi == 1u ? d6 : (wi == 2u ? d7 : 0u);
                const uint32_t f0 = __builtin_amdgcn_alignbyte(e1, e0, sb), f1 = __builtin_amdgcn_alignbyte(e2, e1, sb);
                const uint32_t f2 = __builtin_amdgcn_alignbyte(e3, e2, sb), f3 = __builtin_amdgcn_alignbyte(e4, e3, sb);
                const uint32_t f4 = __builtin_amdgcn_alignbyte(e5, e4, sb);
                const uint64_t a0 = rd64(m4), a1 = rd64(m4 + 8u);
                const uint32_t a2 = (uint32_t)rd64(m4 + 16u);
                const uint64_t x0 = a0 ^ (((uint64_t)f1 << 32) | f0);
                const uint64_t x1 = a1 ^ (((uint64_t)f3 << 32) | f2);
                const uint32_t x2 = a2 ^ f4;
                uint32_t common = x0 ? (uint32_t)(__builtin_ctzll(x0) >> 3)
                                     : (x1 ? 8u + (uint32_t)(__builtin_ctzll(x1) >> 3)
                                           : (x2 ? 16u + (uint32_t)(__builtin_ctz(x2) >> 3) : 20u));
                uint32_t lim = fwd_have < 20u ? fwd_have : 20u;
                if (lim > maxlen) lim = maxlen;
                if (common > lim) common = lim;
                dl = common;
                if (common == lim && lim < maxlen) {
                    // every compared byte matched and more may follow: 8 bytes per lane per round trip
                    const uint32_t c0 = bcast(cand, last) + 4u;       // un-backtracked candidate + 4
                    for (;;) {
                        const uint32_t a = m4 + dl + 8u * g;
                        uint32_t c = 0u;
                        if (a < limit) {
                            const uint32_t rem = limit - a;
                            const uint32_t b = c0 + dl + 8u * g;
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
                }
            }
            cur = m4 + dl;
            dl = cur - (mstart + 4u);                                 // duplicate_length counts from the backtracked start + 4
            // ------------------------------------------------------------------ table: cur-2 (compress.rs:460-461)
            if (g == 0u) {
                const uint32_t q = cur - 2u;
                const uint64_t x = rd64(q);
                const uint32_t qi = use_h5 ? hidx5(x) : hidx4((uint32_t)x);
                tbl[qi] = (uint16_t)q;
            }
            // ------------------------------------------------------------------ emit (compress.rs:463-486)
            emit_literals(lit_start, lit_len, dl < 15u ? dl : 15u);
            stage_room(8u);
            if (g == 0u) { stg[opos - obase] = (uint8_t)(offset & 0xFFu); stg[opos - obase + 1u] = (uint8_t)(offset >> 8); }
            opos += 2u;
            if (dl >= 15u) put_length_ext(dl - 15u);
            lit_start = cur;
            base = cur;
            i0 = 0u;
        }
        emit_literals(lit_start, n - lit_start, 0u);   // handle_last_literals, compress.rs:237-247
        stage_flush(true);
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
