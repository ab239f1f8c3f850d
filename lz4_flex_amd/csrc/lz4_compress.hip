// lz4_compress.hip -- batched LZ4 block encoder for MI355X (gfx950, wave64).
//
// Replaces the per-block call lz4_flex::block::compress_into / compress_internal (reference
// src/block/compress.rs:318-489, table selection :554-568, hashes src/block/hashtable.rs:19-34)
// for MANY independent blocks at once, and produces THE SAME BYTES as the reference encoder:
// same hash, same 4096-entry direct-mapped table semantics (zero-initialised, last writer
// wins), same skip schedule, same backward/forward extension, same end-of-block rules.
//
// Work decomposition (MI355X-first):
//   * one GROUP of G lanes (G = 8 or 16: half / one DPP row of the wave) owns one block; the ENCODER
//     wavefront of a workgroup holds 64/G blocks.
//   * the block's 4096-entry hash table lives in LDS (u16 entries = 8 KiB for blocks <= 64 KiB, u32 = 16 KiB
//     above); all table traffic is LDS traffic.  The table is the resource that bounds blocks in flight per
//     CU (160 KiB LDS / 8 KiB, 16 in practice), and a block is a serial chain of ~3 500 steps: the kernel is
//     bound by the length of one step's dependency chain, not by bandwidth.
//   * the serial greedy probe loop is executed G probes at a time: probe i of a sequence sits at a position
//     that depends only on i (skip schedule), so lane g evaluates probe i0+g; the first verified candidate
//     (ballot + ctz) wins and only probes up to it update the table.  Two probes of one batch that fall in the
//     same bucket are resolved exactly (DPP row shifts): a later probe sees the earlier probe's position as
//     its candidate and only the last one is stored, which is what the serial loop does.
//   * a step is three memory round trips (candidate bytes; 8 bytes behind + 8*G bytes after the match on both
//     sides; the next step's probe bytes + the cur-2 update's bytes), every load issued unconditionally with a
//     clamped address and as early as its address is known.
//   * the encoder wavefront issues no stores: sequences go through an LDS queue to the EMITTER wavefront of the
//     workgroup, which writes tokens, literals, offsets, the block's length and status, and prefetches the
//     input stream ahead of the encoders (see EmitQ / emitter_wave / the MODE list below).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lz4_device.h"

namespace lz4flex_dev {

// Phase timing (tools/phase_profile.py builds a private copy with -DLZ4FLEX_PROFILE_PHASES; the shipped
// library has none of it): wave-level s_memtime deltas accumulated per code region.
#ifdef LZ4FLEX_PROFILE_PHASES
__device__ unsigned long long g_phase_cycles[8];
__device__ unsigned long long g_phase_counts[8];
#define PHASE_DECL unsigned long long _pt = __builtin_readcyclecounter(); unsigned long long _pacc[8] = {0,0,0,0,0,0,0,0}; unsigned _pcnt[8] = {0,0,0,0,0,0,0,0};
#define PHASE_MARK(k) { const unsigned long long _n = __builtin_readcyclecounter(); _pacc[k] += _n - _pt; _pcnt[k]++; _pt = _n; }
#define PHASE_COUNT(k) _pcnt[k]++;
#define PHASE_WAIT_VM asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#define PHASE_FLUSH if (threadIdx.x == 0) { for (int _k = 0; _k < 8; ++_k) { atomicAdd(&g_phase_cycles[_k], _pacc[_k]); atomicAdd(&g_phase_counts[_k], (unsigned long long)_pcnt[_k]); } }
#else
#define PHASE_DECL
#define PHASE_MARK(k)
#define PHASE_COUNT(k)
#define PHASE_WAIT_VM
#define PHASE_FLUSH
#endif

#define LZ4_MFLIMIT 12u
#define LZ4_END_OFFSET 6u
#define LZ4_MIN_LENGTH 13u
#define LZ4_MAX_DISTANCE 65535u

__device__ __forceinline__ uint32_t cld32(const uint8_t* p) {
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}
__device__ __forceinline__ uint64_t cld64(const uint8_t* p) {
    uint64_t v;
    __builtin_memcpy(&v, p, 8);
    return v;
}
__device__ __forceinline__ void cst32(uint8_t* p, uint32_t v) { __builtin_memcpy(p, &v, 4); }

// src/block/hashtable.rs:19-21, index = hash >> 4 (:52-53,78)
__device__ __forceinline__ uint32_t hidx4(uint32_t x) { return ((x * 2654435761u) >> 16) >> 4; }
// src/block/hashtable.rs:27-34 (little endian), index = hash >> 4
__device__ __forceinline__ uint32_t hidx5(uint64_t x) { return (uint32_t)(((x << 24) * 889523592379ull) >> 52); }

// src/block/compress.rs:588-590
__device__ __forceinline__ uint64_t max_output_size(uint32_t n) { return 20ull + ((uint64_t)n * 110ull) / 100ull; }

template <int N>
__device__ __forceinline__ uint32_t row_shr(uint32_t v, uint32_t fill) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)v, 0x110 + N, 0xF, 0xF, false);
}
template <int N>
__device__ __forceinline__ uint32_t row_shl(uint32_t v, uint32_t fill) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)v, 0x100 + N, 0xF, 0xF, false);
}

// position of probe `i` of a sequence whose probing started at `base`
// (src/block/compress.rs:367-378: step = (32 + i) >> 5)
__device__ __forceinline__ uint32_t probe_pos(uint32_t base, uint32_t i) {
    const uint32_t q = 1u + (i >> 5), r = i & 31u;   // q < 2^15 for any block below 4 GiB: operands fit 24 bits, the products 32
    return base + 16u * __umul24(q, q - 1u) + __umul24(r, q);
}

template <int G>
struct Grp {
    uint32_t g;       // lane within group
    uint32_t shift;   // first lane of the group within the wave
    __device__ __forceinline__ uint32_t ballot(bool p) const {
        return (uint32_t)(__ballot(p) >> shift) & ((G == 32) ? 0xFFFFFFFFu : ((1u << G) - 1u));
    }
    __device__ __forceinline__ uint32_t bcast(uint32_t v, uint32_t src) const { return __shfl(v, (int)(shift + src)); }
    // bcast for a source lane that is the same in every lane of the group: the value of lane `src`, OR-reduced over the
    // group with DPP (swap pairs, swap pair of pairs, mirror the half row[, mirror the row]) -- 4..5 VALU instructions
    // on the sequence's dependency chain instead of a ds_bpermute round trip through the LDS crossbar (~130 cycles)
    __device__ __forceinline__ uint32_t bcast_u(uint32_t v, uint32_t src) const {
        if (G != 8 && G != 16) return bcast(v, src);
        uint32_t x = g == src ? v : 0u;
        x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]
        x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]
        x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x141, 0xF, 0xF, true);   // row_half_mirror: lanes i <-> 7 - i of each 8
        if (G == 16) x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x140, 0xF, 0xF, true);   // row_mirror: i <-> 15 - i
        return x;
    }
};

// nearest EARLIER lane of the group with the same bucket: returns its distance (1..G-1) or 0
template <int G, int N>
struct FwdConflict {
    static __device__ __forceinline__ uint32_t run(uint32_t idx, uint32_t g) {
        const uint32_t far = FwdConflict<G, N + 1>::run(idx, g);
        const uint32_t v = row_shr<N>(idx, 0xFFFFFFFFu);
        return (g >= (uint32_t)N && v == idx) ? (uint32_t)N : far;   // nearer distance overrides
    }
};
template <int G>
struct FwdConflict<G, G> {
    static __device__ __forceinline__ uint32_t run(uint32_t, uint32_t) { return 0u; }
};
// is there a LATER lane (distance 1..G-1, lane index <= last) with the same bucket?
template <int G, int N>
struct BwdConflict {
    static __device__ __forceinline__ bool run(uint32_t idx, uint32_t g, uint32_t last) {
        const uint32_t v = row_shl<N>(idx, 0xFFFFFFFFu);
        const bool hit = (g + (uint32_t)N <= last) && ((g % G) + (uint32_t)N < (uint32_t)G) && v == idx;
        return hit || BwdConflict<G, N + 1>::run(idx, g, last);
    }
};
template <int G>
struct BwdConflict<G, G> {
    static __device__ __forceinline__ bool run(uint32_t, uint32_t, uint32_t) { return false; }
};

// group copy of literals: 4 bytes per lane per step, exact tail
template <int G>
__device__ __forceinline__ void lit_copy(uint8_t* dst, const uint8_t* src, uint32_t len, uint32_t g) {
    for (uint32_t i = 4u * g; i < len; i += 4u * G) {
        if (i + 4u <= len) cst32(dst + i, cld32(src + i));
        else for (uint32_t k = i; k < len; ++k) dst[k] = src[k];
    }
}

// token + literal-length extension + literals. Returns new output position. (compress.rs:463-478 / :237-247)
template <int G>
__device__ __forceinline__ uint32_t emit_literals(uint8_t* out, uint32_t o, const uint8_t* in, uint32_t lit_start,
                                                  uint32_t lit_len, uint32_t token_low, uint32_t g) {
    const uint32_t token = ((lit_len < 15u ? lit_len : 15u) << 4) | token_low;
    if (g == 0u) out[o] = (uint8_t)token;
    o += 1u;
    if (lit_len >= 15u) {   // write_integer, compress.rs:224-233
        uint32_t rem = lit_len - 15u;
        const uint32_t n255 = rem / 255u;
        for (uint32_t k = g; k < n255; k += G) out[o + k] = 0xFFu;
        o += n255;
        if (g == 0u) out[o] = (uint8_t)(rem - n255 * 255u);
        o += 1u;
    }
    lit_copy<G>(out + o, in + lit_start, lit_len, g);
    return o + lit_len;
}

// LDS-typed views: a generic pointer would make every access a FLAT instruction, which is as slow as a global one
typedef __attribute__((address_space(3))) uint8_t lds_u8;
typedef uint32_t __attribute__((address_space(3), aligned(1))) lds_u32_unaligned;
typedef uint32_t __attribute__((ext_vector_type(4))) u32x4;
typedef u32x4 __attribute__((address_space(3))) lds_u128;
typedef volatile u32x4 __attribute__((address_space(3))) lds_vu128;
typedef volatile uint32_t __attribute__((address_space(3))) lds_vu32;
typedef volatile uint64_t __attribute__((address_space(3))) lds_vu64;

// Sequence queue between the encoder wave and the emitter wave of a workgroup (one per block, in LDS).
// The encoder's chain is probe -> verify -> extend -> next probe; writing the sequence out (token, literals,
// offset, length bytes: compress.rs:463-486) is not on that chain, but every global store the encoder wave
// issues costs it ~80 issue cycles and its acknowledgement is waited for by the next load (one in-order
// vmcnt).  So the encoder only pushes {lit_start, lit_len, offset, match_len - 4} records; a second wavefront
// pops them and does all the output formatting and all the stores.
#ifndef LZ4_PF_AHEAD
#define LZ4_PF_AHEAD 2048u          // bytes of input kept warm ahead of the encoder
#endif
#ifndef LZ4_EM_SLEEP
#define LZ4_EM_SLEEP 8
#endif
#define LZ4_EQ_DEPTH 16u
#define LZ4_EQ_FINAL 0xFFFFFFFFu     // record.offset: "last literals" record, the block ends after it
#define LZ4_EQ_SELF 0xFFFFFFFFu      // head value: the encoder wrote the block's output and status itself
struct EmitQ {                       // LDS layout per block: 16 records of 16 B, then head, tail, prog, pad
    lds_u8* p;
    __device__ __forceinline__ uint32_t head() const { return *reinterpret_cast<lds_vu32*>(p + 16u * LZ4_EQ_DEPTH); }
    __device__ __forceinline__ void set_head(uint32_t v) const { *reinterpret_cast<lds_vu32*>(p + 16u * LZ4_EQ_DEPTH) = v; }
    __device__ __forceinline__ uint32_t tail() const { return *reinterpret_cast<lds_vu32*>(p + 16u * LZ4_EQ_DEPTH + 4u); }
    __device__ __forceinline__ void set_tail(uint32_t v) const { *reinterpret_cast<lds_vu32*>(p + 16u * LZ4_EQ_DEPTH + 4u) = v; }
    __device__ __forceinline__ uint32_t prog() const { return *reinterpret_cast<lds_vu32*>(p + 16u * LZ4_EQ_DEPTH + 8u); }
    __device__ __forceinline__ void set_prog(uint32_t v) const { *reinterpret_cast<lds_vu32*>(p + 16u * LZ4_EQ_DEPTH + 8u) = v; }
    __device__ __forceinline__ void put(uint32_t slot, uint32_t a, uint32_t b, uint32_t c, uint32_t d) const {
        const u32x4 v = {a, b, c, d};
        *reinterpret_cast<lds_vu128*>(p + 16u * (slot & (LZ4_EQ_DEPTH - 1u))) = v;
    }
    __device__ __forceinline__ u32x4 get(uint32_t slot) const {
        return *reinterpret_cast<lds_vu128*>(p + 16u * (slot & (LZ4_EQ_DEPTH - 1u)));
    }
};
#define LZ4_EQ_BYTES (16u * LZ4_EQ_DEPTH + 16u)
// MODE 4 adds, behind the queue, an LDS ring of the block's input around the encoder's position, written by the
// emitter wave (which then doubles as the "filler"): {lo, hi} (8 B, + 8 B pad) and RING + RING_PAD bytes.
#define LZ4_RG_SIZE 1024u       // ring bytes per block (power of two)
#define LZ4_RG_PAD 16u          // mirror of ring bytes [0,16): unaligned reads across the wrap
#define LZ4_RG_HIST 64u         // bytes kept behind the encoder's first probe
#define LZ4_RG_CHUNK 128u       // filler granularity: 8 lanes x 16 B
#define LZ4_EQ_BYTES_RING (LZ4_EQ_BYTES + 16u + LZ4_RG_SIZE + LZ4_RG_PAD)
struct InRing {                      // encoder-side view
    lds_u8* ctl;                     // {lo, hi}
    lds_u8* ring;
    __device__ __forceinline__ uint64_t window() const { return *reinterpret_cast<lds_vu64*>(ctl); }
    __device__ __forceinline__ void set_window(uint32_t lo, uint32_t hi) const {
        *reinterpret_cast<lds_vu64*>(ctl) = (uint64_t)lo | ((uint64_t)hi << 32);
    }
    // 8 bytes at any byte position: three ALIGNED dwords + two byte-funnel shifts (a misaligned ds_read stalls
    // the LDS pipe for much longer than that; the mirror pad keeps the third dword inside the allocation)
    __device__ __forceinline__ uint64_t ld64(uint32_t pos) const {
        const uint32_t o = pos & (LZ4_RG_SIZE - 1u);
        typedef uint32_t __attribute__((address_space(3))) lds_u32a;
        lds_u32a* a = reinterpret_cast<lds_u32a*>(ring + (o & ~3u));
        const uint32_t d0 = a[0], d1 = a[1], d2 = a[2];
        const uint32_t sh = o & 3u;
        const uint32_t v0 = __builtin_amdgcn_alignbyte(d1, d0, sh);
        const uint32_t v1 = __builtin_amdgcn_alignbyte(d2, d1, sh);
        return (uint64_t)v0 | ((uint64_t)v1 << 32);
    }
    __device__ __forceinline__ uint32_t ld8(uint32_t pos) const { return ring[pos & (LZ4_RG_SIZE - 1u)]; }
};
__device__ __forceinline__ InRing in_ring_of(const EmitQ& q) { return InRing{q.p + LZ4_EQ_BYTES, q.p + LZ4_EQ_BYTES + 16u}; }

// HM: hash selection known at compile time (0: 4-byte hash, 1: 5-byte hash) or per block at run time (2)
// EQ: sequences are pushed to the block's EmitQ (an emitter wave writes the output and the block's status)
// instead of being written here; the return value is then LZ4FLEX_DEV_QUEUED unless the block was handled inline.
#define LZ4FLEX_DEV_QUEUED 0x7FFFFFFF
// RG (with EQ): current-side bytes come from the block's LDS input ring whenever the whole wave's reads are covered
template <int G, typename TblT, int HM, bool EQ, bool RG>
__device__ __forceinline__ int32_t encode_block(const uint8_t* __restrict__ in, uint32_t n, uint8_t* __restrict__ out,
                                                uint32_t cap, uint32_t flags, TblT* tbl, const Grp<G> grp,
                                                uint32_t* produced, volatile uint32_t* progress, const EmitQ eq) {
    const uint32_t g = grp.g;
    if ((uint64_t)cap < max_output_size(n)) {   // compress.rs:338-340
        if (EQ && g == 0u) eq.set_head(LZ4_EQ_SELF);
        return LZ4FLEX_DEV_E_OUTPUT_TOO_SMALL;
    }
    uint32_t o = 0u;
    if (n < LZ4_MIN_LENGTH) {   // compress.rs:343-346
        o = emit_literals<G>(out, o, in, 0u, n, 0u, g);
        *produced = o;
        if (EQ && g == 0u) eq.set_head(LZ4_EQ_SELF);
        return 0;
    }
    uint32_t q_head = 0u, q_tail = 0u;   // q_tail: last value read of the emitter's tail (it only grows)
    const InRing rg = in_ring_of(eq);
    uint32_t rg_lo = 0u, rg_hi = 0u;     // last window read from the filler (the real one only moves forward)
    const bool frame_tbl = (flags & 2u) != 0u;        // FrameEncoder: HashTable4K + hash5 always
    const bool continuation = (flags & 1u) != 0u;     // table holds only unreachable entries; pos 0 is probed
    const bool use_h5 = HM == 2 ? (frame_tbl || n >= 65535u) : (HM == 1);   // compress.rs:559-566
    const uint32_t end_check = n - LZ4_MFLIMIT;       // compress.rs:349
    // zero the table (HashTable::new / clear)
    {
        uint4* t4 = reinterpret_cast<uint4*>(tbl);
        const uint32_t n16 = (4096u * (uint32_t)sizeof(TblT)) / 16u;
        for (uint32_t k = g; k < n16; k += G) t4[k] = make_uint4(0u, 0u, 0u, 0u);
    }
    const uint32_t idx0 = use_h5 ? hidx5(cld64(in)) : hidx4(cld32(in));
    // compress.rs:353-359: default mode seeds position 0 (a no-op store on a zeroed table) and starts at 1.
    // continuation mode probes position 0 first: it can only miss (every entry is unreachable), stores
    // 0 in its bucket and counts as probe #0 of the first sequence.
    uint32_t lit_start = 0u;
    uint32_t base = continuation ? 0u : 1u;   // probing origin of the current sequence
    uint32_t i0 = continuation ? 1u : 0u;     // index of the first probe of the next batch
    PHASE_DECL
    const uint32_t limit = n - LZ4_END_OFFSET;                                // matches end 6 bytes before the end
    // The 8 input bytes at this lane's probe position are loaded one step AHEAD (at the end of the previous
    // step, together with the bytes the cur-2 table update needs), so a step starts with its probe data
    // in flight or already there.  p + 8 <= n - 4 for every valid probe, so the load is always 8 bytes.
    uint64_t x = 0ull;
    {
        const uint32_t p0 = probe_pos(base, i0 + g);
        x = cld64(in + (p0 <= end_check ? p0 : 0u));
    }
    uint64_t x0 = 0ull;   // probe bytes of the FIRST batch of the current sequence: lanes 0 and 7 hold its first 15 literals
    uint32_t x0_base = 0xFFFFFFFFu;   // position lane 0's x0 was read from
    for (;;) {
        // ------------------------------------------------------------------ may this step read the ring?  (wave-uniform)
        bool fast = false;
        if (RG) {
            const uint32_t pfirst = probe_pos(base, i0);
            if (g == 0u) eq.set_prog(pfirst);
            // consecutive probe positions (i < 32) and every current-side read of the step inside [lo, hi)
            bool ok = i0 + G <= 32u && pfirst + (10u * G + 16u) <= rg_hi && (rg_lo == 0u || rg_lo + 8u <= pfirst);
            if (!__all(ok)) {
                const uint64_t snap = rg.window();
                rg_lo = (uint32_t)snap; rg_hi = (uint32_t)(snap >> 32);
                ok = i0 + G <= 32u && pfirst + (10u * G + 16u) <= rg_hi && (rg_lo == 0u || rg_lo + 8u <= pfirst);
            }
            fast = __all(ok);
            if (fast) { PHASE_COUNT(5) }
        }
        // ------------------------------------------------------------------ probe batch
        if (i0 == 0u) { x0 = x; x0_base = base; }
        const uint32_t i = i0 + g;
        const uint32_t p = probe_pos(base, i);
        const bool valid = p <= end_check;                                    // compress.rs:381
        uint32_t idx = 0xFFFF0000u + g;   // distinct sentinels for invalid lanes
        uint32_t cand = 0u;
        const uint32_t cur4 = (uint32_t)x;
        bool cand_ok = false;
        if (valid) {
            idx = use_h5 ? hidx5(x) : hidx4(cur4);
            cand = (uint32_t)tbl[idx];
            // a zero entry is "position 0": real in default mode, and in continuation
            // mode only in the bucket position 0 was stored to
            cand_ok = !continuation || cand != 0u || idx == idx0;
        }
        PHASE_MARK(0)   // loop top: probe bytes wait + hash + table read issue
        // The candidate's 4 bytes are requested straight from the table value.  Every load of a step is issued
        // unconditionally (lanes with nothing to read load position 0): a load inside a divergent branch is
        // waited for inside it, which turns one round trip into several.
        bool try_m = valid && cand_ok && (p - cand) <= LZ4_MAX_DISTANCE;        // compress.rs:403-405
        uint32_t cand4 = cld32(in + (try_m ? cand : 0u));
        // Same-bucket probes earlier in this batch supersede the table content (the serial loop would have stored
        // them first).  Resolved while the load is in flight; the superseding candidate is a probe position of
        // this very batch, so its 4 bytes are that lane's probe bytes - no second load.
        const uint32_t d = FwdConflict<G, 1>::run(idx, g);
        const bool anyd = __any(d != 0u);
        PHASE_MARK(1)   // table read wait + conflict resolution
        if (anyd) {
            const uint32_t src4 = grp.bcast(cur4, g - d);                     // lanes with d == 0 read themselves
            if (d != 0u) {
                cand = probe_pos(base, i - d);
                try_m = (p - cand) <= LZ4_MAX_DISTANCE;
                cand4 = src4;
            }
        }
        const bool is_match = (cand4 == cur4) && try_m;                         // compress.rs:432-438
        const uint32_t mm = grp.ballot(is_match);
        PHASE_MARK(2)   // candidate round trip + verify
        const uint32_t vm = grp.ballot(valid);
        const uint32_t last = mm ? (uint32_t)__builtin_ctz(mm) : (G - 1u);     // last probe that executes
        // table stores of the executed probes (compress.rs:393), last writer per bucket only.  A later lane
        // with the same bucket exists only if some lane saw an earlier one (d != 0): skip the scan otherwise.
        bool superseded = false;
        if (anyd) superseded = BwdConflict<G, 1>::run(idx, g, last);
        if (valid && g <= last && !superseded) tbl[idx] = (TblT)p;
        if (mm == 0u) {
            if (vm != (((G == 32) ? 0xFFFFFFFFu : ((1u << G) - 1u)))) break;   // ran past end_check: last literals
            i0 += G;
            const uint32_t pn = probe_pos(base, i0 + g);
            if (RG && fast && i0 + G <= 32u && __all(probe_pos(base, i0) + G + 8u <= rg_hi)) x = rg.ld64(pn);
            else x = cld64(in + (pn <= end_check ? pn : 0u));
            continue;
        }
        uint32_t cur = grp.bcast_u(p, last);
        uint32_t cnd = grp.bcast_u(cand, last);
        const uint32_t offset = cur - cnd;                                    // compress.rs:409
        PHASE_MARK(3)   // table stores + winner broadcast
        // ------------------------------------------------------------------ extension: ONE memory round trip for the
        // first G bytes backwards and the first 8*G bytes forwards (the forward count starts at the verified
        // position + 4 whatever the backtrack finds: the bytes in between are known equal)
        const uint32_t m4 = cur + 4u, c4 = cnd + 4u;
        const bool bk_ok0 = (cnd > g) && (cur > lit_start + g);
        const uint32_t fa = m4 + 8u * g, fb = c4 + 8u * g;
        const bool f8 = fa + 8u <= limit;                                     // a full 8-byte forward chunk
        uint32_t bka;
        uint64_t fwa;
        if (RG && fast) {                     // current side from the ring (lanes with nothing to read re-read cur)
            bka = rg.ld8(bk_ok0 ? cur - 1u - g : cur);
            fwa = rg.ld64(f8 ? fa : cur);
        } else {
            bka = in[bk_ok0 ? cur - 1u - g : 0u];
            fwa = cld64(in + (f8 ? fa : 0u));
        }
        const uint32_t bkb = in[bk_ok0 ? cnd - 1u - g : 0u];
        const uint64_t fdiff = fwa ^ cld64(in + (f8 ? fb : 0u));
        uint32_t c = 0u;                      // equal bytes seen by this lane in the first forward round (0..8)
        if (f8) c = fdiff ? (uint32_t)(__builtin_ctzll(fdiff) >> 3) : 8u;
        if (__any(!f8 && fa < limit)) {       // rare: the last (< 8 byte) chunk before the end of the block
            if (!f8 && fa < limit) {
                const uint32_t rem = limit - fa;
                while (c < rem && in[fa + c] == in[fb + c]) ++c;
            }
        }
        const uint32_t okm = grp.ballot(bk_ok0 && bka == bkb);
        PHASE_MARK(4)   // extension round trip
        // ---- forward (count_same_bytes :156-216)
        uint32_t dl = 0u;
        {
            uint32_t part = grp.ballot(c != 8u);
            if (part != 0u) {
                const uint32_t f = (uint32_t)__builtin_ctz(part);
                dl = 8u * f + grp.bcast_u(c, f);
            } else {
                dl = 8u * G;
                for (;;) {
                    const uint32_t a = m4 + dl + 8u * g;
                    uint32_t c2 = 0u;
                    if (a < limit) {
                        const uint32_t rem = limit - a;
                        const uint32_t b = c4 + dl + 8u * g;
                        if (rem >= 8u) {
                            const uint64_t diff = cld64(in + a) ^ cld64(in + b);
                            c2 = diff ? (uint32_t)(__builtin_ctzll(diff) >> 3) : 8u;
                        } else {
                            while (c2 < rem && in[a + c2] == in[b + c2]) ++c2;
                        }
                    }
                    part = grp.ballot(c2 != 8u);
                    if (part == 0u) { dl += 8u * G; continue; }
                    const uint32_t f = (uint32_t)__builtin_ctz(part);
                    dl += 8u * f + grp.bcast_u(c2, f);
                    break;
                }
            }
        }
        const uint32_t cur_end = m4 + dl;     // the forward count starts at the verified position + 4 whatever the backtrack finds
        // ------------------------------------------------------------------ requests for the NEXT step: the bytes of
        // the cur-2 table update (compress.rs:460-461) and of the first probe batch after this match
        const uint32_t q = cur_end - 2u;
        uint64_t qx, xn;                                                      // q + 8 <= n: matches end >= 6 bytes early
        {
            const bool pv = cur_end + g <= end_check;
            if (RG && fast && __all(cur_end + G + 8u <= rg_hi)) {
                qx = rg.ld64(q);
                xn = rg.ld64(pv ? cur_end + g : q);
            } else {
                qx = cld64(in + q);
                xn = cld64(in + (pv ? cur_end + g : 0u));
            }
        }
        // ---- backtrack (compress.rs:442-448), off the critical path: the next step only needs cur_end
        {
            uint32_t nb = (uint32_t)__builtin_ctz(~okm);                      // G..31 bits are 0 in okm => nb <= G
            cur -= nb; cnd -= nb;
            while (nb == (uint32_t)G) {                                       // rare: more than G bytes backwards
                const bool ok = (cnd > g) && (cur > lit_start + g) && in[cur - 1u - g] == in[cnd - 1u - g];
                const uint32_t okm2 = grp.ballot(ok);
                nb = (uint32_t)__builtin_ctz(~okm2);
                cur -= nb; cnd -= nb;
            }
        }
        const uint32_t lit_len = cur - lit_start;                             // compress.rs:451
        dl = cur_end - (cur + 4u);                                            // duplicate_length counts from the backtracked start + 4
        if (EQ) {
            // ---- hand the sequence to the emitter wave (compress.rs:463-486 happen there)
            while (__any(q_head - q_tail >= LZ4_EQ_DEPTH)) {   // looks full: refresh the tail, wait if it really is (rare)
                const uint32_t t = eq.tail();
                if (t == q_tail) __builtin_amdgcn_s_sleep(1);
                q_tail = t;
            }
            if (g == 0u) {
                eq.put(q_head, lit_start, lit_len, offset, dl);
                eq.set_head(q_head + 1u);
                if (!RG) eq.set_prog(cur_end);
            }
            q_head += 1u;
        } else {
            // ------------------------------------------------------------------ emit (compress.rs:463-486)
            if (x0_base == lit_start && lit_len <= 14u && dl < 270u && o + 20u <= cap) {
                // short literal run: the literals sit in the first batch's probe bytes (lane 0: bytes 0..7 of the run,
                // lane 7: bytes 7..14).  token+literals as 8-byte stores, then offset + length byte as one 4-byte
                // store (bytes past the sequence are rewritten by the next one; the capacity check above keeps
                // them inside `out`)
                const uint32_t tk = (lit_len << 4) | (dl < 15u ? dl : 15u);
                const uint64_t w0 = (uint64_t)tk | (x0 << 8);
                const uint32_t w1 = offset | ((dl - 15u) << 16);
#ifndef LZ4FLEX_ABL_NOSTORE
                if (g == 0u) __builtin_memcpy(out + o, &w0, 8);
                if (G >= 8 && g == 7u && lit_len > 7u) __builtin_memcpy(out + o + 8u, &x0, 8);
                if (g == 0u) __builtin_memcpy(out + o + 1u + lit_len, &w1, 4);
#endif
                o += 3u + lit_len + (dl >= 15u ? 1u : 0u);
#ifdef LZ4FLEX_ABL_NOGENERIC
            } else if (false) {
#else
            } else {
#endif
                o = emit_literals<G>(out, o, in, lit_start, lit_len, dl < 15u ? dl : 15u, g);
                if (g == 0u) { out[o] = (uint8_t)(offset & 0xFFu); out[o + 1u] = (uint8_t)(offset >> 8); }
                o += 2u;
                if (dl >= 15u) {
                    const uint32_t rem = dl - 15u;
                    const uint32_t n255 = rem / 255u;
                    for (uint32_t k = g; k < n255; k += G) out[o + k] = 0xFFu;
                    o += n255;
                    if (g == 0u) out[o] = (uint8_t)(rem - n255 * 255u);
                    o += 1u;
                }
            }
        }
        PHASE_MARK(6)   // next-step requests + emit
        PHASE_WAIT_VM
        PHASE_MARK(7)   // what is left of the next-step round trip after the emit
#ifndef LZ4FLEX_ABL_NOQ
        if (g == 0u) tbl[use_h5 ? hidx5(qx) : hidx4((uint32_t)qx)] = (TblT)q;
#endif
        lit_start = cur_end;                                                  // compress.rs:487
        base = cur_end;
        i0 = 0u;
        if (progress && g == 0u) *progress = cur_end;
        x = xn;
    }
    // handle_last_literals, compress.rs:237-247
    if (EQ) {
        while (__any(q_head - q_tail >= LZ4_EQ_DEPTH)) {
            const uint32_t t = eq.tail();
            if (t == q_tail) __builtin_amdgcn_s_sleep(1);
            q_tail = t;
        }
        if (g == 0u) {
            eq.put(q_head, lit_start, n - lit_start, LZ4_EQ_FINAL, 0u);
            eq.set_head(q_head + 1u);
        }
        PHASE_FLUSH
        return LZ4FLEX_DEV_QUEUED;
    }
    o = emit_literals<G>(out, o, in, lit_start, n - lit_start, 0u, g);
    PHASE_FLUSH
    *produced = o;
    return 0;
}

// The emitter wave: lane group j pops block j's sequence records and writes the compressed block
// (compress.rs:463-486, :237-247), then the block's length and status.  It also walks ahead of the encoder
// touching the input lines it is about to read (see MODE 2 below).
template <int G, bool RG>
__device__ __forceinline__ void emitter_wave(const CompressArgs& a, uint32_t b, bool live, const EmitQ eq, const Grp<G> grp) {
    const uint32_t g = grp.g;
    const uint8_t* in = a.in_base + (live ? a.in_off[b] : 0ull);
    uint8_t* out = a.out_base + (live ? a.out_off[b] : 0ull);
    const uint32_t n = live ? a.in_len[b] : 0u;
    uint32_t o = 0u, tail = 0u, pf = 0u, acc = 0u;
    const InRing rg = in_ring_of(eq);
    const uint32_t n_fill = n & ~(LZ4_RG_CHUNK - 1u);   // whole chunks only; the encoder reads the tail from memory
    uint32_t rlo = 0u, rhi = 0u;                       // the ring holds input positions [rlo, rhi)
    bool done = !live;
    for (;;) {
        if (!__any(!done)) break;
        bool worked = false;
        if (!done) {
            const uint32_t head = eq.head();
            if (head == LZ4_EQ_SELF) {
                done = true;
            } else if (tail != head) {
                const u32x4 r = eq.get(tail);
                const uint32_t lit_start = r.x, lit_len = r.y, offset = r.z, dl = r.w;
                if (offset == LZ4_EQ_FINAL) {
                    o = emit_literals<G>(out, o, in, lit_start, lit_len, 0u, g);
                    if (g == 0u) { a.status[b] = 0; a.out_len[b] = o; }
                    done = true;
                } else {
                    o = emit_literals<G>(out, o, in, lit_start, lit_len, dl < 15u ? dl : 15u, g);
                    if (g == 0u) { out[o] = (uint8_t)(offset & 0xFFu); out[o + 1u] = (uint8_t)(offset >> 8); }
                    o += 2u;
                    if (dl >= 15u) {   // write_integer, compress.rs:224-233
                        const uint32_t rem = dl - 15u;
                        const uint32_t n255 = rem / 255u;
                        for (uint32_t k = g; k < n255; k += G) out[o + k] = 0xFFu;
                        o += n255;
                        if (g == 0u) out[o] = (uint8_t)(rem - n255 * 255u);
                        o += 1u;
                    }
                }
                tail += 1u;
                if (g == 0u) eq.set_tail(tail);
                worked = true;
            }
            const uint32_t pos = eq.prog();
            if (!RG) {
                // stream prefetch: keep the lines [pos, pos + LZ4_PF_AHEAD) of the input on their way into L2
                if (pf < pos) pf = pos & ~127u;
                if (!done && pf + 128u * G < pos + LZ4_PF_AHEAD) {
                    const uint32_t at = pf + 128u * g;
                    if (at + 4u <= n) acc += *reinterpret_cast<const volatile uint32_t*>(in + (at & ~3u));
                    pf += 128u * G;
                }
            }
        }
        if (RG) {
            // ---- input ring: one 128-byte chunk per block and iteration, lanes 0..7 of the group 16 B each.
            // The chunk [rhi, rhi+128) may overwrite ring positions below prog - HIST only; {lo, hi} are published
            // after the data (a wave's LDS operations execute in order).
            bool fill = false;
            if (!done) {
                const uint32_t pos = eq.prog();
                if (pos >= rhi) {   // the ring is behind the encoder (start, long match): restart it around pos
                    const uint32_t back = pos < LZ4_RG_HIST ? pos : LZ4_RG_HIST;
                    rlo = rhi = (pos - back) & ~(LZ4_RG_CHUNK - 1u);
                }
                fill = rhi + LZ4_RG_CHUNK <= n_fill && rhi + LZ4_RG_CHUNK + LZ4_RG_HIST <= pos + LZ4_RG_SIZE;
            }
            if (__any(fill)) {
                const bool mine = fill && g < 8u;
                u32x4 v = {0u, 0u, 0u, 0u};
                __builtin_memcpy(&v, in + (mine ? rhi + 16u * g : 0u), 16);
                if (mine) {
                    const uint32_t ro = (rhi + 16u * g) & (LZ4_RG_SIZE - 1u);
                    *reinterpret_cast<lds_u128*>(rg.ring + ro) = v;
                    if (ro == 0u) *reinterpret_cast<lds_u128*>(rg.ring + LZ4_RG_SIZE) = v;
                }
                if (fill) {
                    rhi += LZ4_RG_CHUNK;
                    if (rhi - rlo > LZ4_RG_SIZE) rlo = rhi - LZ4_RG_SIZE;
                    if (g == 0u) rg.set_window(rlo, rhi);
                    worked = true;
                }
            }
        }
        if (!__any(worked)) __builtin_amdgcn_s_sleep(LZ4_EM_SLEEP);
    }
    if (acc == 0x9E3779B9u && live) eq.set_prog(acc);   // keeps the prefetch loads alive
}


// ---------------------------------------------------------------------------------------------------
// General encoder: the full signature of compress_internal<T, USE_DICT> (src/block/compress.rs:318-489):
// a prefix before input_pos, an external dictionary, a stream offset and a table that PERSISTS across the
// blocks of a chain.  Used by block::compress_into_with_dict (init_dict, :571-583) and by Linked frames
// (src/frame/compress.rs:280-299, :327-356), whose blocks depend on each other and therefore run one after
// another inside one group (a "chain").  Table entries are u32 stream positions (HashTable4K semantics).
struct ChainBlock {
    uint64_t in_off;      // start of `input` (prefix included) in in_base
    uint64_t dict_off;    // start of ext_dict in in_base
    uint32_t in_len;      // input.len()
    uint32_t in_pos;      // input_pos: first byte to compress
    uint32_t dict_len;
    uint32_t so;          // input_stream_offset
    uint32_t repos;       // HashTable4K::reposition(repos) before this block (hashtable.rs:113-117); 0 = none
    uint32_t flags;       // bit0: 4-byte hash (HashTable4KU16 path of compress_into_with_dict); bit1: clear table + init_dict
};

template <int G>
__device__ __forceinline__ int32_t encode_general(const uint8_t* __restrict__ in, uint32_t n, uint32_t ipos,
                                                  const uint8_t* __restrict__ dict, uint32_t dict_len, uint32_t so,
                                                  bool use_h4, uint8_t* __restrict__ out, uint32_t cap, uint32_t* tbl,
                                                  const Grp<G> grp, uint32_t* produced) {
    const uint32_t g = grp.g;
    const bool use_dict = dict_len != 0u;
    if ((uint64_t)cap < max_output_size(n - ipos)) return LZ4FLEX_DEV_E_OUTPUT_TOO_SMALL;   // :338-340
    uint32_t o = 0u;
    if (n - ipos < LZ4_MIN_LENGTH) {                                                       // :343-346
        o = emit_literals<G>(out, o, in, ipos, n - ipos, 0u, g);
        *produced = o;
        return 0;
    }
    const uint32_t eds = so - dict_len;                // ext_dict_stream_offset, :348
    const uint32_t end_check = n - LZ4_MFLIMIT;
    uint32_t lit_start = ipos;
    uint32_t base = ipos;
    if (ipos == 0u && so == 0u) {                      // :353-359
        if (g == 0u) tbl[use_h4 ? hidx4(cld32(in)) : hidx5(cld64(in))] = 0u;
        base = 1u;
    }
    uint32_t i0 = 0u;
    for (;;) {
        const uint32_t i = i0 + g;
        const uint32_t p = probe_pos(base, i);
        const bool valid = p <= end_check;
        uint32_t idx = 0xFFFF0000u + g;
        uint32_t cs = 0u, cur4 = 0u;                   // candidate as a stream position
        if (valid) {
            if (use_h4) { cur4 = cld32(in + p); idx = hidx4(cur4); }
            else { const uint64_t x = cld64(in + p); idx = hidx5(x); cur4 = (uint32_t)x; }
            cs = tbl[idx];
        }
        const uint32_t d = FwdConflict<G, 1>::run(idx, g);
        if (d != 0u) cs = probe_pos(base, i - d) + so;
        const uint32_t ps = p + so;
        const bool in_input = cs >= so;                                        // :407-411
        const bool in_dict = !in_input && use_dict && cs >= eds;               // :412-421
        const uint32_t cand = in_input ? cs - so : cs - eds;
        const uint8_t* src = in_input ? in : dict;
        bool is_match = false;
        if (valid && (ps - cs) <= LZ4_MAX_DISTANCE && (in_input || in_dict))   // :403-405, :422-429
            is_match = cld32(src + cand) == cur4;                              // :432-438
        const uint32_t mm = grp.ballot(is_match);
        const uint32_t vm = grp.ballot(valid);
        const uint32_t last = mm ? (uint32_t)__builtin_ctz(mm) : (G - 1u);
        if (valid && g <= last && !BwdConflict<G, 1>::run(idx, g, last)) tbl[idx] = ps;
        if (mm == 0u) {
            if (vm != ((1u << G) - 1u)) break;
            i0 += G;
            continue;
        }
        uint32_t cur = grp.bcast(p, last);
        uint32_t cnd = grp.bcast(cand, last);
        const bool m_in = grp.bcast(in_input ? 1u : 0u, last) != 0u;
        const uint8_t* msrc = m_in ? in : dict;
        const uint32_t msrc_len = m_in ? n : dict_len;
        const uint32_t offset = (cur + so) - grp.bcast(cs, last);             // :409 / :419
        for (;;) {                                                             // backtrack :442-448
            const bool ok = (cnd > g) && (cur > lit_start + g) && in[cur - 1u - g] == msrc[cnd - 1u - g];
            const uint32_t okm = grp.ballot(ok);
            const uint32_t nb = (uint32_t)__builtin_ctz(~okm);
            cur -= nb; cnd -= nb;
            if (nb < (uint32_t)G) break;
        }
        const uint32_t lit_len = cur - lit_start;
        cur += 4u; cnd += 4u;
        // count_same_bytes :156-216: bounded by the input end - 6 and by the end of the candidate's buffer
        const uint32_t max_in = n - LZ4_END_OFFSET > cur ? n - LZ4_END_OFFSET - cur : 0u;
        const uint32_t max_c = msrc_len - cnd;
        const uint32_t max_m = max_in < max_c ? max_in : max_c;
        uint32_t dl = 0u;
        for (;;) {
            const uint32_t k = dl + 8u * g;
            uint32_t c = 0u;
            if (k < max_m) {
                const uint32_t rem = max_m - k;
                if (rem >= 8u) {
                    const uint64_t diff = cld64(in + cur + k) ^ cld64(msrc + cnd + k);
                    c = diff ? (uint32_t)(__builtin_ctzll(diff) >> 3) : 8u;
                } else {
                    while (c < rem && in[cur + k + c] == msrc[cnd + k + c]) ++c;
                }
            }
            const uint32_t part = grp.ballot(c != 8u);
            if (part == 0u) { dl += 8u * G; continue; }
            const uint32_t f = (uint32_t)__builtin_ctz(part);
            dl += 8u * f + grp.bcast(c, f);
            break;
        }
        cur += dl;
        if (g == 0u) {                                                         // :460-461
            const uint32_t q = cur - 2u;
            tbl[use_h4 ? hidx4(cld32(in + q)) : hidx5(cld64(in + q))] = q + so;
        }
        o = emit_literals<G>(out, o, in, lit_start, lit_len, dl < 15u ? dl : 15u, g);
        if (g == 0u) { out[o] = (uint8_t)(offset & 0xFFu); out[o + 1u] = (uint8_t)(offset >> 8); }
        o += 2u;
        if (dl >= 15u) {
            const uint32_t rem = dl - 15u;
            const uint32_t n255 = rem / 255u;
            for (uint32_t k = g; k < n255; k += G) out[o + k] = 0xFFu;
            o += n255;
            if (g == 0u) out[o] = (uint8_t)(rem - n255 * 255u);
            o += 1u;
        }
        lit_start = cur;
        base = cur;
        i0 = 0u;
    }
    o = emit_literals<G>(out, o, in, lit_start, n - lit_start, 0u, g);
    *produced = o;
    return 0;
}

// One group per chain; the chain's blocks are encoded in order with one persistent u32 table in LDS.
// tbl_state (nullable, 4096 u32 per chain): loaded before the first block unless it is cleared, saved at the end.
__global__ void __launch_bounds__(64) lz4_compress_chain_kernel(const uint8_t* in_base, const ChainBlock* blocks,
                                                               const uint32_t* chain_first, const uint32_t* chain_count,
                                                               uint32_t n_chains, uint8_t* out_base, const uint64_t* out_off,
                                                               const uint32_t* out_cap, uint32_t* out_len, int32_t* status,
                                                               uint32_t* tbl_state) {
    constexpr int G = 8;
    __shared__ __attribute__((aligned(16))) uint32_t tables[64 / G][4096];
    const uint32_t lane = threadIdx.x;
    Grp<G> grp;
    grp.g = lane % G;
    grp.shift = (lane / G) * G;
    const uint32_t c = blockIdx.x * (64 / G) + lane / G;
    if (c >= n_chains) return;
    uint32_t* tbl = &tables[lane / G][0];
    const uint32_t first = chain_first[c], count = chain_count[c];
    if (tbl_state) for (uint32_t k = grp.g; k < 4096u; k += G) tbl[k] = tbl_state[(size_t)c * 4096u + k];
    else for (uint32_t k = grp.g; k < 4096u; k += G) tbl[k] = 0u;
    for (uint32_t j = 0; j < count; ++j) {
        const ChainBlock b = blocks[first + j];
        const uint8_t* in = in_base + b.in_off;
        const uint8_t* dict = in_base + b.dict_off;
        const bool use_h4 = (b.flags & 1u) != 0u;
        if (b.flags & 2u) {
            for (uint32_t k = grp.g; k < 4096u; k += G) tbl[k] = 0u;
            // init_dict (compress.rs:571-583): positions 0,3,6,... in order, later entries win
            if (grp.g == 0u)
                for (uint32_t i = 0u; i + 8u <= b.dict_len; i += 3u)
                    tbl[use_h4 ? hidx4(cld32(dict + i)) : hidx5(cld64(dict + i))] = i;
        }
        if (b.repos != 0u)
            for (uint32_t k = grp.g; k < 4096u; k += G) { const uint32_t v = tbl[k]; tbl[k] = v > b.repos ? v - b.repos : 0u; }
        uint32_t produced = 0u;
        const int32_t st = encode_general<G>(in, b.in_len, b.in_pos, dict, b.dict_len, b.so, use_h4,
                                             out_base + out_off[first + j], out_cap[first + j], tbl, grp, &produced);
        if (grp.g == 0u) {
            status[first + j] = st;
            out_len[first + j] = st == 0 ? produced : 0u;
        }
    }
    if (tbl_state) for (uint32_t k = grp.g; k < 4096u; k += G) tbl_state[(size_t)c * 4096u + k] = tbl[k];
}

hipError_t launch_compress_chain(const uint8_t* in_base, const void* blocks, const uint32_t* chain_first,
                                 const uint32_t* chain_count, uint32_t n_chains, uint8_t* out_base,
                                 const uint64_t* out_off, const uint32_t* out_cap, uint32_t* out_len, int32_t* status,
                                 uint32_t* tbl_state, hipStream_t s) {
    if (n_chains == 0u) return hipSuccess;
    const uint32_t grid = (n_chains + 7u) / 8u;
    hipLaunchKernelGGL(lz4_compress_chain_kernel, dim3(grid), dim3(64), 0, s, in_base,
                       reinterpret_cast<const ChainBlock*>(blocks), chain_first, chain_count, n_chains, out_base, out_off,
                       out_cap, out_len, status, tbl_state);
    return hipGetLastError();
}

// MODE 0: the encoder wavefront alone (everything read from HBM/L2, output written inline).
// MODE 2: plus a second wavefront per workgroup that only walks ahead of the encoders and touches the input
//         lines they are about to need, so that the encoders' current-side loads hit L2 instead of paying the
//         first-touch HBM latency inside their serial chain (an in-order vmcnt makes self-prefetching useless:
//         a load behind a missing prefetch waits for it).
// MODE 3: (default) the second wavefront is the EMITTER: it pops the sequence records the encoders push into
//         their LDS queues, writes the compressed blocks and their status, and does MODE 2's prefetch.
// MODE 4: MODE 3 + the emitter also copies the input into an LDS ring per block that serves the encoders'
//         current-side reads (experiment; measured slower than MODE 3).
// BPW = blocks per workgroup = 64 / G.  With u16 tables a CU's 160 KiB of LDS hold two workgroups of eight
// blocks.  (Measured with smaller workgroups: four workgroups of five blocks = 20 tables do fit, five of four
// do not, and 16 384 blocks need 4 rounds either way; half-filled waves of four blocks are as fast as full ones.)
template <int G, typename TblT, int MODE, int BPW>
__global__ void __launch_bounds__(MODE == 0 ? 64 : 128) lz4_compress_blocks_kernel(CompressArgs a) {
    constexpr bool EQ = MODE == 3 || MODE == 4;
    constexpr bool RG = MODE == 4;
    constexpr uint32_t EQB = RG ? LZ4_EQ_BYTES_RING : LZ4_EQ_BYTES;
    extern __shared__ __attribute__((aligned(16))) uint8_t dyn_lds[];
    TblT* tables = reinterpret_cast<TblT*>(dyn_lds);                                        // [BPW][4096]
    uint8_t* rings = dyn_lds + (size_t)BPW * 4096u * sizeof(TblT);                          // per-block side structures (MODE 2..4)
    uint32_t* progress = reinterpret_cast<uint32_t*>(rings);                                // [BPW] (MODE 2)
    const uint32_t lane = threadIdx.x & 63u;
    Grp<G> grp;
    grp.g = lane % G;
    grp.shift = (lane / G) * G;
    const uint32_t j = lane / G;                                                            // block slot of this lane group
    const uint32_t b = blockIdx.x * BPW + j;
    if (EQ) {
        // sequence queues (+ input rings, MODE 4) behind the tables; the second wave is the emitter
        if (threadIdx.x < BPW) {
            const EmitQ q0{(lds_u8*)(rings + (size_t)threadIdx.x * EQB)};
            q0.set_head(0u); q0.set_tail(0u); q0.set_prog(0u);
            if (RG) in_ring_of(q0).set_window(0u, 0u);
        }
        __syncthreads();
        if (threadIdx.x >= 64u) {
            const bool live = j < BPW && b < a.n;
            emitter_wave<G, RG>(a, b, live, EmitQ{(lds_u8*)(rings + (size_t)(live ? j : 0u) * EQB)}, grp);
            return;
        }
    }
    if (MODE == 2) {
        if (threadIdx.x < BPW) progress[threadIdx.x] = 0u;
        __syncthreads();
        if (threadIdx.x >= 64u) {
            // ---- the prefetch wave: lane group j follows block j
            const bool live = j < BPW && b < a.n;
            const uint32_t n = live ? a.in_len[b] : 0u;
            const uint8_t* in = a.in_base + (live ? a.in_off[b] : 0ull);
            uint32_t pf = 0u, acc = 0u;
            for (;;) {
                const uint32_t pos = live ? *reinterpret_cast<volatile uint32_t*>(&progress[j]) : 0xFFFFFFFFu;
                const bool done = pos == 0xFFFFFFFFu;
                if (!__any(!done)) break;
                if (!done) {
                    if (pf < pos) pf = pos & ~127u;
                    if (pf < pos + LZ4_PF_AHEAD - 128u * G) {
                        const uint32_t at = pf + 128u * grp.g;
                        if (at + 4u <= n) acc += *reinterpret_cast<const volatile uint32_t*>(in + (at & ~3u));
                        pf += 128u * G;
                    }
                }
                __builtin_amdgcn_s_sleep(64);
            }
            if (acc == 0x9E3779B9u && live) progress[j] = acc;   // keeps the loads alive
            return;
        }
    }
    if (j >= BPW) return;
    if (b >= a.n) {
        if (MODE == 2 && grp.g == 0u) progress[j] = 0xFFFFFFFFu;
        return;
    }
    const uint32_t n = a.in_len[b];
    const uint32_t flags = a.flags ? a.flags[b] : 0u;
    uint32_t produced = 0u;
    int32_t st;
    {
        // the hash choice (compress.rs:559-566) is the same for every block of a typical batch: pick the
        // specialised loop when the whole wave agrees
        const bool h5 = (flags & 2u) != 0u || n >= 65535u;
        const uint8_t* in = a.in_base + a.in_off[b];
        uint8_t* out = a.out_base + a.out_off[b];
        TblT* tbl = tables + (size_t)j * 4096u;
        volatile uint32_t* pg = MODE == 2 ? &progress[j] : nullptr;
        const EmitQ eq{(lds_u8*)(rings + (size_t)(EQ ? j : 0u) * EQB)};
        if (__all(h5)) st = encode_block<G, TblT, 1, EQ, RG>(in, n, out, a.out_cap[b], flags, tbl, grp, &produced, pg, eq);
        else if (__all(!h5)) st = encode_block<G, TblT, 0, EQ, RG>(in, n, out, a.out_cap[b], flags, tbl, grp, &produced, pg, eq);
        else st = encode_block<G, TblT, 2, EQ, RG>(in, n, out, a.out_cap[b], flags, tbl, grp, &produced, pg, eq);
    }
    if (MODE == 2 && grp.g == 0u) *reinterpret_cast<volatile uint32_t*>(&progress[j]) = 0xFFFFFFFFu;
    if (grp.g == 0u && st != LZ4FLEX_DEV_QUEUED) {   // a queued block's length and status come from the emitter wave
        a.status[b] = st;
        a.out_len[b] = st == 0 ? produced : 0u;
    }
}

template <int G, typename TblT, int MODE, int BPW>
static hipError_t launch_c(const CompressArgs& a, hipStream_t s) {
    static_assert(BPW * G <= 64 && BPW <= 8, "one encoder wave per workgroup");
    const uint32_t grid = (a.n + BPW - 1u) / BPW;
    const size_t lds = (size_t)BPW * 4096u * sizeof(TblT) +
                       (MODE == 2 ? 64u : (MODE == 3 ? (size_t)BPW * LZ4_EQ_BYTES : (MODE == 4 ? (size_t)BPW * LZ4_EQ_BYTES_RING : 0u)));
    auto kern = lz4_compress_blocks_kernel<G, TblT, MODE, BPW>;
    if (lds > 65536u) {   // the attribute is per device: remember which devices have it (per instantiation)
        static unsigned long long have = 0ull;   // benign race: setting it twice is harmless
        int dev = 0;
        (void)hipGetDevice(&dev);
        const unsigned long long bit = 1ull << (dev & 63);
        if (!(have & bit)) {
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
            have |= bit;
        }
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(MODE == 0 ? 64 : 128), lds, s, a);
    return hipGetLastError();
}

template <int G, typename TblT, int BPW>
static hipError_t launch_m(const CompressArgs& a, int mode, hipStream_t s) {
    // MODE 2 (prefetch-only second wavefront) and MODE 4 (LDS input ring) were round-1 experiments that measured slower than
    // MODE 3; they are no longer instantiated
    if (mode == 3) return launch_c<G, TblT, 3, BPW>(a, s);
#ifdef LZ4FLEX_ALL_VARIANTS   // MODE 0 (the group encoder without an emitter wavefront) is a cross-check: variant builds only
    return launch_c<G, TblT, 0, BPW>(a, s);
#else
    return hipErrorInvalidValue;
#endif
}

// variant: bits 0..7 = lanes per block (8 or 16), bit 8 = blocks may exceed 64 KiB (u32 table),
// bits 9..10 + bit 12 = MODE of lz4_compress_blocks_kernel (0, 2, 3; bit 12: 4)
hipError_t launch_compress(const CompressArgs& a, int variant, hipStream_t s) {
    if (a.n == 0u) return hipSuccess;
    const int G = variant & 0xFF;
    const bool big = (variant & 0x100) != 0;
    const int mode = ((variant >> 9) & 3) | ((variant & 0x1000) ? 4 : 0);
    if (G == 8) return big ? launch_m<8, uint32_t, 8>(a, mode, s) : launch_m<8, uint16_t, 8>(a, mode, s);
    if (G == 16) return big ? launch_m<16, uint32_t, 4>(a, mode, s) : launch_m<16, uint16_t, 4>(a, mode, s);
    return hipErrorInvalidValue;
}

}  // namespace lz4flex_dev

#ifdef LZ4FLEX_PROFILE_PHASES
extern "C" int lz4flex_debug_phase(unsigned long long* cycles, unsigned long long* counts, int reset) {
    if (reset) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(lz4flex_dev::g_phase_cycles), z, sizeof z);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(lz4flex_dev::g_phase_counts), z, sizeof z);
        return 0;
    }
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(cycles, HIP_SYMBOL(lz4flex_dev::g_phase_cycles), 64);
    (void)hipMemcpyFromSymbol(counts, HIP_SYMBOL(lz4flex_dev::g_phase_counts), 64);
    return 0;
}
#endif
