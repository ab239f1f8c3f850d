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
//   * a persistent workgroup (12 wavefronts since round 6: two workgroups per CU are the 24 wavefronts that 80 VGPRs allow; rounds 2 - 5:
//     9) owns one 64 KiB window at a time, the window lives in LDS;
//   * the last wavefront ("indexer") walks the NEXT window 64 positions per step through the 4096-entry hash table
//     (LDS, u16) and stores, for every position, the distance to the most recent earlier position with the same
//     hash (cand[], 2 B per position, in a slot of the workgroup's workspace);
//   * wavefronts 0..10 ("workers") each own a segment of ~6 KiB of the CURRENT window.  Per step of 64 positions:
//     the lanes whose candidate distance differs from their predecessor's ("heads") count their true match
//     length against the LDS window (16 B per iteration, all heads of the step at once); a DPP prefix maximum
//     gives every position the match that reaches furthest; one-step lazy evaluation is a lane compare; the
//     greedy walk over the step is scalar (ballot, s_ff1, v_readlane) and touches no memory; the selected
//     sequences are encoded lane-parallel straight into the segment's body (a prefix sum of their sizes places them);
//   * segments are independent parses (a match never crosses a segment end) that may reference the whole window;
//     after a barrier every worker places its segment's bytes: the literals left over at a segment's end are
//     carried into the first sequence of the next segment (its token is written at that point).
// Memory traffic: the input twice (the indexer's stream, the workers' window) and the output once are 2.2 GiB per GiB of input; cand[] is
// 2 GiB written + 2 GiB read back and the segment bodies are written and read once more -- the 512 workgroups' workspace (~340 KB each)
// does not fit the L2s (32 MiB), so all of it crosses the fabric to the Infinity Cache: the counters on the L2's fabric side see ~10 GB
// per GiB (7.7 x the algorithmic bytes; profiles/traffic.json), of which HBM itself has to serve the input and the output.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lz4_device.h"

namespace lz4flex_dev {
namespace wave {

typedef __attribute__((address_space(3))) uint8_t lds_u8;
typedef __attribute__((address_space(3))) uint16_t lds_u16;
typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) uint8_t g_u8;          // global memory, named: pointers that cross a call would be flat
typedef __attribute__((address_space(1))) u32x4 g_u32x4;
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) u32x2 g_u32x2;
typedef __attribute__((address_space(1))) uint32_t g_u32;
typedef __attribute__((address_space(1))) int32_t g_i32;

constexpr uint32_t WINDOW = 65536u;
#ifndef LZ4W_WORKERS
#define LZ4W_WORKERS 11
#endif
constexpr uint32_t WORKERS = LZ4W_WORKERS;   // worker wavefronts = segments per window
constexpr uint32_t GROUPS = WINDOW / 512u;   // segment boundaries are multiples of 512 (one cand[] group)
// segment w of a full window = [seg_lo(w), seg_lo(w + 1)); a shorter window clips them.  The workers do not get equal shares: a
// later segment sees more of the window, finds more candidates and costs more per position, and the three workers that share their
// SIMD with the indexer get more done (JSON tiles, matching cycles per window with 11 / 12 groups each: 174 173 156 169 190 177 186
// 189 202 213 209 k -- the window waits for the slowest), so the segments are 12 13 13 13 12 11 12 12 10 10 10 groups long (the
// scalar model has the same table; eight workers, rounds 2 - 5: 16 17 17 17 15 16 15 15)
__host__ __device__ constexpr uint32_t seg_lo(uint32_t w) {
    if (WORKERS == 11u) {
        constexpr uint32_t lo[12] = {0u, 12u, 25u, 38u, 51u, 63u, 74u, 86u, 98u, 108u, 118u, 128u};
        return 512u * lo[w < 11u ? w : 11u];
    }
    if (WORKERS == 8u) {
        constexpr uint32_t lo[9] = {0u, 16u, 33u, 50u, 67u, 82u, 98u, 113u, 128u};
        return 512u * lo[w < 8u ? w : 8u];
    }
    return 512u * ((GROUPS * w) / WORKERS);
}
__host__ __device__ constexpr uint32_t seg_longest() {
    uint32_t m = 0u;
    for (uint32_t w = 0u; w < WORKERS; ++w) m = seg_lo(w + 1u) - seg_lo(w) > m ? seg_lo(w + 1u) - seg_lo(w) : m;
    return m;
}
constexpr uint32_t SEG = seg_longest();   // longest segment
constexpr uint32_t HIST = WINDOW / 2u;      // history in front of a block (see Item)
// Start of segment j (0 .. WORKERS) of a window whose first `skip` positions are history.  Without history in front of the block
// the segments are the fixed ones, clipped (only an anchored last window has skip != 0); with it (skip >= HIST in every window)
// the eight segments share the parsed part in the same proportions -- clipped, half of the workers would have nothing to do.
// Starts other than `skip` itself are multiples of 512.
// send: 0 = the fixed segments; else the eight segments share [skip, send) in the fixed ones' proportions (WINDOW with history or sliding
// windows; the end of a SUB-WINDOW, see Item::sub)
__device__ __forceinline__ uint32_t seg_start(uint32_t j, uint32_t skip, uint32_t send) {
    if (send == 0u) return seg_lo(j) > skip ? seg_lo(j) : skip;
    if (j == 0u) return skip;
    if (j >= WORKERS) return send;
    const uint32_t span = send > skip ? send - skip : 0u;
    const uint32_t v = (skip + span * (seg_lo(j) / 512u) / 128u) & ~511u;
    return v > skip ? v : skip;
}

constexpr uint32_t CAP = 1024u;           // longest match a head counts
constexpr uint32_t SKIPD = 64u;           // a position buried this deep in a running match is not evaluated
constexpr uint32_t CARRY_SLOTS = 16u;        // per block: a ring of {out_pos, pend, window} records, one cache line each
constexpr uint32_t CARRY_DWORDS = CARRY_SLOTS * 16u;
constexpr uint32_t CARRY_SPINS = 1u << 18;   // x (s_sleep(16) + a load from L2): a good fraction of a second for another workgroup's window
constexpr uint32_t LONGK = 84u;           // heads that match this far ...
constexpr uint32_t NEARP = 20u;           // ... and are followed within this many positions by another such head stop counting there ...
constexpr uint32_t LONGN = 8u;            // ... when the superstep holds at least this many of them
constexpr uint32_t HBITS = 12u;
constexpr uint32_t THREADS = 64u * (WORKERS + 1u);
#ifndef LZ4W_RING_SLOTS
#define LZ4W_RING_SLOTS 1
#endif
constexpr uint32_t RING_SLOTS = LZ4W_RING_SLOTS; // chunk slots of the indexer (one wavefront, LDS operations in order: one is enough; rounds 2 - 5 had two)
// per worker: two arrays of 64 words (a superstep's compaction buffers) behind 16 bytes whose last word is entry 0 of the 129-entry
// array a superstep of more than 64 heads gathers from
constexpr uint32_t TMP_OFF = 16u;
constexpr uint32_t CMP_OFF = TMP_OFF + 256u;
constexpr uint32_t WORKER_LDS = CMP_OFF + 256u;
constexpr uint32_t CHUNK = 1024u;         // the indexer streams the next window in 1 KiB chunks (16 steps)
constexpr uint32_t CHUNK_SLOT = CHUNK + 16u;      // + the first bytes of the next chunk (positions 1021..1023 hash across the end)
constexpr uint32_t IDX_DEPTH = 4u;        // chunks in flight (registers) ahead of the one being indexed
// LDS layout (80 960 B with eleven workers: two workgroups per CU)
constexpr uint32_t L_WIN = 0u;                              // the window + 64 B of slack for the 16-byte compares
constexpr uint32_t L_TAB = WINDOW + 64u;                    // the indexer's table, 4096 x u16
constexpr uint32_t L_STG = L_TAB + (2u << HBITS);
constexpr uint32_t L_RING = L_STG + WORKERS * WORKER_LDS;   // the indexer's chunk slot(s)
constexpr uint32_t L_META = L_RING + RING_SLOTS * CHUNK_SLOT;
// L_META, words: 5 per worker (SegMeta), BlkCarry[2] (4), the drawn items (giq, 4), place_segment's mailbox (3), run_flag[2] (see index_window)
constexpr uint32_t META_WORDS = 5u * WORKERS + 4u + 4u + 3u + 2u;
constexpr uint32_t LDS_BYTES = L_META + ((META_WORDS * 4u + 63u) & ~63u);
static_assert(LDS_BYTES <= 81920u, "two workgroups per CU");
static_assert(WORKERS >= 1u && WORKERS <= 15u && seg_lo(WORKERS) == WINDOW, "segments tile the window");
// workspace per workgroup
constexpr uint32_t SLOT_BYTES = 2u * WINDOW;                // cand[] of one window, transposed: see index_window
constexpr uint32_t BODY_STRIDE = (SEG + SEG / 128u + 255u) / 256u * 256u + 256u;   // a segment's encoded bytes never exceed SEG + SEG/255 + 16
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
__device__ __forceinline__ uint32_t umin3(uint32_t a, uint32_t b, uint32_t c) { const uint32_t t = a < b ? a : b; return t < c ? t : c; }
__device__ __forceinline__ uint32_t ctz64(uint64_t x) { return (uint32_t)__builtin_ctzll(x); }
__device__ __forceinline__ uint32_t len_ext_bytes(uint32_t v) { return v >= 15u ? (v - 15u) / 255u + 1u : 0u; }   // compress.rs:237-247

// v_ffbl_b32: index of the lowest set bit, 0xFFFFFFFF for 0.  As inline assembly: written as __ffs(x) - 1 hipcc guards the zero case
// with a compare and a select per dword (first_diff: 22 vector instructions instead of 13, a tenth of the workers' instructions)
#ifndef LZ4W_NO_HWFFBL
__device__ __forceinline__ uint32_t ffbl(uint32_t x) { uint32_t r; asm("v_ffbl_b32 %0, %1" : "=v"(r) : "v"(x)); return r; }
#else
__device__ __forceinline__ uint32_t ffbl(uint32_t x) { return (uint32_t)(__ffs((int)x) - 1); }
#endif
// arguments of a non-inlined function arrive in VGPRs as flat pointers: make them scalar, global-address-space pointers
template <typename G, typename T>
__device__ __forceinline__ G* uni_gptr(T* p) {
    const uint64_t v = (uint64_t)p;
    return (G*)(((uint64_t)uni((uint32_t)(v >> 32)) << 32) | uni((uint32_t)v));
}

// ---- indexer --------------------------------------------------------------------------------------------------
// d(p) = distance from p to the most recent earlier position of the window with the same 4-byte hash (0: none).
// All 64 lookups of a step precede its 64 inserts (LDS operations of a wavefront execute in order); when lanes of a
// step share a bucket the highest one stays (the GPU tests compare with the scalar model, which assumes it).
// Memory: the window is streamed in 1 KiB chunks, 16 B per lane, IDX_DEPTH chunks in flight in registers (a single
// wavefront has to cover the HBM latency by itself), each chunk passes through an LDS slot from which the steps read
// their unaligned 4-byte values.  cand[] is stored transposed so that one 8-byte load gives a worker lane its
// distances for the 4 steps of a 256-block: block g = positions [256 g, 256 g + 256), lane i, step u -> u16 at
// (64 g + i) * 8 + 2 u.
struct ChunkRegs { u32x4 main; uint32_t extra; };

// The chunk loads are inline assembly with hand-counted waits.  Written as plain loads hipcc sank them to their use
// (register pressure), volatile loads are each followed by s_waitcnt vmcnt(0): either way a single wavefront saw the
// full HBM latency 64 times per window.  gfx950 retires vector memory operations in issue order on one vmcnt, so "at
// most N younger operations outstanding" means the two loads of the chunk have landed.  Between chunk_issue and
// chunk_wait the destination registers must not be read, copied or spilled: tests/test_isa_checks.py verifies that on
// the generated ISA (the "; lz4w-wait" marker names the registers).
// lz4_flex_amd/build.py runs that check on the ISA it is about to ship and compiles this file with -DLZ4W_PLAIN_LOADS if it
// fails: ordinary loads the compiler waits for itself -- the slow form described above, never a wrong one.
__device__ __forceinline__ void chunk_issue(ChunkRegs& r, const g_u8* __restrict__ gwin, uint32_t c, uint32_t lane) {
    const g_u8* pm = gwin + c * CHUNK + 16u * lane;
    const g_u8* pe = gwin + c * CHUNK + CHUNK;                     // same address in every lane
#ifdef LZ4W_PLAIN_LOADS
    r.main = *reinterpret_cast<const g_u32x4*>(pm);
    r.extra = *reinterpret_cast<const g_u32*>(pe);
#else
    asm volatile("global_load_dwordx4 %0, %1, off ; lz4w-load" : "=v"(r.main) : "v"(pm) : "memory");
    asm volatile("global_load_dword %0, %1, off ; lz4w-load" : "=v"(r.extra) : "v"(pe) : "memory");
#endif
}
template <int N>
__device__ __forceinline__ void chunk_wait(ChunkRegs& r) {
#ifndef LZ4W_PLAIN_LOADS
    asm volatile("s_waitcnt vmcnt(%2) ; lz4w-wait %0 %1" : "+v"(r.main), "+v"(r.extra) : "n"(N) : "memory");
#endif
}
// 16 steps of one chunk that sits in the LDS slot.  lp = slot + (lane & ~3): every LDS address below is lp + constant.
// FULL: every position is active (p < act_n), no predication.
template <bool FULL>
__device__ __forceinline__ void index_chunk(const lds_u8* lp, lds_u16* tab, g_u8* __restrict__ cand_lane, uint32_t c, uint32_t act_n,
                                            uint32_t lane) {
    // 8 steps at a time, in three passes, so that the wavefront sees two LDS round trips per 8 steps and not per step: (1) the
    // 4-byte values of all 8 steps (nothing in the table can change them), (2) lookup + insert of step 0, 1, ... back to
    // back (LDS executes a wavefront's operations in order: the lookup of step k + 1 sees the inserts of step k without any
    // wait), (3) distances.  Written in one loop per step hipcc has to wait for each lookup before the next step's read.
#pragma unroll
    for (uint32_t g8 = 0; g8 < 2u; ++g8) {
        uint32_t h[8], e[8];
        // the 4 bytes at offset 64 st + lane from two ALIGNED dwords: a byte-unaligned LDS access is executed one lane per
        // cycle (64 cycles per instruction; tools/ubench_lds.hip), an aligned one in 2-4.  The eight reads are inline
        // assembly so that they are issued together into eight register pairs (hipcc reused one pair and waited eight times).
        uint64_t xr[8];
#pragma unroll
        for (uint32_t u = 0; u < 8u; ++u)
            asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(xr[u]) : "v"((uint32_t)(uintptr_t)lp), "n"((g8 * 8u + u) * 16u), "n"((g8 * 8u + u) * 16u + 1u) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xr[0]), "+v"(xr[1]), "+v"(xr[2]), "+v"(xr[3]), "+v"(xr[4]), "+v"(xr[5]), "+v"(xr[6]), "+v"(xr[7]) :: "memory");
#pragma unroll
        for (uint32_t u = 0; u < 8u; ++u) {
            const uint32_t x = __builtin_amdgcn_alignbyte((uint32_t)(xr[u] >> 32), (uint32_t)xr[u], lane & 3u);
            h[u] = (x * 2654435761u) >> (32u - HBITS);
        }
#pragma unroll
        for (uint32_t u = 0; u < 8u; ++u) {
            const uint32_t p = c * CHUNK + (g8 * 8u + u) * 64u + lane;
            e[u] = p;                                           // inactive position: distance 0
            if (FULL || p < act_n) {
                e[u] = tab[h[u]];
                tab[h[u]] = (uint16_t)p;
            }
        }
#pragma unroll
        for (uint32_t g4 = 0; g4 < 2u; ++g4) {
            uint32_t pk[2] = {0u, 0u};
#pragma unroll
            for (uint32_t v = 0; v < 4u; ++v) {
                const uint32_t u = g4 * 4u + v;
                const uint32_t p = c * CHUNK + (g8 * 8u + u) * 64u + lane;
                pk[v >> 1] |= ((p - e[u]) & 0xFFFFu) << (16u * (v & 1u));
            }
            const u32x2 val = {pk[0], pk[1]};
            *reinterpret_cast<g_u32x2*>(cand_lane + (size_t)(c * 4u + g8 * 2u + g4) * 512u) = val;      // block (4 c + ..), lane: (64 g + lane) * 8
        }
    }
}

// RUN WINDOWS (round 6).  A window of at least RUN_MIN bytes whose bytes are all equal (zero pages, padding) is not indexed and not
// matched: it becomes ONE sequence -- [offset 1, everything from the window's first parsed position (its second byte when nothing
// precedes that in the window) to its last match end] -- the reference's encoding of a run (src/block/compress.rs:156-216:
// count_same_bytes is unbounded; 30 000 zeros are one match) one window at a time: 4 MiB of zeros are 64 sequences, 0.40 % (rounds
// 4 - 5: ~16 sequences per window, 0.64 %), and a decoder sees one long periodic copy per 64 KiB instead of a chain of KiB-long ones.
// The indexer decides: the first KiB comes for free (its registers are on their way anyway), the rest of the window is only read
// when that is one byte repeated.  The scalar model has the same rule (tests/sim/wave_encoder_model.c, run_window).
constexpr uint32_t RUN_MIN = 8192u;
// are bytes [from, wl) of the window all equal to the bytes of `splat` (one byte, four times)?  plain loads, whole wavefront
__device__ __attribute__((noinline)) uint32_t run_scan(const uint8_t* __restrict__ gwin_, uint32_t from_, uint32_t wl_, uint32_t splat_, uint32_t lane) {
    const g_u8* __restrict__ gwin = uni_gptr<const g_u8>(gwin_);
    const uint32_t from = uni(from_), wl = uni(wl_), splat = uni(splat_);
    uint32_t acc = 0u;
    const uint32_t n16 = from + ((wl - from) & ~15u);
    for (uint32_t i0 = from; i0 < n16; i0 += 4096u) {                  // (uniform trip count, the lane's share inside)
        u32x4 v[4];
#pragma unroll
        for (uint32_t j = 0; j < 4u; ++j) {
            const uint32_t o = i0 + 1024u * j + 16u * lane;
            v[j] = u32x4{splat, splat, splat, splat};
            if (o < n16) __builtin_memcpy(&v[j], (const void*)(gwin + o), 16);      // (any alignment)
        }
#pragma unroll
        for (uint32_t j = 0; j < 4u; ++j) acc |= (v[j].x ^ splat) | (v[j].y ^ splat) | (v[j].z ^ splat) | (v[j].w ^ splat);
        if (__builtin_amdgcn_ballot_w64(acc != 0u) != 0ull) return 0u;
    }
    if (lane < wl - n16) acc |= (uint32_t)gwin[n16 + lane] ^ (splat & 255u);
    return __builtin_amdgcn_ballot_w64(acc != 0u) == 0ull ? 1u : 0u;
}

// returns 1 when the window is a run window (see above; only asked when run_ok_: the caller has checked the lengths), else 0
__device__ __attribute__((noinline)) uint32_t index_window(const uint8_t* __restrict__ gwin_, uint32_t wl_, uint32_t act_n_, uint32_t rd_n_,
                             uint8_t* __restrict__ slot_t_, lds_u8* lds, uint32_t lane, uint32_t run_ok_) {
    const g_u8* __restrict__ gwin = uni_gptr<const g_u8>(gwin_);
    g_u8* __restrict__ slot_t = uni_gptr<g_u8>(slot_t_);
    const uint32_t wl = uni(wl_), act_n = uni(act_n_), rd_n = uni(rd_n_), run_ok = uni(run_ok_);
    lds_u16* tab = (lds_u16*)(lds + L_TAB);
    {
        lds_u32* t4 = (lds_u32*)(lds + L_TAB);
        for (uint32_t i = lane; i < (2u << HBITS) / 4u; i += 64u) t4[i] = 0u;
    }
    const uint32_t nchunks = (wl + CHUNK - 1u) / CHUNK;
    // main part: chunks whose 1028 bytes are readable and whose 1024 positions are all active -- straight-line code, so
    // that the compiler's s_waitcnt counts stay exact and IDX_DEPTH chunks really are in flight
    uint32_t n_main = rd_n >= CHUNK + 4u ? (rd_n - 4u) / CHUNK : 0u;
    n_main = n_main < act_n / CHUNK ? n_main : act_n / CHUNK;
    n_main = n_main < nchunks ? n_main : nchunks;
    g_u8* cand_lane = slot_t + 8u * lane;
    const uint32_t lane4 = lane & ~3u;
    if (n_main != 0u) {
        const uint32_t last = n_main - 1u;
        ChunkRegs q[IDX_DEPTH];
#pragma unroll
        for (uint32_t j = 0; j < IDX_DEPTH; ++j) chunk_issue(q[j], gwin, j < last ? j : last, lane);   // past the end: the last chunk again
#ifndef LZ4W_NO_RUN_WINDOWS
        if (run_ok) {
            // the first KiB (six younger loads: q[0] has landed): one byte repeated?  Then -- rare -- everything that is in flight
            // lands before the compiler may touch it, and the rest of the window is looked at
            chunk_wait<6>(q[0]);
            const uint32_t x0 = q[0].main.x, splat = (x0 & 255u) * 0x01010101u;
            const uint32_t dif = (q[0].main.x ^ splat) | (q[0].main.y ^ splat) | (q[0].main.z ^ splat) | (q[0].main.w ^ splat);
            if (__builtin_amdgcn_ballot_w64((dif != 0u) | (splat != uni(splat))) == 0ull) {
                chunk_wait<0>(q[0]); chunk_wait<0>(q[1]); chunk_wait<0>(q[2]); chunk_wait<0>(q[3]);
                if (run_scan(gwin_, CHUNK, wl, uni(splat), lane)) return 1u;
            }
        }
#endif
        // one chunk: wait for its two loads (N = vector memory operations issued after them), LDS slot, refill the
        // registers, 16 steps.  Issue order per chunk: [wait], 2 loads, 4 cand[] stores.
#define LZ4W_ONE_CHUNK(N, CC, J)                                                                    \
        {                                                                                           \
            lds_u8* sl = lds + L_RING + ((J) & (RING_SLOTS - 1u)) * CHUNK_SLOT;                                    \
            chunk_wait<N>(q[J]);                                                                    \
            __builtin_memcpy((void*)(sl + 16u * lane), &q[J].main, 16);                             \
            __builtin_memcpy((void*)(sl + CHUNK), &q[J].extra, 4);   /* every lane writes the same dword */ \
            const uint32_t cn = (CC) + IDX_DEPTH;                                                   \
            chunk_issue(q[J], gwin, cn < last ? cn : last, lane);                                   \
            index_chunk<true>(sl + lane4, tab, cand_lane, (CC), act_n, lane);                       \
        }
        static_assert(IDX_DEPTH == 4u, "the wait counts below are for four chunks in flight");
        uint32_t c = 0u;
        if (n_main >= IDX_DEPTH) {                       // first round: fewer stores are in flight yet
            LZ4W_ONE_CHUNK(6, 0u, 0)
            LZ4W_ONE_CHUNK(10, 1u, 1)
            LZ4W_ONE_CHUNK(14, 2u, 2)
            LZ4W_ONE_CHUNK(18, 3u, 3)
            c = IDX_DEPTH;
        }
        for (; c + IDX_DEPTH <= n_main; c += IDX_DEPTH) {   // steady state: 3 x 2 loads + 4 x 4 stores are younger
            LZ4W_ONE_CHUNK(22, c, 0)
            LZ4W_ONE_CHUNK(22, c + 1u, 1)
            LZ4W_ONE_CHUNK(22, c + 2u, 2)
            LZ4W_ONE_CHUNK(22, c + 3u, 3)
        }
#undef LZ4W_ONE_CHUNK
        // the <= 3 chunks left are already requested (q[0..2]): drain everything, then index them
        chunk_wait<0>(q[0]); chunk_wait<0>(q[1]); chunk_wait<0>(q[2]); chunk_wait<0>(q[3]);
        for (uint32_t j = 0; c < n_main; ++c, ++j) {
            lds_u8* sl = lds + L_RING + (j & (RING_SLOTS - 1u)) * CHUNK_SLOT;
            const ChunkRegs r = j == 0u ? q[0] : (j == 1u ? q[1] : q[2]);
            __builtin_memcpy((void*)(sl + 16u * lane), &r.main, 16);
            __builtin_memcpy((void*)(sl + CHUNK), &r.extra, 4);
            index_chunk<true>(sl + lane4, tab, cand_lane, c, act_n, lane);
        }
    }
    // tail (the block's last chunk or two): readable bytes end inside the chunk and / or its last positions may not start a
    // match; filled synchronously, zero padded
    for (uint32_t c = n_main; c < nchunks; ++c) {
        lds_u8* sl = lds + L_RING;
        const uint32_t base = c * CHUNK;
        for (uint32_t i = lane; i < (CHUNK + 16u) / 16u; i += 64u) {
            const uint32_t o = base + 16u * i;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (o + 16u <= rd_n) __builtin_memcpy(&v, (const void*)(gwin + o), 16);
            __builtin_memcpy((void*)(sl + 16u * i), &v, 16);
        }
        const uint32_t part = rd_n > base ? ((rd_n - base) & ~15u) : 0u;             // bytes of this chunk covered by 16-byte loads
        if (part < CHUNK + 16u && lane < 16u) {
            const uint32_t o = base + part + lane;
            if (part + lane < CHUNK + 16u) sl[part + lane] = o < rd_n ? gwin[o] : (uint8_t)0;
        }
        index_chunk<false>(sl + lane4, tab, cand_lane, c, act_n, lane);
    }
    return 0u;
}

// ---- worker ---------------------------------------------------------------------------------------------------
// Encoded sequences go straight to the segment's body in the workspace (global memory, L2): every sequence of a call knows its
// place from a prefix sum of the sizes, so all of them are written at once with exact-size stores (any alignment: the address
// mode of HSA queues is "unaligned").  Rounds 2 - 6 staged them in LDS first -- as many sequences per round as the buffer took,
// a 16 B-per-lane flush behind every round: ~500 instructions per call of encode_seqs where this takes ~80.
__device__ __forceinline__ void st_u8(g_u8* base, uint32_t o, uint32_t v) { base[o] = (uint8_t)v; }
__device__ __forceinline__ void st_u16(g_u8* base, uint32_t o, uint32_t v) { const uint16_t t = (uint16_t)v; __builtin_memcpy((void*)(base + o), &t, 2); }
__device__ __forceinline__ void st_u32(g_u8* base, uint32_t o, uint32_t v) { __builtin_memcpy((void*)(base + o), &v, 4); }
__device__ __forceinline__ void st_u64(g_u8* base, uint32_t o, uint64_t v) { __builtin_memcpy((void*)(base + o), &v, 8); }

// n < 16 literal bytes, exactly: ONE 16-byte read (over-reading source bytes is harmless), then 8/4/2/1-byte stores
__device__ __forceinline__ void copy_lit_small(g_u8* base, uint32_t o, const lds_u8* src, uint32_t n) {
    u32x4 v;
    __builtin_memcpy(&v, (const void*)src, 16);
    const bool n8 = (n & 8u) != 0u, n4 = (n & 4u) != 0u, n2 = (n & 2u) != 0u;
    const uint32_t w4 = n8 ? v.z : v.x;                               // the dword at byte offset (n & 8)
    const uint32_t wq = n8 ? (n4 ? v.w : v.z) : (n4 ? v.y : v.x);     // the dword at byte offset (n & 12)
    if (n8) st_u64(base, o, (uint64_t)v.x | ((uint64_t)v.y << 32));
    if (n4) st_u32(base, o + (n & 8u), w4);
    if (n2) st_u16(base, o + (n & 12u), wq);
    if (n & 1u) st_u8(base, o + (n & 14u), wq >> (n2 ? 16 : 0));
}

// One sequence of any shape by the whole wavefront (uniform arguments): [token, literal length bytes, literals] unless it is the
// segment's first sequence (whose token is written when the segments are placed), then offset and match length bytes, at body + o.
__device__ __attribute__((noinline)) void emit_generic(uint8_t* body_, uint32_t o_, uint32_t lit_src_, uint32_t lit_, uint32_t off_, uint32_t ml_, uint32_t first_, uint32_t lane) {
    g_u8* const body = uni_gptr<g_u8>(body_);
    uint32_t o = uni(o_);
    const uint32_t lit_src = uni(lit_src_), lit = uni(lit_), off = uni(off_), mlc = uni(ml_) - 4u, first = uni(first_);
    const lds_u8* const win = reinterpret_cast<const lds_u8*>((uintptr_t)0) + L_WIN;
    if (!first) {
        const uint32_t ne = len_ext_bytes(lit), hdr = 1u + ne;
        for (uint32_t i0 = 0u; i0 < hdr; i0 += 64u) {                  // (uniform trip counts, the lane's share inside)
            const uint32_t j = i0 + lane;
            uint32_t byte;
            if (j == 0u) byte = ((lit < 15u ? lit : 15u) << 4) | (mlc < 15u ? mlc : 15u);
            else byte = (j < ne) ? 255u : (lit - 15u) % 255u;
            if (j < hdr) st_u8(body, o + j, byte);
        }
        o += hdr;
        const uint32_t n16 = lit & ~15u;
        for (uint32_t i0 = 0u; i0 < n16; i0 += 1024u) {
            const uint32_t i = i0 + 16u * lane;
            if (i < n16) {
                u32x4 v;
                __builtin_memcpy(&v, (const void*)(win + lit_src + i), 16);
                __builtin_memcpy((void*)(body + o + i), &v, 16);
            }
        }
        if (lane < lit - n16) st_u8(body, o + n16 + lane, win[lit_src + n16 + lane]);
        o += lit;
    }
    const uint32_t me = len_ext_bytes(mlc), tail = 2u + me;
    for (uint32_t i0 = 0u; i0 < tail; i0 += 64u) {
        const uint32_t j = i0 + lane;
        uint32_t byte;
        if (j == 0u) byte = off & 255u;
        else if (j == 1u) byte = off >> 8;
        else byte = (j - 1u < me) ? 255u : (mlc - 15u) % 255u;
        if (j < tail) st_u8(body, o + j, byte);
    }
}

// What a segment's encoder carries from one call to the next (all uniform).
struct EncState {
    uint32_t body_len;                // bytes written to the body
    uint32_t has, first_lit, first_ml;   // the segment's first sequence (its token is written when the segments are placed)
    uint32_t last_end;                // end of the last encoded match: the next sequence's literals start here
};

// Lane-parallel encoding of the chosen sequences (lane k < npend: psq = match end << 16 | distance, psp = match start; the
// literals of sequence 0 start at st.last_end) into the segment's body.  A sequence with >= 15 literals or a match of
// >= 274 bytes needs length bytes beyond the lane-parallel form: such "hard" sequences are written one at a time by the whole
// wavefront, each at the place the prefix sum gave it.  Runs once per ~4 supersteps, two call sites.
#ifndef LZ4W_EXP_CALL_ENC       // inlined (round 6): JSON 3.44 -> 3.37 ms, text 4.67 -> 4.49 -- a call cost 16 scratch operations for callee-saved registers and six readfirstlanes
__device__ __forceinline__
#else
__device__ __attribute__((noinline))
#endif
EncState encode_seqs(uint32_t psq, uint32_t psp, uint32_t npend_, uint8_t* body_, uint32_t lane, EncState st_) {
    uint32_t npend = uni(npend_);
    lds_u8* const lds = reinterpret_cast<lds_u8*>((uintptr_t)0);
    {
        // A sequence without literals whose match has its predecessor's distance CONTINUES that match (matches end at CAP bytes, at
        // the LONGK cut among crowded long heads, where the walk took a better end for a while): the two are one match.  Runs
        // become a sequence per call instead of one per KiB (zeros: 0.74 % -> 0.45 %; and a decoder sees one long periodic copy
        // instead of a chain of short ones, each waiting for the one before).  Rare: one ballot decides.
        const uint32_t se0 = psq >> 16, off0 = psq & 0xFFFFu;
        const uint32_t pe0 = dpp_wave_shr1(se0, 0xFFFFFFFFu), offp = dpp_wave_shr1(off0, 0xFFFFFFFFu);
        const uint64_t live = npend >= 64u ? ~0ull : (1ull << npend) - 1ull;
        const uint64_t mm = __builtin_amdgcn_ballot_w64((psp == pe0) & (off0 == offp)) & live & ~1ull;
        if (mm != 0ull) {
            const uint64_t hm = live & ~mm;                                    // the sequences that stay (lane 0 is one)
            const uint64_t above = hm & ~((2ull << lane) - 1ull);              // ... behind this lane
            const uint32_t nh = above != 0ull ? ctz64(above) : npend;          // the next of them: the run ends in the lane before it
            const uint32_t e2 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((nh - 1u) & 63u) << 2), (int)se0);
            const uint32_t nheads = (uint32_t)__builtin_popcountll(hm);
            const bool head = __builtin_amdgcn_inverse_ballot_w64(hm);
            const uint32_t rk = __builtin_amdgcn_mbcnt_hi((uint32_t)(hm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hm, 0u));   // heads below this lane
            const uint32_t dst = head ? rk : nheads + (lane - rk);             // (the others go behind: every lane sends somewhere else)
            psq = (uint32_t)__builtin_amdgcn_ds_permute((int)((dst & 63u) << 2), (int)((e2 << 16) | off0));
            psp = (uint32_t)__builtin_amdgcn_ds_permute((int)((dst & 63u) << 2), (int)psp);
            npend = nheads;
        }
    }
    g_u8* const body = uni_gptr<g_u8>(body_);
    uint32_t body_len = uni(st_.body_len), has = uni(st_.has), first_lit = uni(st_.first_lit), first_ml = uni(st_.first_ml);
    uint32_t last_end = uni(st_.last_end);
    if (npend != 0u) {
        const bool issel = lane < npend;
        const uint32_t se = psq >> 16, off = psq & 0xFFFFu, sp = psp;
        const uint32_t pe = dpp_wave_shr1(se, last_end);
        const uint32_t lit = sp - pe, len = se - sp, mlc = len - 4u;
        const uint64_t hardm = __builtin_amdgcn_ballot_w64(issel & ((lit >= 15u) | (mlc >= 270u)));
        const bool first = (has == 0u) & (lane == 0u);                  // the segment's first sequence: its token comes later
        const uint32_t ext = mlc >= 15u ? 1u : 0u;
        uint32_t size = issel ? ((first ? 2u : 3u + lit) + ext) : 0u;
        if (hardm != 0ull) {                                            // the length bytes of the hard ones
            const uint32_t more = (first ? 0u : len_ext_bytes(lit)) + len_ext_bytes(mlc) - ext;
            size += __builtin_amdgcn_inverse_ballot_w64(hardm) ? more : 0u;
        }
        const uint32_t incl = wave_incl_add(size);
        const uint32_t o0 = body_len + (incl - size);
        const uint32_t was_first = has == 0u ? 1u : 0u;
        if (has == 0u) { has = 1u; first_lit = rdlane(lit, 0u); first_ml = rdlane(len, 0u); }
        const uint64_t ordm = (npend >= 64u ? ~0ull : (1ull << npend) - 1ull) & ~hardm;
        if (__builtin_amdgcn_inverse_ballot_w64(ordm)) {
            uint32_t o = o0;
            if (!first) {
                st_u8(body, o, (lit << 4) | (mlc < 15u ? mlc : 15u));
                copy_lit_small(body, o + 1u, lds + L_WIN + pe, lit);
                o += 1u + lit;
            }
            st_u16(body, o, off);
            if (ext) st_u8(body, o + 2u, mlc - 15u);
        }
        for (uint64_t hm = hardm; hm != 0ull; hm &= hm - 1ull) {
            const uint32_t q = ctz64(hm);
            emit_generic(body_, rdlane(o0, q), rdlane(pe, q), rdlane(lit, q), rdlane(off, q), rdlane(len, q), q == 0u ? was_first : 0u, lane);
        }
        body_len += rdlane(incl, 63u);
        last_end = rdlane(se, npend - 1u);
    }
    EncState r;
    r.body_len = body_len; r.has = has; r.first_lit = first_lit; r.first_ml = first_ml; r.last_end = last_end;
    return r;
}

template <uint32_t V> struct UConst { static constexpr uint32_t value = V; };

// One segment [s0, s1) of the window in LDS (s0 < s1).  mfl_end: positions p < mfl_end may start a match (p <= n - 12);
// mend: matches end here at the latest (segment end, block end - 5, 65535).
//
// The segment is walked in SUPERSTEPS of up to 256 positions (4 steps of 64: lane i looks at positions b + 64 u + i).
// Only ~1 position in 5 is a head, and everything expensive happens per head, so the heads of a superstep are
// compacted into the 64 lanes (rank = popcount of the head ballots below the lane, a 256-byte LDS buffer), counted,
// prefix-maximised and gathered back to the positions; a scalar greedy walk over the eligibility masks marks the chosen
// positions, which join the pending sequences in lanes (encode_seqs takes them a wavefront at a time).
// A superstep holds at most 128 heads: 256 positions if that many fit, else 128 (the scalar model walks the same
// supersteps); more than 64 heads pass through the wavefront in two chunks of 64.
//
// What is counted here is instructions, not latencies: 18 wavefronts share a CU whose ONE scalar unit retires about an
// instruction per cycle, and a vector instruction occupies its SIMD for 4 cycles whatever the number of active lanes;
// measured, one more scalar instruction per superstep costs 13 cycles, one more vector instruction 6.  Hence the
// compile-time superstep size (no masks for "step u is part of this superstep"), the hand-written walk, v_mbcnt's
// accumulator operand, LDS addressed from 0 (the kernel checks), ...
__device__ __attribute__((noinline)) void match_segment(const uint8_t* __restrict__ cand_t_, uint8_t* body_, uint32_t w_, uint32_t lane,
                              uint32_t s0_, uint32_t s1_, uint32_t mfl_end_, uint32_t mend_, unsigned long long* prof_) {
    const uint32_t w = uni(w_), s0 = uni(s0_), s1 = uni(s1_), mfl_end = uni(mfl_end_), mend = uni(mend_);
    const g_u8* __restrict__ cand_t = uni_gptr<const g_u8>(cand_t_);
    lds_u8* const lds = reinterpret_cast<lds_u8*>((uintptr_t)0);
    lds_u8* const win = lds + L_WIN;
    lds_u8* const stg = lds + L_STG + w * WORKER_LDS;
    lds_u32* const cmp = (lds_u32*)(stg + CMP_OFF);               // compaction buffer: 64 x 4 B
    lds_u32* const tmp = (lds_u32*)(stg + TMP_OFF);               // second one
    static_assert(RING_SLOTS == 1u || RING_SLOTS == 2u, "ring slots");
    EncState st;
    st.body_len = 0u; st.has = 0u; st.first_lit = 0u; st.first_ml = 0u; st.last_end = s0;
    uint32_t cursor = s0, carry = 0u, dlast = 0u;
    // sequences chosen but not encoded yet: lane k < npend holds the k-th
    uint32_t psq = 0u, psp = 0u, npend = 0u;
#ifdef LZ4W_PROF_STEPS      // tools: cycles per part of a superstep -> prof[8..15] (heads, compaction + lengths, scan, walk, merge, supersteps, loop + cand[] wait, encode_seqs)
    uint64_t pt[7] = {0, 0, 0, 0, 0, 0, 0}, pn = 0, pt0 = __builtin_readcyclecounter();
#define LZ4W_TICK(i) { const uint64_t t_ = __builtin_readcyclecounter(); pt[i] += t_ - pt0; pt0 = t_; }
#elif defined(LZ4W_MARK)   // tools: phase boundaries visible in a -S listing
#define LZ4W_TICK(i) asm volatile("; LZ4W_PHASE_END " #i);
#else
#define LZ4W_TICK(i)
#endif
    auto mbcnt = [&](uint64_t m, uint32_t base) -> uint32_t {      // bits of m below the lane + base
        return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, base));
    };
    auto ld4 = [&](uint32_t pos) -> uint32_t {                     // 4 bytes at any position from two aligned dwords
        const lds_u32* ap = (const lds_u32*)(win + (pos & ~3u));
        return __builtin_amdgcn_alignbyte(ap[1], ap[0], pos & 3u);
    };
    auto ld16a = [&](uint32_t pos) -> u32x4 {                      // 16 bytes at any position from five aligned dwords
        const lds_u32* ap = (const lds_u32*)(win + (pos & ~3u));
        const uint32_t w0 = ap[0], w1 = ap[1], w2 = ap[2], w3 = ap[3], w4 = ap[4], sh = pos & 3u;
        u32x4 v;
        v.x = __builtin_amdgcn_alignbyte(w1, w0, sh); v.y = __builtin_amdgcn_alignbyte(w2, w1, sh);
        v.z = __builtin_amdgcn_alignbyte(w3, w2, sh); v.w = __builtin_amdgcn_alignbyte(w4, w3, sh);
        return v;
    };
    auto first_diff = [](const u32x4& va, const u32x4& vc) -> uint32_t {   // index of the first differing bit, 128 if none
        const uint32_t f0 = ffbl(va.x ^ vc.x), f1 = ffbl(va.y ^ vc.y), f2 = ffbl(va.z ^ vc.z), f3 = ffbl(va.w ^ vc.w);   // 0xFFFFFFFF: equal
        return umin3(umin3(f0, f1 | 32u, f2 | 64u), f3 | 96u, 128u);        // f < 32 or all ones: "or" is "add" or keeps "none"
    };
    auto first_diff32 = [](const u32x4& va0, const u32x4& vc0, const u32x4& va1, const u32x4& vc1) -> uint32_t {   // the same over 32 bytes: 256 if none
        const uint32_t f0 = ffbl(va0.x ^ vc0.x), f1 = ffbl(va0.y ^ vc0.y), f2 = ffbl(va0.z ^ vc0.z), f3 = ffbl(va0.w ^ vc0.w);
        const uint32_t f4 = ffbl(va1.x ^ vc1.x), f5 = ffbl(va1.y ^ vc1.y), f6 = ffbl(va1.z ^ vc1.z), f7 = ffbl(va1.w ^ vc1.w);
        return umin3(umin3(umin3(umin3(f0, f1 | 32u, f2 | 64u), f3 | 96u, f4 | 128u), f5 | 160u, f6 | 192u), f7 | 224u, 256u);
    };

    // One superstep of NS steps at b (a multiple of 64 NS); t_u = distances of step u.  false: more than 128 heads, nothing
    // was changed, the caller halves.
    auto superstep = [&](auto nsc, const uint32_t b, const uint32_t t0, const uint32_t t1, const uint32_t t2, const uint32_t t3) -> bool {
        constexpr uint32_t NS = decltype(nsc)::value;
#ifdef LZ4W_PROF_STEPS
        LZ4W_TICK(5)
#endif
#ifdef LZ4W_EXP_PAD_VALU     // tools: what does one more vector / scalar instruction per superstep cost?
        { uint32_t pad_ = lane; _Pragma("unroll") for (int i_ = 0; i_ < LZ4W_EXP_PAD_VALU; ++i_) asm volatile("v_add_u32 %0, 1, %0" : "+v"(pad_)); asm volatile("" :: "v"(pad_)); }
#endif
#ifdef LZ4W_EXP_PAD_SALU
        { uint32_t pad_ = s0; _Pragma("unroll") for (int i_ = 0; i_ < LZ4W_EXP_PAD_SALU; ++i_) asm volatile("s_add_u32 %0, %0, 1" : "+s"(pad_) :: "scc"); asm volatile("" :: "s"(pad_)); }
#endif
        const uint32_t cend = carry >> 16;
        const uint32_t e1 = b + 64u * NS < s1 ? b + 64u * NS : s1;
        const uint32_t tl = NS == 4u ? t3 : (NS == 2u ? t1 : t0);
        // (one comparison: min(cursor, cend - (SKIPD - 1)) >= e1)
        const uint32_t deep = (cend > SKIPD - 1u ? cend : SKIPD - 1u) - (SKIPD - 1u);
        if ((cursor < deep ? cursor : deep) >= e1) {              // everything here lies deep inside a match already taken
            dlast = rdlane(tl, 63u);
            return true;
        }
        // heads: the distance changes, and the candidate's first 4 bytes equal the position's (both read as aligned dword
        // pairs: an unaligned LDS access costs a cycle per active lane).  Positions at or behind the window's last match
        // start (and with them everything behind a clipped segment end) carry distance 0: the indexer writes it.
        const uint32_t p0 = b + lane, p1 = p0 + 64u, p2 = p0 + 128u, p3 = p0 + 192u;
        // (conditions are kept as 64-bit lane masks: a bool that is changed under a branch makes hipcc materialise it in a
        // VGPR and compare again)
        auto bal = [](bool c) -> uint64_t { return __builtin_amdgcn_ballot_w64(c); };
        uint64_t K0 = bal(t0 != dpp_wave_shr1(t0, dlast)), K1 = 0ull, K2 = 0ull, K3 = 0ull;
        if (NS > 1u) K1 = bal(t1 != dpp_wave_shr1(t1, rdlane(t0, 63u)));
        if (NS > 2u) {
            K2 = bal(t2 != dpp_wave_shr1(t2, rdlane(t1, 63u)));
            K3 = bal(t3 != dpp_wave_shr1(t3, rdlane(t2, 63u)));
        }
        // "no candidate" is distance 0, and only two kinds of position carry it: the window's position 0 (never a head: its
        // predecessor value is the initial dlast, 0 as well; with history in front of the segment the b < s0 branch below masks it)
        // and the positions from the block's last match start on (>= mfl_end) -- so the test is paid in a block's last supersteps only
#ifndef LZ4W_NO_EDGE_T0
        if (b + 64u * NS > mfl_end)
#endif
        {
            // (the compares stay inside the branch: hipcc hoisted them -- four vector instructions in every superstep -- until the
            // empty asm made their operands the branch's own)
            uint32_t z0 = t0, z1 = t1, z2 = t2, z3 = t3;
            asm volatile("" : "+v"(z0), "+v"(z1), "+v"(z2), "+v"(z3));
            K0 &= bal(z0 != 0u);
            if (NS > 1u) K1 &= bal(z1 != 0u);
            if (NS > 2u) { K2 &= bal(z2 != 0u); K3 &= bal(z3 != 0u); }
        }
        if (b < s0) {
            // the segment starts inside this superstep (only the anchored last window of a block, see win_base): positions before
            // s0 are history and no heads; s0 itself has no predecessor (a head whenever it has a candidate), as at an aligned start
            const uint64_t N0 = bal(t0 != 0u), N1 = bal(t1 != 0u), N2 = bal(t2 != 0u), N3 = bal(t3 != 0u);
            K0 = (K0 | (N0 & bal(p0 == s0))) & bal(p0 >= s0);
            if (NS > 1u) K1 = (K1 | (N1 & bal(p1 == s0))) & bal(p1 >= s0);
            if (NS > 2u) { K2 = (K2 | (N2 & bal(p2 == s0))) & bal(p2 >= s0); K3 = (K3 | (N3 & bal(p3 == s0))) & bal(p3 >= s0); }
        }
        if (cend >= b + SKIPD) {                                 // positions buried >= SKIPD deep in the running best match
            const uint32_t T = cend - SKIPD;                      // p <= T: buried
            K0 &= bal(p0 > T);
            if (NS > 1u) K1 &= bal(p1 > T);
            if (NS > 2u) { K2 &= bal(p2 > T); K3 &= bal(p3 > T); }
        }
        // all 4-byte reads are issued before the first compare (one LDS round trip); lanes without a candidate read
        // something harmless.  Own side: dwords at b + 64 u + (lane & ~3), one address for all the steps.
        const uint32_t lane3 = lane & 3u;                        // (recomputed per superstep: registers that live across the call of encode_seqs are scarce)
        const lds_u32* own = (const lds_u32*)(win + (b + (lane & ~3u)));
        uint32_t a0 = 0u, a1 = 0u, a2 = 0u, a3 = 0u, g0 = 1u, g1 = 1u, g2 = 1u, g3 = 1u;
        a0 = __builtin_amdgcn_alignbyte(own[1], own[0], lane3);
        if (NS > 1u) a1 = __builtin_amdgcn_alignbyte(own[17], own[16], lane3);
        if (NS > 2u) { a2 = __builtin_amdgcn_alignbyte(own[33], own[32], lane3); a3 = __builtin_amdgcn_alignbyte(own[49], own[48], lane3); }
        g0 = ld4(p0 - t0);
        if (NS > 1u) g1 = ld4(p1 - t1);
        if (NS > 2u) { g2 = ld4(p2 - t2); g3 = ld4(p3 - t3); }
        const uint64_t M0 = K0 & bal(a0 == g0), M1 = NS > 1u ? K1 & bal(a1 == g1) : 0ull,
                       M2 = NS > 2u ? K2 & bal(a2 == g2) : 0ull, M3 = NS > 2u ? K3 & bal(a3 == g3) : 0ull;
        const uint32_t c0 = (uint32_t)__builtin_popcountll(M0), c1 = (uint32_t)__builtin_popcountll(M1),
                       c2 = (uint32_t)__builtin_popcountll(M2), c3 = (uint32_t)__builtin_popcountll(M3);
        const uint32_t H = c0 + c1 + c2 + c3;
        if (NS > 2u && H > 128u) return false;                   // (128 positions hold at most 128 heads)
        dlast = rdlane(tl, 63u);
        LZ4W_TICK(0)
        if (H == 0u && cursor >= e1) return true;                // no head, nothing to select: the running best is unchanged
        // compaction: head of rank r -> lane r, as (position << 16 | distance); v_mbcnt adds the heads of the earlier steps.
        // A superstep holds up to 128 heads: more than 64 are counted in two chunks of 64 (ranks 0..63, then 64..) -- the
        // positions' part of a superstep (heads, scan, walk, merge) is paid once for them; round 2 first halved such a
        // superstep and paid everything twice (JSON tiles: 44 of a window's 256 blocks, text: 184).
        const uint32_t r0 = mbcnt(M0, 0u), r1 = mbcnt(M1, c0), r2 = mbcnt(M2, c0 + c1), r3 = mbcnt(M3, c0 + c1 + c2);   // heads before this position
        const bool hh0 = __builtin_amdgcn_inverse_ballot_w64(M0), hh1 = __builtin_amdgcn_inverse_ballot_w64(M1),
                   hh2 = __builtin_amdgcn_inverse_ballot_w64(M2), hh3 = __builtin_amdgcn_inverse_ballot_w64(M3);
        // the match of one chunk's heads that reaches furthest: bestv[r] = best among heads 0..r of the chunk and everything
        // before it (cin); returns the best behind the chunk's last head
        auto chunk_best = [&](const uint32_t hv, const uint64_t hmask, const uint32_t cin, uint32_t& bestv) -> uint32_t {
            const uint32_t p = hv >> 16, d = hv & 0xFFFFu;
            // true match lengths of the heads in the lanes of hmask (their first 4 bytes are known to match)
            uint32_t lim = __builtin_elementwise_sub_sat(mend, p);
            lim = lim < CAP ? lim : CAP;
            uint32_t k = 4u;
            // The lanes still counting are a 64-bit lane mask in scalar registers (`am`), the loop below is uniform control flow
            // around exec-masked bodies, and every condition becomes a mask by a ballot of ONE compare outside the masked body
            // (a bool that is the AND of two compares, or that is set under a branch, goes through v_cndmask + v_cmp to be
            // balloted; carried around the loop it also drags the round counter into a vector register).
            uint64_t am = hmask & __builtin_amdgcn_ballot_w64(lim > 4u);
#ifdef LZ4W_EXP_NOLEN
            am = 0ull;
#endif
            if (am != 0ull) {
                uint32_t bits = 0u;
                if (__builtin_amdgcn_inverse_ballot_w64(am)) {     // 16 bytes, branch-free: most candidates end here
                    u32x4 va, vc;
                    // (ten aligned dword reads + 8 v_alignbyte instead of two unaligned 16-byte reads: an unaligned LDS access
                    // takes the CU's LDS pipe for a cycle per active lane, and ~48 lanes are active here; 4.38 -> 4.25 ms.  The
                    // later rounds have few active lanes: there the unaligned reads are the cheaper ones, measured)
                    va = ld16a(p + 4u); vc = ld16a(p + 4u - d);
                    bits = first_diff(va, vc);
                    k = 4u + (bits >> 3);
                }
                asm volatile("" : "+v"(bits));          // (the compare stays OUTSIDE the masked body: moved into it, its result comes back through v_cndmask + v_cmp)
                am = __builtin_amdgcn_ballot_w64(bits == 128u) & __builtin_amdgcn_ballot_w64(k < lim);
                for (uint32_t rounds = 0u; am != 0ull; ++rounds) {
                    if (rounds == 2u && (uint32_t)__builtin_popcountll(am) >= LONGN) {       // (a handful of long matches is ordinary data: no check)
                        // Heads still matching after LONGK bytes.  In a run (zeros, a repeated record) EVERY position is one, and 64
                        // lanes reading unaligned to the cap keep the LDS pipe busy for 16 ms per GiB: a head followed within NEARP
                        // positions by another such head stays at LONGK bytes (it starts just before a match that is at least as
                        // long as it is known to be); the last one of such a group goes on.
                        static_assert(LONGK == 4u + 16u + 2u * 32u, "the check sits behind the third compare round");
                        const uint64_t above = (am >> lane) >> 1;                                      // active lanes behind this one, bit 0 = lane + 1
                        const uint32_t next = lane + 1u + ctz64(above | (1ull << 63));                 // the next active lane (anything if none)
                        const uint32_t pn = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((next & 63u) << 2), (int)p);
                        am &= __builtin_amdgcn_ballot_w64(above == 0ull) | __builtin_amdgcn_ballot_w64(pn - p > NEARP);
                        if (am == 0ull) break;
                    }
                    bits = 0u;
                    if (__builtin_amdgcn_inverse_ballot_w64(am)) {      // 32 bytes per further round
                        const lds_u8* ap = win + p + k;
                        u32x4 va0, vc0, va1, vc1;
                        __builtin_memcpy(&va0, (const void*)ap, 16);
                        __builtin_memcpy(&vc0, (const void*)(ap - d), 16);
                        __builtin_memcpy(&va1, (const void*)(ap + 16), 16);
                        __builtin_memcpy(&vc1, (const void*)(ap - d + 16), 16);
                        bits = first_diff32(va0, vc0, va1, vc1);
                        k += bits >> 3;
                    }
                    asm volatile("" : "+v"(bits));
                    am = __builtin_amdgcn_ballot_w64(bits == 256u) & __builtin_amdgcn_ballot_w64(k < lim);
                }
            }
            k = k < lim ? k : lim;
            const uint32_t own_e = __builtin_amdgcn_inverse_ballot_w64(hmask & __builtin_amdgcn_ballot_w64(k >= 4u)) ? hv + (k << 16) : 0u;      // (p + k) << 16 | d
            bestv = wave_incl_max(own_e);
            bestv = bestv > cin ? bestv : cin;
            return rdlane(bestv, 63u);
        };
        // back to positions: every position takes the best of the heads at or before it (a gather by rank), then the
        // eligibility mask of each step: a match of >= 4 that does not yield to the next position (one-step lazy
        // evaluation: the successor reaches further by more than a byte; the last position of a superstep has none)
        auto incl_rank = [](uint32_t rbefore, uint64_t m) -> uint32_t {      // rbefore + (lane's bit of m): one v_addc with m as the carry-in
            uint32_t ri; uint64_t co;
            asm("v_addc_co_u32_e64 %0, %1, 0, %2, %3" : "=v"(ri), "=s"(co) : "v"(rbefore), "s"(m));
            return ri;                                            // <= H; H == all heads passed: the new running best
        };
        const uint32_t ri0 = incl_rank(r0, M0), ri1 = incl_rank(r1, M1), ri2 = incl_rank(r2, M2), ri3 = incl_rank(r3, M3);
        uint32_t q0, q1 = 0u, q2 = 0u, q3 = 0u;
        if (NS == 1u || H <= 64u) {
            if (hh0) cmp[r0] = (p0 << 16) | t0;
            if (NS > 1u) if (hh1) cmp[r1] = (p1 << 16) | t1;
            if (NS > 2u) { if (hh2) cmp[r2] = (p2 << 16) | t2; if (hh3) cmp[r3] = (p3 << 16) | t3; }
            const uint64_t hmask = H >= 64u ? ~0ull : (1ull << H) - 1ull;      // lanes 0 .. H - 1
            uint32_t hv = 0u;
            if (__builtin_amdgcn_inverse_ballot_w64(hmask)) hv = cmp[lane];
            uint32_t bestv;
            const uint32_t cin = carry;
            carry = chunk_best(hv, hmask, cin, bestv);
            const uint32_t bestsh = dpp_wave_shr1(bestv, cin);      // the same before head r, i.e. with r heads passed
            LZ4W_TICK(1)
            auto best_at = [&](uint32_t ri) -> uint32_t { return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(ri << 2), (int)bestsh); };
            q0 = best_at(ri0);
            if (NS > 1u) q1 = best_at(ri1);
            if (NS > 2u) { q2 = best_at(ri2); q3 = best_at(ri3); }
            if (H == 64u) {                                      // rank 64 = all heads passed: the new running best
                q0 = ri0 >= 64u ? carry : q0;
                q1 = ri1 >= 64u ? carry : q1;
                q2 = ri2 >= 64u ? carry : q2;
                q3 = ri3 >= 64u ? carry : q3;
            }
        } else {
            // 65..128 heads: ranks 0..63 through the wavefront, then ranks 64.. ; the results go to a 129-entry array in LDS --
            // best[r] = the best with r heads passed: best[0] the running best before the superstep, best[1 + r] = bestv of
            // rank r -- which starts one word below tmp | cmp (contiguous; LDS operations of a wavefront execute in order: a
            // chunk's heads are read before its results overwrite them); the positions gather from it
            lds_u32* const best = tmp - 1;
            static_assert(TMP_OFF + 256u == CMP_OFF && TMP_OFF >= 4u, "tmp and cmp are contiguous, the word below them is free");
            best[0] = carry;
            if (hh0 && r0 < 64u) cmp[r0] = (p0 << 16) | t0;
            if (hh1 && r1 < 64u) cmp[r1] = (p1 << 16) | t1;
            if (NS > 2u) { if (hh2 && r2 < 64u) cmp[r2] = (p2 << 16) | t2; if (hh3 && r3 < 64u) cmp[r3] = (p3 << 16) | t3; }
            uint32_t hv = cmp[lane], bestv;
            const uint32_t cmid = chunk_best(hv, ~0ull, carry, bestv);
            best[1u + lane] = bestv;
            if (hh0 && r0 >= 64u) cmp[r0 - 64u] = (p0 << 16) | t0;
            if (hh1 && r1 >= 64u) cmp[r1 - 64u] = (p1 << 16) | t1;
            if (NS > 2u) { if (hh2 && r2 >= 64u) cmp[r2 - 64u] = (p2 << 16) | t2; if (hh3 && r3 >= 64u) cmp[r3 - 64u] = (p3 << 16) | t3; }
            const uint64_t hmask1 = H >= 128u ? ~0ull : (1ull << (H - 64u)) - 1ull;   // lanes 0 .. H - 65
            hv = 0u;
            if (__builtin_amdgcn_inverse_ballot_w64(hmask1)) hv = cmp[lane];
            carry = chunk_best(hv, hmask1, cmid, bestv);
            best[65u + lane] = bestv;
            LZ4W_TICK(1)
            q0 = best[ri0]; q1 = best[ri1];
            if (NS > 2u) { q2 = best[ri2]; q3 = best[ri3]; }
        }
        const uint32_t e0 = q0 >> 16, ee1 = q1 >> 16, ee2 = q2 >> 16, ee3 = q3 >> 16;
        auto elig = [&](uint32_t pp, uint32_t e, uint32_t enext) -> uint64_t {
            return bal(e >= pp + 4u) & bal(enext <= e + 1u);
        };
        uint64_t em0 = elig(p0, e0, dpp_wave_shl1(e0, NS > 1u ? rdlane(ee1, 0u) : 0u)), em1 = 0ull, em2 = 0ull, em3 = 0ull;
        if (NS > 1u) em1 = elig(p1, ee1, dpp_wave_shl1(ee1, NS > 2u ? rdlane(ee2, 0u) : 0u));
        if (NS > 2u) { em2 = elig(p2, ee2, dpp_wave_shl1(ee2, rdlane(ee3, 0u))); em3 = elig(p3, ee3, dpp_wave_shl1(ee3, 0u)); }
        const uint32_t e1c = e1 < mfl_end ? e1 : mfl_end;
        if (e1c < b + 64u * NS) {                                // the segment's or the block's last positions
            em0 &= bal(p0 < e1c); em1 &= bal(p1 < e1c); em2 &= bal(p2 < e1c); em3 &= bal(p3 < e1c);
        }
        LZ4W_TICK(2)
        // ---- greedy walk (scalar): the first eligible position at or behind the cursor, step by step.  The loop only marks
        // the chosen positions (one bit each) and hops to the end of the chosen match: 5 scalar instructions, 2 branches and a
        // v_readlane per sequence, written out because the scalar unit is the bottleneck.  cb = cursor - b;
        // er = end of the position's best match relative to its step's base (>= 64: the next sequence starts behind the step) ----
        uint64_t S0 = 0ull, S1 = 0ull, S2 = 0ull, S3 = 0ull;
        uint32_t cb = cursor - b;
        auto walk = [&](auto offc, uint64_t em, uint32_t e, uint64_t& S, uint32_t& cbr) {
            constexpr uint32_t OFF = decltype(offc)::value;
            uint32_t sb = b + OFF;                                   // (one scalar add, then ONE vector subtraction: hipcc made two of e - b - OFF)
            asm volatile("" : "+s"(sb));
            const uint32_t er = e - sb;
            uint64_t m; uint32_t t, c;
            if constexpr (OFF == 0u) {
                asm volatile(
                    "s_cmp_lt_u32 %[cb], 64\n\t"
                    "s_cbranch_scc0 2f\n\t"
                    "s_lshr_b64 %[m], %[em], %[cb]\n\t"
                    "s_cbranch_scc0 2f\n"
                    "0:\n\t"
                    "s_ff1_i32_b64 %[t], %[m]\n\t"
                    "s_add_u32 %[t], %[t], %[cb]\n\t"
                    "s_bitset1_b64 %[S], %[t]\n\t"
                    "v_readlane_b32 %[cb], %[er], %[t]\n\t"
                    "s_cmp_lt_u32 %[cb], 64\n\t"
                    "s_cbranch_scc0 2f\n\t"
                    "s_lshr_b64 %[m], %[em], %[cb]\n\t"
                    "s_cbranch_scc1 0b\n"
                    "2:"
                    : [m] "=&s"(m), [t] "=&s"(t), [S] "+s"(S), [cb] "+s"(cbr)
                    : [em] "s"(em), [er] "v"(er)
                    : "scc");
                return;
            }
            asm volatile(
                "s_max_u32 %[c], %[cb], %[off]\n\t"
                "s_sub_u32 %[c], %[c], %[off]\n\t"
                "s_cmp_lt_u32 %[c], 64\n\t"
                "s_cbranch_scc0 2f\n\t"
                "s_lshr_b64 %[m], %[em], %[c]\n\t"
                "s_cbranch_scc0 1f\n"
                "0:\n\t"
                "s_ff1_i32_b64 %[t], %[m]\n\t"
                "s_add_u32 %[t], %[t], %[c]\n\t"
                "s_bitset1_b64 %[S], %[t]\n\t"
                "v_readlane_b32 %[c], %[er], %[t]\n\t"
                "s_cmp_lt_u32 %[c], 64\n\t"
                "s_cbranch_scc0 1f\n\t"
                "s_lshr_b64 %[m], %[em], %[c]\n\t"
                "s_cbranch_scc1 0b\n"
                "1:\n\t"
                "s_add_u32 %[cb], %[c], %[off]\n"
                "2:"
                : [m] "=&s"(m), [t] "=&s"(t), [c] "=&s"(c), [S] "+s"(S), [cb] "+s"(cbr)
                : [em] "s"(em), [er] "v"(er), [off] "n"(OFF)
                : "scc");
        };
#ifndef LZ4W_EXP_NOWALK
        walk(UConst<0u>{}, em0, e0, S0, cb);
        if (NS > 1u) walk(UConst<64u>{}, em1, ee1, S1, cb);
        if (NS > 2u) { walk(UConst<128u>{}, em2, ee2, S2, cb); walk(UConst<192u>{}, em3, ee3, S3, cb); }
#else
        asm volatile("" :: "s"(em0), "s"(em1), "s"(em2), "s"(em3), "v"(q0), "v"(q1), "v"(q2), "v"(q3));
#endif
        cursor = b + cb;
        cursor = cursor > e1 ? cursor : e1;
        LZ4W_TICK(3)
#ifdef LZ4W_PROF_STEPS
        pn += 1;
#endif
        const uint32_t n0 = (uint32_t)__builtin_popcountll(S0), n1 = (uint32_t)__builtin_popcountll(S1),
                       n2 = (uint32_t)__builtin_popcountll(S2), n3 = (uint32_t)__builtin_popcountll(S3);
        const uint32_t nsel = n0 + n1 + n2 + n3;
        if (nsel == 0u) { LZ4W_TICK(4) return true; }
#ifdef LZ4W_EXP_NOENC
        asm volatile("" :: "s"(S0), "s"(S1), "s"(S2), "s"(S3));
        return true;
#endif
        // ---- the chosen sequences join the pending ones: lane k = k-th sequence not encoded yet (through LDS: best |
        // position of the chosen positions by rank).  Encoding costs the same instructions for 5 sequences as for 60, so it
        // waits for a full wavefront. ----
        auto scatter = [&](uint32_t base) {                      // chosen position of rank r -> slot base + r
            if (__builtin_amdgcn_inverse_ballot_w64(S0)) { const uint32_t r = mbcnt(S0, base); cmp[r] = q0; tmp[r] = p0; }
            if (NS > 1u) if (__builtin_amdgcn_inverse_ballot_w64(S1)) { const uint32_t r = mbcnt(S1, base + n0); cmp[r] = q1; tmp[r] = p1; }
            if (NS > 2u) {
                if (__builtin_amdgcn_inverse_ballot_w64(S2)) { const uint32_t r = mbcnt(S2, base + n0 + n1); cmp[r] = q2; tmp[r] = p2; }
                if (__builtin_amdgcn_inverse_ballot_w64(S3)) { const uint32_t r = mbcnt(S3, base + n0 + n1 + n2); cmp[r] = q3; tmp[r] = p3; }
            }
        };
        if (npend + nsel > 64u) {
            // no room: the new sequences go to registers first, the pending ones are
            // encoded, the new ones become the pending ones.  Only two values live across the call.
            scatter(0u);
            uint32_t nq = 0u, np = 0u;
            if (lane < nsel) { nq = cmp[lane]; np = tmp[lane]; }
            LZ4W_TICK(4)
            st = encode_seqs(psq, psp, npend, body_, lane, st);
            psq = nq; psp = np; npend = nsel;
#ifdef LZ4W_PROF_STEPS
            LZ4W_TICK(6)
#endif
            return true;
        }
        scatter(npend);
        if ((lane >= npend) & (lane < npend + nsel)) { psq = cmp[lane]; psp = tmp[lane]; }
        npend += nsel;
        LZ4W_TICK(4)
        return true;
    };

    // cand[]: 8 bytes per lane and 256-block (see index_window), fetched one block ahead (two blocks ahead: same time).  The load is unconditional (the
    // block behind the last one is still inside the workspace): a conditional load made hipcc wait for the data right
    // where it was requested.
    u32x2 dn = *reinterpret_cast<const g_u32x2*>(cand_t + ((s0 >> 8) * 512u + lane * 8u));
#ifndef LZ4W_NO_PRIO
    // Issue priority by what is LEFT of the segment, in quarters (s_setprio 3 2 1 0): the window waits for its slowest worker, and
    // with the vector ports saturated a worker that has fallen behind only catches up if it issues before the others of its SIMD
    // (wait behind matching: 23 k of a window's 238 k cycles before).  JSON 3.39 -> 3.15 ms, text 4.54 -> 4.20; other mappings
    // (3 3 2 1 / 3 2 1 1 / 3 3 0 0) and a priority for the indexer: slower (profiles/r06_encoder.txt)
    const uint32_t prio_q = (s1 - s0 + 3u) / 4u;
    uint32_t prio_drop = s0 + prio_q, prio_lv = 3u;
    __builtin_amdgcn_s_setprio(3);
#endif
    for (uint32_t B0 = s0 & ~255u; B0 < s1; B0 += 256u) {       // (s0 is a multiple of 512 except in an anchored last window)
        // the four steps of this 256-block: as one superstep if it holds at most 128 heads, else in halves
        const uint32_t dq0 = dn.x & 0xFFFFu, dq1 = dn.x >> 16, dq2 = dn.y & 0xFFFFu, dq3 = dn.y >> 16;
        dn = *reinterpret_cast<const g_u32x2*>(cand_t + (((B0 >> 8) + 1u) * 512u + lane * 8u));
#ifndef LZ4W_NO_PRIO
        if (B0 >= prio_drop) {                                   // (one compare per 256-block; a division here was 25 scalar instructions)
            prio_drop += prio_q;
            prio_lv = prio_lv != 0u ? prio_lv - 1u : 0u;
            if (prio_lv == 2u) __builtin_amdgcn_s_setprio(2);
            else if (prio_lv == 1u) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
        }
#endif
        if (superstep(UConst<4u>{}, B0, dq0, dq1, dq2, dq3)) continue;
#pragma unroll 1
        for (uint32_t hf = 0u; hf < 2u; ++hf) {
            const uint32_t bh = B0 + 128u * hf;
            if (bh >= s1) break;
            const uint32_t ta = hf ? dq2 : dq0, tb = hf ? dq3 : dq1;
            superstep(UConst<2u>{}, bh, ta, tb, 0u, 0u);          // (128 positions: never more than 128 heads)
        }
    }
    st = encode_seqs(psq, psp, npend, body_, lane, st);
#ifndef LZ4W_NO_PRIO
    __builtin_amdgcn_s_setprio(3);                 // the serial part of a window (barrier, placement, the next window's load) at the highest priority
#endif
#ifdef LZ4W_PROF_STEPS
    if (prof_ && lane == 0u) {
        for (int i = 0; i < 5; ++i) atomicAdd(prof_ + 8 + i, (unsigned long long)pt[i]);
        atomicAdd(prof_ + 13, (unsigned long long)pn);
        atomicAdd(prof_ + 14, (unsigned long long)pt[5]);
        atomicAdd(prof_ + 15, (unsigned long long)pt[6]);
    }
#endif
    if (lane == 0u) {
        lds_u32* mp = (lds_u32*)(lds + L_META) + 5u * w;
        mp[0] = st.has; mp[1] = st.first_lit; mp[2] = st.first_ml; mp[3] = s1 - st.last_end; mp[4] = st.body_len;
    }
}

// bytes [0, n) with a per-byte generator; 64 lanes
template <typename F>
__device__ __forceinline__ void put_bytes(g_u8* dst, uint32_t n, uint32_t lane, F f) {
    for (uint32_t i = lane; i < n; i += 64u) dst[i] = (uint8_t)f(i);
}
// n bytes global -> global, any alignment on either side: 16 bytes per lane and access, four in flight
__device__ __forceinline__ void copy_bytes(g_u8* dst, const g_u8* src, uint32_t n, uint32_t lane) {
    uint32_t i = 0u;
    for (; i + 4096u <= n; i += 4096u) {
        u32x4 v[4];
#pragma unroll
        for (uint32_t j = 0; j < 4u; ++j) __builtin_memcpy(&v[j], (const void*)(src + i + 1024u * j + 16u * lane), 16);
#pragma unroll
        for (uint32_t j = 0; j < 4u; ++j) __builtin_memcpy((void*)(dst + i + 1024u * j + 16u * lane), &v[j], 16);
    }
    {
        u32x4 v[4];
#pragma unroll
        for (uint32_t j = 0; j < 4u; ++j) {
            const uint32_t o = i + 1024u * j + 16u * lane;
            v[j] = u32x4{0u, 0u, 0u, 0u};
            if (o + 16u <= n) __builtin_memcpy(&v[j], (const void*)(src + o), 16);
        }
#pragma unroll
        for (uint32_t j = 0; j < 4u; ++j) {
            const uint32_t o = i + 1024u * j + 16u * lane;
            if (o + 16u <= n) __builtin_memcpy((void*)(dst + o), &v[j], 16);
        }
    }
    const uint32_t done = i + ((n - i) & ~15u);
    if (done + lane < n) dst[done + lane] = src[done + lane];
}
__device__ __forceinline__ void put_len_header(g_u8* dst, uint32_t lit, uint32_t ml_nibble, uint32_t lane) {
    const uint32_t ne = len_ext_bytes(lit);
    put_bytes(dst, 1u + ne, lane, [&](uint32_t j) -> uint32_t {
        if (j == 0u) return ((lit < 15u ? lit : 15u) << 4) | ml_nibble;
        return (j < ne) ? 255u : (lit - 15u) % 255u;
    });
}

// Place segment w of the current window (after the barrier: every worker's SegMeta is final).
__device__ __attribute__((noinline)) void place_segment(lds_u8* lds, const uint8_t* __restrict__ gin_, uint32_t blk_len_, uint32_t win_idx_, bool last_win_,
                              uint32_t wl_, uint32_t wbase_, uint32_t wskip_, uint32_t send_, const uint8_t* body_, uint8_t* gout_, uint32_t carry_slot_, uint32_t w_, uint32_t lane,
                              uint32_t* out_len_, int32_t* status_, uint32_t* gcarry_, uint32_t iter_, uint32_t spins_max_, uint32_t runwin_) {
    const g_u8* __restrict__ gin = uni_gptr<const g_u8>(gin_);
    const uint32_t runwin = uni(runwin_);                          // a run window (index_window): segment 0 is the whole parsed part, the others are empty
    const g_u8* body = uni_gptr<const g_u8>(body_);
    g_u8* gout = uni_gptr<g_u8>(gout_);
    g_u32* out_len = uni_gptr<g_u32>(out_len_);
    g_i32* status = uni_gptr<g_i32>(status_);
    const uint32_t blk_len = uni(blk_len_), win_idx = uni(win_idx_), wl = uni(wl_), carry_slot = uni(carry_slot_), w = uni(w_);
    const uint32_t wbase = uni(wbase_), wskip = uni(wskip_);       // the window's first byte in the block; its first wskip positions are history
    const uint32_t spins_max = uni(spins_max_);                    // CARRY_SPINS (tests: 1 -- a window gives up at once)
    const bool last_win = uni((uint32_t)last_win_) != 0u;
    const uint32_t send = uni(send_);
    const lds_u32* mp = (const lds_u32*)(lds + L_META);
    lds_u32* cp = (lds_u32*)(lds + L_META) + 5u * WORKERS;          // BlkCarry[2]
    // Where this window's output starts and how many literals the block has pending: from the previous window, which this
    // workgroup placed itself (LDS) or, when the windows of a block are dealt to different workgroups (gcarry: a ring of
    // {out_pos, pend, window} records per block in the workspace), another one did -- that wait is bounded, a window that gives
    // up poisons the rest of its block (pend bit 31): the block is left with status 66 and launch_compress_wave's second launch
    // encodes it again.
    g_u32* gcarry = uni_gptr<g_u32>(gcarry_);
    uint32_t out_pos = 0u, pend = 0u, poisoned = 0u;
    if (win_idx != 0u) {
        if (gcarry != nullptr) {
            // ONE wavefront of the workgroup polls global memory (with all eight of all 512 workgroups polling the same few
            // cache lines the carry took 60 microseconds per window to get through); the others wait for it in LDS
            const uint32_t iter = uni(iter_);
            typedef volatile __attribute__((address_space(3))) uint32_t lds_vu32;
            lds_vu32* box = (lds_vu32*)(lds + L_META) + 5u * WORKERS + 8u;     // {out_pos, pend | poisoned << 31, iteration}
#ifndef LZ4W_NO_PRIO
            __builtin_amdgcn_s_setprio(0);             // waiting is not work: the polls below must not issue before the CU's other workgroup's matching
#endif
            if (w == WORKERS - 1u) {
                g_u32* sl = gcarry + 16u * (win_idx & (CARRY_SLOTS - 1u));
                uint32_t spins = 0u;
                while (__hip_atomic_load(sl + 2, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != win_idx && ++spins < spins_max) __builtin_amdgcn_s_sleep(16);
                out_pos = __hip_atomic_load(sl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                pend = __hip_atomic_load(sl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (spins >= spins_max) pend = 0x80000000u;
                if (lane == 0u) { box[0] = out_pos; box[1] = pend; }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (lane == 0u) box[2] = iter;
            } else {
                uint32_t spins = 0u;
                while (box[2] != iter && ++spins < (CARRY_SPINS << 4)) __builtin_amdgcn_s_sleep(2);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                out_pos = box[0];
                pend = spins >= (CARRY_SPINS << 4) ? 0x80000000u : box[1];
            }
#ifndef LZ4W_NO_PRIO
            __builtin_amdgcn_s_setprio(3);
#endif
            poisoned = pend >> 31;
            pend &= 0x7FFFFFFFu;
            if (poisoned) { out_pos = 0u; pend = 0u; }
            out_pos = uni(out_pos); pend = uni(pend); poisoned = uni(poisoned);
        } else {
            out_pos = cp[2u * (carry_slot ^ 1u)]; pend = cp[2u * (carry_slot ^ 1u) + 1u];
        }
    }
    auto seg_at = [&](uint32_t j) -> uint32_t {                     // start of segment j, clipped to the parsed part of the window
        const uint32_t v = runwin ? (j == 0u ? wskip : wl) : seg_start(j, wskip, send);
        return v < wl ? v : wl;
    };
    auto seg_len = [&](uint32_t j) -> uint32_t { return seg_at(j + 1u) - seg_at(j); };
    for (uint32_t j = 0; j < w; ++j) {
        const uint32_t sl = seg_len(j);
        if (mp[5u * j] != 0u) {
            const uint32_t L = pend + mp[5u * j + 1u];
            out_pos += 1u + len_ext_bytes(L) + L + mp[5u * j + 4u];
            pend = mp[5u * j + 3u];
        } else {
            pend += sl;
        }
    }
    const uint32_t sl = seg_len(w);
    const uint32_t abs0 = wbase + seg_at(w);                       // block-relative start of this segment
    if (gcarry != nullptr && w == WORKERS - 1u && !last_win && lane == 0u) {
        // the next window's carry depends on sizes only: hand it on BEFORE the bytes are copied (the windows of a block form a
        // chain through global memory; with the copies inside it a block advanced one window per 4 microseconds)
        uint32_t eo = out_pos, ep = pend;
        if (mp[5u * w] != 0u) {
            const uint32_t L = pend + mp[5u * w + 1u];
            eo += 1u + len_ext_bytes(L) + L + mp[5u * w + 4u];
            ep = mp[5u * w + 3u];
        } else {
            ep += sl;
        }
        g_u32* so = gcarry + 16u * ((win_idx + 1u) & (CARRY_SLOTS - 1u));
        __hip_atomic_store(so, eo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(so + 1, ep | (poisoned << 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(so + 2, win_idx + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (mp[5u * w] != 0u) {
        const uint32_t fl = mp[5u * w + 1u], L = pend + fl, ml = mp[5u * w + 2u] - 4u, bl = mp[5u * w + 4u];
        put_len_header(gout + out_pos, L, ml < 15u ? ml : 15u, lane);
        out_pos += 1u + len_ext_bytes(L);
        copy_bytes(gout + out_pos, gin + (abs0 + fl - L), L, lane);
        out_pos += L;
        copy_bytes(gout + out_pos, body, bl, lane);
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
            copy_bytes(gout + out_pos, gin + (blk_len - pend), pend, lane);
            out_pos += pend;
            if (lane == 0u) { *out_len = poisoned ? 0u : out_pos; *status = poisoned ? 66 /* a window never got its carry: launch_compress_wave's second launch encodes the block again */ : 0; }
        } else if (lane == 0u) {
            cp[2u * carry_slot] = out_pos;
            cp[2u * carry_slot + 1u] = pend;
        }
    }
}

// the current window into LDS: the worker threads, 16 B per thread and load, four loads in flight per thread
__device__ __attribute__((noinline)) void load_window(const uint8_t* __restrict__ g_, uint32_t wl_, uint32_t rd_n_, lds_u8* lds, uint32_t tid) {
    constexpr uint32_t NT = 64u * WORKERS;
    const g_u8* __restrict__ g = uni_gptr<const g_u8>(g_);
    const uint32_t wl = uni(wl_), rd_n = uni(rd_n_);                     // rd_n: the block's bytes from the window's first one on (>= wl)
    const uint32_t mis = (uint32_t)((16u - ((uintptr_t)g & 15u)) & 15u);    // bytes up to the first 16 B boundary
    const uint32_t head = mis < wl ? mis : wl;
    if (tid < head) lds[L_WIN + tid] = g[tid];
    const uint32_t nvec = (wl - head) / 16u;                                // <= 4096
    const g_u8* gp = g + head + 16u * tid;
    lds_u8* lp = lds + L_WIN + head + 16u * tid;
#pragma unroll 1
    for (uint32_t i0 = 0; i0 < nvec; i0 += 4u * NT) {
        u32x4 v[4];
#pragma unroll
        for (uint32_t j = 0; j < 4u; ++j) {
            v[j] = u32x4{0u, 0u, 0u, 0u};
            if (i0 + tid + NT * j < nvec) v[j] = *reinterpret_cast<const g_u32x4*>(gp + 16u * NT * j);
        }
#pragma unroll
        for (uint32_t j = 0; j < 4u; ++j)
            if (i0 + tid + NT * j < nvec) __builtin_memcpy((void*)(lp + 16u * NT * j), &v[j], 16);
        gp += 64u * NT;
        lp += 64u * NT;
    }
    const uint32_t done = head + 16u * nvec;
    if (tid < wl - done) lds[L_WIN + done + tid] = g[done + tid];
    // 64 bytes of slack behind the window, read by the head test and the 16-byte compares of its last positions: the block's next
    // bytes where it goes on (a head is a position whose 4 bytes equal its candidate's -- the last three positions of a window
    // that is not the block's last look past it, and with zeros there they were heads for the model and not for the kernel:
    // one head more or less decides whether a superstep is halved; found by tools/gpu_fuzz.py), zeros behind the block's end
    if (tid < 64u) lds[L_WIN + wl + tid] = wl + tid < rd_n ? g[wl + tid] : (uint8_t)0;
}

// History (a Linked frame's blocks, src/frame/compress.rs:280-299,327-356: the reference keeps the previous 64 KiB of the stream as
// the next block's dictionary): a block whose flags promise >= HIST readable bytes of the stream in front of it
// (LZ4FLEX_BLOCK_HISTORY, include/lz4flex_amd.h) is encoded as the item [block - HIST, block + len): `len` and `in_off` below are
// the ITEM's, its first `hist` positions are history only (indexed and loaded, never parsed or emitted), and its windows advance
// by HIST instead of WINDOW, so every parsed position has between 32 and 64 KiB of the stream behind it in its window -- the
// anchored-last-window mechanism, applied to every window.  All blocks of a launch still encode side by side: the history
// is INPUT, nothing waits.  Price: every byte is indexed and loaded twice (the indexer, 160 k of a window's 300 k cycles,
// becomes the longer half).
// SUB-WINDOWS (round 5; CompressArgs::sub = 2, 3 or 4): a block of at most 64 KiB is ONE window, one workgroup, and a worker's walk over its
// 8 KiB segment is what the block waits for -- with fewer blocks than workgroups most of the chip idles (160 text blocks: 160 of 512
// workgroups for 0.23 ms; a scalar compress_into: one).  Such a block is cut into `sub` items of 64 KiB / sub parsed bytes each: item k is
// the window [0, (k + 1) quarter) of the block with its first k quarters as history -- the anchored-last-window mechanism again --, its
// eight segments share the parsed quarter, and the items of a block are drawn by different workgroups like the windows of a long
// block (the carry ring).  Price: item k indexes and loads k + 1 quarters (2.5 x the indexing for sub = 4: free on an idle chip), the
// segments are shorter (ratio + 0.1 ... 0.3 %), and the bytes depend on `sub` (the scalar model takes it as a parameter).
struct Item {
    uint32_t blk, win, nwin, len, skip, hist, slide;     // slide: 0, or the bytes the windows advance by (HIST with history in front of the block; CompressArgs::slide for a long block)
    uint32_t sub;                                        // 0, or the parsed bytes per sub-window (WINDOW / CompressArgs::sub, rounded up to 512) of a block cut into sub-windows
    uint64_t in_off;
};
// window geometry: window t.win covers [win_base, win_base + win_len) of the item and parses [win_from, that end)
__device__ __forceinline__ uint32_t win_stride(const Item& t) { return t.slide != 0u ? t.slide : WINDOW; }
// the LAST window of an item longer than a window is anchored at the item's end and overlaps the window before it, so the tail can
// match backwards like the reference's (src/block/compress.rs:403-405: the window is the previous 64 KiB; a 66 675-byte block is
// 65 536 + 1 139 bytes)
__device__ __forceinline__ uint32_t win_base(const Item& t) {
    if (t.sub != 0u) return 0u;
    return (t.win + 1u == t.nwin && t.len > WINDOW) ? t.len - WINDOW : t.win * win_stride(t);
}
__device__ __forceinline__ uint32_t win_from(const Item& t) {
    if (t.sub != 0u) return t.win * t.sub;
    return t.win == 0u ? t.hist : (t.win - 1u) * win_stride(t) + WINDOW;
}
__device__ __forceinline__ uint32_t win_skip(const Item& t) { return win_from(t) - win_base(t); }
__device__ __forceinline__ uint32_t win_len(const Item& t) {
    if (t.sub != 0u) return t.win + 1u == t.nwin ? t.len : (t.win + 1u) * t.sub;     // (t.len <= WINDOW)
    const uint32_t base = win_base(t);
    return t.len > base ? (t.len - base < WINDOW ? t.len - base : WINDOW) : 0u;
}
// what the window's eight segments share (seg_start's `send`)
__device__ __forceinline__ uint32_t win_send(const Item& t) { return t.sub != 0u ? win_len(t) : (t.slide != 0u ? WINDOW : 0u); }
__device__ __forceinline__ void item_load(const CompressArgs& a, Item& it) {
    // first window of block it.blk (or invalid)
    it.win = 0u; it.nwin = 0u; it.len = 0u; it.skip = 0u; it.hist = 0u; it.slide = 0u; it.sub = 0u; it.in_off = 0ull;
    if (it.blk >= a.n) return;
    const uint32_t len = a.in_len[it.blk];
    const uint32_t cap = a.out_cap[it.blk];
    const uint32_t h = (a.flags != nullptr && (a.flags[it.blk] >> 8) >= HIST && len != 0u && len <= 0xFFFFFFFFu - HIST) ? HIST : 0u;
    it.hist = h;
    it.len = len + h;
    it.in_off = a.in_off[it.blk] - h;
    // (sliding windows without history: every window start of a block longer than a window sees >= 32 KiB behind it, like the
    // reference's continuously sliding window, src/block/compress.rs:403-405 -- 4 MiB log blocks 0.3027 -> 0.2928, the reference 0.2947)
    // Round 5: the stride of a long block is a parameter -- by 48 KiB (16 KiB of history at every window start) 0.2940, still below the
    // reference, with a third more indexing instead of twice as much.
    it.slide = h != 0u ? HIST : ((a.slide != 0u && len > WINDOW) ? a.slide : 0u);
    it.nwin = it.slide != 0u ? (it.len <= WINDOW ? 1u : 1u + (it.len - WINDOW + it.slide - 1u) / it.slide)
                      : (len == 0u ? 1u : (uint32_t)(((uint64_t)len + WINDOW - 1u) / WINDOW));
    if (a.sub >= 2u && a.sub <= 4u && h == 0u && len <= WINDOW) {                                  // sub-windows: see Item
        const uint32_t q = ((WINDOW + a.sub - 1u) / a.sub + 511u) & ~511u;                             // 32 768, 22 016 (three: 43 groups of 512), 16 384
        if (len > q) { it.sub = q; it.nwin = (len + q - 1u) / q; }
    }
    const uint64_t need = 20ull + (uint64_t)len * 110ull / 100ull;   // get_maximum_output_size, compress.rs:588-590
    if ((uint64_t)cap < need) { it.skip = 1u; it.nwin = 1u; }
}
// Two ways of dealing work to the persistent workgroups.  Block mode: workgroup g owns blocks g, g + G, ... and walks
// their windows in order (the carry between windows stays in LDS).  Window mode (fewer blocks than workgroups -- few, large
// blocks; one 1 GiB block would otherwise be one workgroup's 16 384 windows): workgroup g joins the team of block g mod n
// and draws that block's windows from the block's atomic counter (a team of one draws them in order; a drawn window is
// always held by a running workgroup, so the wait for the previous window's carry cannot deadlock whatever is resident);
// when its block has no windows left it moves on to the next block that has.
// redo != 0 (the second launch behind a window-mode launch): only blocks whose status is `redo` -- a window of theirs gave up
// waiting for its predecessor's carry (a time-sliced or preempted GPU) -- are encoded, again, in block mode, where nothing waits
__device__ __forceinline__ void item_seek(const CompressArgs& a, Item& it, int32_t redo) {
    if (redo != 0)
        while (it.blk < a.n && a.status[it.blk] != redo) it.blk += gridDim.x;
    item_load(a, it);
}
__device__ __forceinline__ void item_next_block(const CompressArgs& a, Item& it, int32_t redo) {
    if (it.win + 1u < it.nwin) { it.win += 1u; return; }
    it.blk += gridDim.x;
    item_seek(a, it, redo);
}
// window mode, one thread: the next window for this workgroup, starting the search at block b0 -> {block, window} (block == n: none)
__device__ __forceinline__ void item_draw(const CompressArgs& a, uint32_t* wctr, uint32_t b0, uint32_t& ob, uint32_t& ow) {
    const uint32_t n = a.n;
    uint32_t b = b0;
    // (its own block, then a few behind it: every block has workgroups that start at it and stay until it is done, so giving up
    // early loses no work -- and with many small blocks a workgroup without work would otherwise try them all, 2 us each)
    const uint32_t tries = n < 8u ? n : 8u;
    for (uint32_t t = 0; t < tries; ++t) {
        Item q;
        q.blk = b;
        item_load(a, q);
        const uint32_t w = atomicAdd(wctr + b, 1u);
        if (w < q.nwin) { ob = b; ow = w; return; }
        b = b + 1u == n ? 0u : b + 1u;
    }
    ob = n; ow = 0u;
}

// A run window's one match, window-relative: [ms, me), asked for (ok) only when the window is long enough and the block format allows
// that match (it starts <= len - 12 and ends <= len - 5, src/block/mod.rs:37-61; ends are 16-bit numbers here as everywhere)
struct RunGeom { uint32_t ok, ms, me; };
__device__ __forceinline__ RunGeom run_geom(const Item& t) {
    RunGeom g;
    const uint32_t base = win_base(t), wl = win_len(t), skip = win_skip(t);
    g.ms = skip > 1u ? skip : 1u;
    uint32_t me = t.len >= 5u + base ? t.len - 5u - base : 0u;
    me = me < wl ? me : wl;
    g.me = me < 65535u ? me : 65535u;
    const uint32_t act_abs = t.len >= 12u ? t.len - 11u : 0u;         // positions p < act_abs may start a match
#ifdef LZ4W_NO_RUN_WINDOWS
    g.ok = 0u;
#else
    g.ok = (wl >= RUN_MIN && !t.skip && base + g.ms < act_abs && g.me >= g.ms + 4u) ? 1u : 0u;
#endif
    return g;
}

// prof (nullable, tools only): cycle sums per role, [0] indexer busy, [1] indexer at barriers, [2] workers matching,
// [3] workers at the barrier behind matching, [4] placing, [5] loading the next window, [6] at the barrier behind loading,
// [7] windows
__device__ __forceinline__ void wave_body(const CompressArgs& a, uint8_t* __restrict__ ws, uint32_t* __restrict__ carry,
                                          unsigned long long* __restrict__ prof, const int32_t redo, const uint32_t carry_spins) {
    extern __shared__ __attribute__((aligned(16))) uint8_t dyn_lds[];
    lds_u8* lds = (lds_u8*)dyn_lds;
    if ((uint32_t)(uintptr_t)lds != 0u) {
        // match_segment addresses LDS from 0 (this kernel has no static LDS, so the dynamic segment starts there -- with today's
        // toolchain).  If that ever stops being true every block reports a device failure; nothing is encoded from wrong addresses.
        for (uint32_t b = blockIdx.x * THREADS + threadIdx.x; b < a.n; b += gridDim.x * THREADS) { a.out_len[b] = 0u; a.status[b] = 66; }
        return;
    }
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t w = uni(threadIdx.x >> 6);
#ifdef LZ4W_EXP_IDX_PRIO     // tools: the indexer's issue priority (the workers: by progress, see match_segment); measured 1 and 2: slower
    if (w == WORKERS) __builtin_amdgcn_s_setprio(LZ4W_EXP_IDX_PRIO);
#endif
    uint8_t* my_ws = ws + (size_t)blockIdx.x * WS_BYTES;
    uint8_t* slots = my_ws;                                        // two cand[] slots
    uint8_t* bodies = my_ws + 2u * SLOT_BYTES;

    const bool wmode = carry != nullptr;                          // windows (not blocks) are dealt to the workgroups
    lds_u32* giq = (lds_u32*)(lds + L_META) + 5u * WORKERS + 4u;   // window mode: item indices drawn by thread 0
    uint32_t* wctr = wmode ? carry + CARRY_DWORDS * (size_t)a.n : nullptr;   // per-block window counters behind the carry slots
    Item it, ix;                                                  // the window being matched; the next one (the indexer runs one window ahead)
    if (threadIdx.x == 0u) giq[6] = 0u;                           // place_segment's mailbox: no iteration yet
    if (wmode) {
        if (threadIdx.x == 0u) {
            uint32_t b = a.n, wn = 0u;
            item_draw(a, wctr, blockIdx.x % a.n, b, wn);
            giq[0] = b; giq[1] = wn;
        }
        __syncthreads();
        it.blk = giq[0];
        item_load(a, it);
        it.win = giq[1];
        ix.blk = a.n;                                             // drawn behind the prologue (see there)
        item_load(a, ix);
    } else {
        it.blk = blockIdx.x;
        item_seek(a, it, redo);
        ix = it;
        item_next_block(a, ix, redo);
    }
    if (it.blk >= a.n) return;
    uint32_t k = 0u;

    lds_u32* run_flag = (lds_u32*)(lds + L_META) + 5u * WORKERS + 11u;   // [slot]: the window whose cand[] slot this would be is a run window
    auto do_index = [&](const Item& t, uint32_t slot) {
        if (t.skip) { if (lane == 0u) run_flag[slot] = 0u; return; }
        const uint32_t base = win_base(t), wl = win_len(t);
        const uint32_t act_abs = t.len >= 12u ? t.len - 11u : 0u;            // positions p < act_abs start 4 readable bytes and may match
        const uint32_t act_n = act_abs > base ? (act_abs - base < wl ? act_abs - base : wl) : 0u;
        // a run window's one match (see index_window): from the first parsed position that has a byte in front of it in the window to
        // the window's last match end; asked for only where that is a match the block format allows
        const RunGeom rg = run_geom(t);
        const uint32_t run = index_window(a.in_base + t.in_off + base, wl, act_n, t.len - base, slots + (size_t)slot * SLOT_BYTES, lds, lane, rg.ok);
        if (lane == 0u) run_flag[slot] = run;
    };
    auto do_load = [&](const Item& t) {
        if (t.skip) return;
        load_window(a.in_base + t.in_off + (size_t)win_base(t), win_len(t), t.len - win_base(t), lds, threadIdx.x);
    };

    // (sums are kept in registers and added to prof[] once, when the workgroup is done: an atomic per tick made the
    // waits look twice as long as they are)
    uint64_t t_prev = prof ? __builtin_readcyclecounter() : 0ull;
    uint64_t t_acc[7] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull};
    auto tick = [&](uint32_t slot) {
        if (prof) {
            const uint64_t t = __builtin_readcyclecounter();
#pragma unroll
            for (uint32_t i = 0; i < 7u; ++i) t_acc[i] += i == slot ? t - t_prev : 0ull;
            t_prev = t;
        }
    };
    // prologue: cand[] of the first window, the first window into LDS
    if (w == WORKERS) { do_index(it, 0u); tick(0u); }
    else { do_load(it); tick(5u); }
    __syncthreads();
    tick(w == WORKERS ? 1u : 6u);
    if (wmode) {
        // The second window is drawn only now, a prologue later: drawn in a row with the first, a workgroup would hold two
        // CONSECUTIVE windows of a block, and every window of the block would wait for the one before it to be matched.
        if (threadIdx.x == 0u) {
            uint32_t b2 = a.n, w2 = 0u;
            item_draw(a, wctr, it.blk, b2, w2);
            giq[2] = b2; giq[3] = w2;
        }
        __syncthreads();
        ix.blk = giq[2];
        item_load(a, ix);
        ix.win = giq[3];
        __syncthreads();                                          // (giq[2..3] are written again in the loop)
    }
    for (;;) {
        const uint32_t wl = win_len(it);
        const bool last_win = it.win + 1u == it.nwin;
        if (w == WORKERS) {
            if (ix.blk < a.n) do_index(ix, (k + 1u) & 1u);
        } else if (!it.skip) {
            const uint32_t base = win_base(it), skip = win_skip(it);
            uint32_t s0 = seg_start(w, skip, win_send(it));
            uint32_t s1 = seg_start(w + 1u, skip, win_send(it));
            s0 = s0 < wl ? s0 : wl;
            s1 = s1 < wl ? s1 : wl;
            const uint32_t act_abs = it.len >= 12u ? it.len - 11u : 0u;
            const uint32_t mfl_end = act_abs > base ? act_abs - base : 0u;      // window-relative, may exceed wl
            uint32_t mend = it.len >= 5u ? it.len - 5u : 0u;                    // block-relative
            mend = mend > base ? mend - base : 0u;
            mend = mend < s1 ? mend : s1;
            mend = mend < 65535u ? mend : 65535u;
            if (run_flag[k & 1u] != 0u) {
                // a run window: worker 0 writes its one sequence (the first of its "segment": offset and length bytes go to the body,
                // the token is written when it is placed), the others have nothing
                const RunGeom rg = run_geom(it);
                lds_u32* mp = (lds_u32*)(lds + L_META) + 5u * w;
                if (w == 0u) {
                    emit_generic(bodies, 0u, 0u, 0u, 1u, rg.me - rg.ms, 1u, lane);
                    if (lane == 0u) { mp[0] = 1u; mp[1] = rg.ms - skip; mp[2] = rg.me - rg.ms; mp[3] = wl - rg.me; mp[4] = 2u + len_ext_bytes(rg.me - rg.ms - 4u); }
                } else if (lane == 0u) {
                    mp[0] = 0u; mp[1] = 0u; mp[2] = 0u; mp[3] = 0u; mp[4] = 0u;
                }
            } else if (s0 < s1) {
                match_segment(slots + (size_t)(k & 1u) * SLOT_BYTES, bodies + (size_t)w * BODY_STRIDE, w, lane, s0, s1, mfl_end, mend, prof);
            } else if (lane == 0u) {                              // an empty segment (history only, or behind the block's end)
                lds_u32* mp = (lds_u32*)(lds + L_META) + 5u * w;
                mp[0] = 0u; mp[1] = 0u; mp[2] = 0u; mp[3] = 0u; mp[4] = 0u;
            }
        }
        tick(w == WORKERS ? 0u : 2u);
        __syncthreads();
        tick(w == WORKERS ? 1u : 3u);
        if (wmode && threadIdx.x == 0u) {                          // the window behind ix
            uint32_t b2 = a.n, w2 = 0u;
            if (ix.blk < a.n) item_draw(a, wctr, ix.blk, b2, w2);
            giq[2] = b2; giq[3] = w2;
        }
        if (w != WORKERS) {
            if (it.skip) {
                if (threadIdx.x == 0u) { a.out_len[it.blk] = 0u; a.status[it.blk] = LZ4FLEX_DEV_E_OUTPUT_TOO_SMALL; }
            } else {
                place_segment(lds, a.in_base + it.in_off, it.len, it.win, last_win, wl, win_base(it), win_skip(it), win_send(it), bodies + (size_t)w * BODY_STRIDE,
                              a.out_base + a.out_off[it.blk], k & 1u, w, lane, a.out_len + it.blk, a.status + it.blk,
                              wmode ? carry + CARRY_DWORDS * (size_t)it.blk : nullptr, k + 1u, carry_spins, run_flag[k & 1u]);
            }
        }
        tick(w == WORKERS ? 1u : 4u);
        if (prof && threadIdx.x == 0u) atomicAdd(prof + 7, 1ull);
        it = ix;
        k += 1u;
        if (it.blk >= a.n) {
            if (prof && lane == 0u)
                for (uint32_t i = 0; i < 7u; ++i)
                    if (t_acc[i] != 0ull) atomicAdd(prof + i, (unsigned long long)t_acc[i]);
#ifdef LZ4W_PROF_WORKERS    // tools: matching cycles per worker -> prof[16 + w] (the segment table above comes from these)
            if (prof && lane == 0u && w < WORKERS) atomicAdd(prof + 16u + w, (unsigned long long)t_acc[2]);
#endif
            break;
        }
        if (w != WORKERS) do_load(it);
        tick(w == WORKERS ? 1u : 5u);
        __syncthreads();
        tick(w == WORKERS ? 1u : 6u);
        if (wmode) { ix.blk = giq[2]; item_load(a, ix); ix.win = giq[3]; }
        else item_next_block(a, ix, redo);
    }
}

__global__ void __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(6, 6))) lz4_compress_wave_kernel(const CompressArgs a, uint8_t* __restrict__ ws,
                                                                    uint32_t* __restrict__ carry, unsigned long long* __restrict__ prof, uint32_t carry_spins) {
    wave_body(a, ws, carry, prof, 0, carry_spins);
}
// the launch behind a window-mode launch (a kernel of its own so that profiles tell the two apart): block mode, only blocks
// left with status `redo`
__global__ void __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(6, 6))) lz4_compress_wave_redo_kernel(const CompressArgs a, uint8_t* __restrict__ ws,
                                                                                                                     int32_t redo) {
    wave_body(a, ws, nullptr, nullptr, redo, CARRY_SPINS);
}

}  // namespace wave

// cand[] slots and segment bodies per workgroup, then the window-carry ring and window counter per block for batches of fewer
// blocks than workgroups
size_t compress_wave_workspace_bytes(int n_workgroups) { return (size_t)n_workgroups * (wave::WS_BYTES + wave::CARRY_DWORDS * 4u + 4u); }

hipError_t launch_compress_wave(const CompressArgs& a, void* workspace, int n_workgroups, hipStream_t s, unsigned long long* prof, bool carry_wait) {
    if (a.n == 0u) return hipSuccess;
    if (!workspace || n_workgroups <= 0) return hipErrorInvalidValue;
    static unsigned long long have = 0ull;
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(have & bit)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wave::lz4_compress_wave_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)wave::LDS_BYTES);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(wave::lz4_compress_wave_redo_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)wave::LDS_BYTES);
        if (e != hipSuccess) return e;
        have |= bit;
    }
    // fewer blocks than workgroups: windows, not blocks, are dealt out (see item_seek); the carry slots and the item counter
    // behind the per-workgroup workspaces start at zero
    uint32_t* carry = nullptr;
    if (a.n < (uint32_t)n_workgroups) {
        carry = (uint32_t*)((uint8_t*)workspace + (size_t)n_workgroups * wave::WS_BYTES);
        const hipError_t e = hipMemsetAsync(carry, 0, (size_t)a.n * (wave::CARRY_DWORDS * 4u + 4u), s);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(wave::lz4_compress_wave_kernel, dim3((uint32_t)n_workgroups), dim3(wave::THREADS), wave::LDS_BYTES, s, a,
                       (uint8_t*)workspace, carry, prof, carry_wait ? wave::CARRY_SPINS : 1u);
    const hipError_t first = hipGetLastError();      // (reading it clears it: a failed first launch must not be reported as success)
    if (first != hipSuccess) return first;
#ifndef LZ4W_EXP_NO_REDO     // (tools: shows what the second launch is for)
    if (carry != nullptr) {
        // A window that gave up waiting for its predecessor's carry (every wait is bounded: a GPU shared with another process, a
        // debugger) left its block with status 66.  Those blocks are encoded again by their own workgroup, window after window,
        // with the carry in LDS: the same bytes, no waiting, and no valid input turns into an error.  Nothing to do: ~10 us.
        const uint32_t g = a.n < (uint32_t)n_workgroups ? a.n : (uint32_t)n_workgroups;
        hipLaunchKernelGGL(wave::lz4_compress_wave_redo_kernel, dim3(g), dim3(wave::THREADS), wave::LDS_BYTES, s, a, (uint8_t*)workspace, 66);
    }
#endif
    return hipGetLastError();
}

}  // namespace lz4flex_dev
