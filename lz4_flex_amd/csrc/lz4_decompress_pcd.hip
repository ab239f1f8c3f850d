// lz4_decompress_pcd.hip -- PARALLEL-CHAIN LZ4 block decoder for gfx950: FEW, LARGE blocks (and small batches).
//
// What it replaces: lz4_flex::block::decompress_into / decompress_internal (src/block/decompress.rs:201-449) for the batch
// shapes that leave most of the chip idle when a block is one serial token chain: BASELINE configs[3] (256 x 4 MiB blocks per
// GPU), configs[2] (160 blocks), a scalar decompress_into of one large block.  Every other decoder here walks a block's token
// chain with ONE lane (or one wavefront's scalar hop): 80-140 MB/s per block whatever its size.  Here ONE WORKGROUP of 1 024
// lanes decodes a block, and both halves of the reference's loop are parallel inside the block:
//
//   PARSE (the token chain, decompress.rs:244-332).  The compressed stream is consumed in tiles of 32 KiB staged in LDS; a tile
//   is cut into 256 parts of 128 bytes and lane k walks part k from an ASSUMED entry (the part's first byte), marking the
//   token positions it visits in an LDS bitmap and noting where its chain leaves the part.  A chain started at a wrong byte
//   falls into step with the true chain after a few sequences, and two chains that share a position are identical from
//   there on.  One wavefront then follows the exits from part 0 (whose entry is true) by pointer jumping: the exit of a live
//   part is the true entry of the part it lands in.  Parts whose entry changed are walked again, only as far as their new chain
//   differs from the old one; this repeats until nothing changes (3 rounds on the benchmark data; NP + 1 at worst).  The set
//   bits of the live parts are the tile's sequences, in order.  No lane needs the output position, and a hop of a walk is ONE
//   LDS round trip (see the walk loop).
//
//   COPY (decompress.rs:334-437).  Sequences are executed 2 048 at a time, two per lane: token re-parsed from the LDS tile, a
//   block-wide prefix sum of literal + match lengths places all of them at once in an LDS WINDOW of the output (26 KiB of
//   history + up to 48 KiB new bytes); all literals are copied at once; a match whose whole source lies inside an earlier match
//   of the batch is RELINKED to that match's source; a match then waits until the sequences that produce its source bytes
//   (found through a table of the first sequence per 64 output bytes) have set their DONE bits, copies 16 bytes at a time and
//   sets its own.  Wavefronts poll independently: a dependency costs an LDS round trip, not a barrier.
//   Long / overlapping / window-straddling matches and long literal runs are copied by a whole wavefront (non-overlapping
//   steps of doubling size for periodic matches); a sequence longer than the window is executed alone by the whole workgroup
//   on the output itself.  The window is written back 16 bytes per lane after every batch.
//
//   ROLES.  A batch of few large blocks gets two workgroups per block: one parses, the other copies (see "roles" below).
//
// It diagnoses nothing: any irregularity (every DecompressError of src/block/mod.rs:82-98, a sink too small, an offset behind
// the output, a tile that does not settle) marks the block, and lz4_decompress_blocks_kernel decodes it again in the
// reference's check order and names the error (as behind lz4_decompress_wave.hip).  The host model of this algorithm is
// tests/sim/pcd_model.cpp (same walker: lz4_pcd_common.h; tests/test_pcd_model.py).  Every wait in here is bounded.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lz4_device.h"
#include "lz4_pcd_common.h"

namespace lz4flex_dev {
namespace pcd {

typedef __attribute__((address_space(3))) uint8_t lds_u8;
typedef __attribute__((address_space(3))) uint16_t lds_u16;
typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr uint32_t align_up(uint32_t v, uint32_t a) { return (v + a - 1u) / a * a; }
constexpr uint32_t ceil_log2(uint32_t v) { uint32_t r = 0u; while ((1u << r) < v) ++r; return r; }

// CT compressed bytes per tile (+ CM staged behind it), parts of P bytes, T threads, S sequences per thread and batch, window = HIST + WNEW
template <uint32_t CT_, uint32_t CM_, uint32_t P_, uint32_t T_, uint32_t S_, uint32_t HIST_, uint32_t WNEW_>
struct Geo {
    static constexpr uint32_t CT = CT_, CM = CM_, P = P_, NP = CT_ / P_, T = T_, S = S_, HIST = HIST_, WNEW = WNEW_, WIN = HIST_ + WNEW_;
    // a sequence longer than this is executed ALONE by the whole workgroup on the output itself (the "giant" path: stores of a
    // pattern for short periods) even when the window would hold it -- round 6: the encoder's run windows are 48 KiB matches
    // with offset 1; inside a batch such a match is one wavefront's copy and one lane's 768 table entries, 58 us each
    static constexpr uint32_t GIANT = WNEW_ < 16384u ? WNEW_ : 16384u;
    static constexpr uint32_t BN = T * S;                         // sequences per batch
    static constexpr uint32_t MW = CT / 32u;                      // mark words
    static constexpr uint32_t PW = P / 32u;                       // mark words per part
    static constexpr uint32_t NW = T / 64u;                       // wavefronts
    static constexpr uint32_t MAXSEQ = CT / 3u + 2u;              // a sequence with a match is >= 3 bytes
    static constexpr uint32_t PPL = (NP + 63u) / 64u;             // parts per lane of the resolving wavefront
    static constexpr uint32_t MAX_ITERS = NP + 2u;
    static constexpr uint32_t KP = (HIST + 16u * T - 1u) / (16u * T);   // 16-byte pieces per thread when the history slides
    // LDS layout
    static constexpr uint32_t L_CT = 16u;                         // (a walk's hop reads from the byte BEFORE its token: that byte exists for the tile's first one too)
    static constexpr uint32_t L_MARK = align_up(L_CT + CT + CM + 16u, 16u);
    static constexpr uint32_t L_ENT = L_MARK + 4u * MW;
    static constexpr uint32_t L_EXT = L_ENT + 4u * NP;
    static constexpr uint32_t L_NXT = L_EXT + 4u * NP;
    static constexpr uint32_t L_RCH = L_NXT + 4u * (NP + 4u);
    static constexpr uint32_t L_TOK = align_up(L_RCH + 4u * (NP + 4u), 16u);
    static constexpr uint32_t L_BST = align_up(L_TOK + 2u * MAXSEQ, 16u);
    static constexpr uint32_t L_MST = align_up(L_BST + 4u * (BN + 1u), 16u);
    static constexpr uint32_t L_OFF = align_up(L_MST + 4u * BN, 16u);       // the batch's match offsets (u16)
    static constexpr uint32_t L_TB = align_up(L_OFF + 2u * BN, 16u);       // per 64 bytes of the batch's output: the first sequence that starts at or behind them (u16)
    static constexpr uint32_t L_DONE = align_up(L_TB + 2u * (WNEW / 64u + 2u), 16u);
    static constexpr uint32_t L_WSUM = L_DONE + 4u * align_up(BN / 32u, 4u);
    static constexpr uint32_t L_CTL = L_WSUM + 4u * align_up(NW * S, 4u);
    static constexpr uint32_t L_WIN = align_up(L_CTL + 4u * 32u, 16u);
    static constexpr uint32_t LDS_BYTES = L_WIN + WIN + 64u;
    static_assert(4u * MW <= 2u * MAXSEQ, "the walks' scratch bitmap fits the token list's space");
    static_assert(CT % P == 0 && P % 32 == 0 && T % 64 == 0 && MW <= T && NP <= T && NP <= 64u * PPL && HIST % 16 == 0, "geometry");
    static_assert(LDS_BYTES <= 160u * 1024u, "LDS");
};
using GeoProd = Geo<CT, CM, P, THREADS, SEQ_PER_LANE, HIST, WNEW>;   // lz4_pcd_common.h: 32 KiB tiles, 128-byte parts, 1 024 lanes x 2 sequences, 26 + 48 KiB window
using GeoTest = Geo<2048u, 256u, 64u, 128u, 2u, 512u, 1024u>;        // tests: boundaries of every kind inside small inputs
// Medium batches (round 4).  The kernel is a chain of latencies with the CU's issue slots mostly empty, and the production geometry
// fits ONE workgroup per CU: from 257 blocks on the batch runs in rounds.  Fewer lanes and less LDS per block make a block slower
// (JSON block alone: 1 024 lanes 0.14 ms, 512 lanes 0.20, 256 lanes 0.30, 128 lanes 0.50) and the CU hold more of them: 512 lanes and
// 53 KB -- two per CU -- win for 257 ... 512 blocks (JSON 512 blocks: 0.23 instead of 0.28 ms), 256 lanes and 29 KB -- four per CU --
// for 513 ... 1 024 (1 024 blocks: 0.38 instead of 0.52); above that a pair of wavefronts per block is ahead
// (profiles/r04_decoder_shapes.txt; capi.cpp launch_decompress_fast holds the thresholds).
using GeoMid512 = Geo<8192u, 1024u, 64u, 512u, 2u, 8192u, 16384u>;
using GeoMid256 = Geo<4096u, 512u, 64u, 256u, 2u, 6144u, 8192u>;

// control words in LDS.  A word is written on one side of a barrier and read on the other: C_BAD (a sequence that does not parse) is
// written before the batch's first barrier and read behind it, C_BAD2 (an offset behind the output) before the second one --
// with one word for both, a fast thread's second write could reach a slow thread's first read and split the workgroup
#ifndef LZ4P_SLEEP
#define LZ4P_SLEEP 1     // x 64 cycles: a wavefront whose open matches all wait for their producers yields its issue slots
#endif
#ifdef LZ4P_PROF     // tools (variant build): cycles of thread 0 per phase, summed over the workgroups -> g_pcd_prof
__device__ unsigned long long g_pcd_prof[32];
#define PCD_PROF_DECL unsigned long long pr_acc[32] = {0}; unsigned long long pr_t0 = __builtin_readcyclecounter();
#define PCD_WAVE_ADD(i, v) { if (lane == 0u) atomicAdd(&g_pcd_prof[i], (unsigned long long)(v)); }
#define PCD_NOW() __builtin_readcyclecounter()
#define PCD_TICK(i) { const unsigned long long t_ = __builtin_readcyclecounter(); pr_acc[i] += t_ - pr_t0; pr_t0 = t_; }
#define PCD_COUNT(i, v) { pr_acc[i] += (v); }
#define PCD_PROF_FLUSH if (tid == 0u) { for (int i_ = 0; i_ < 26; ++i_) if (pr_acc[i_]) atomicAdd(&g_pcd_prof[i_], pr_acc[i_]); }
#else
#define PCD_WAVE_ADD(i, v)
#define PCD_NOW() 0ull
#define PCD_PROF_DECL
#define PCD_TICK(i)
#define PCD_COUNT(i, v)
#define PCD_PROF_FLUSH
#endif
// phases: 0 load tile, 1 walks, 2 resolve + dirty check, 3 token list, 4 batch parse + scan + cut, 5 literals, 6 dependency search,
// 7 matches (polling), 8 write-back + slide, 9 giant sequences; counts: 16 tiles, 17 rounds, 18 batches, 19 sequences, 20 giants,
// 21 / 22 turns of thread 0's polling loop with / without a ready match, 23 matches copied by thread 0's whole wavefront

// ---- roles.  With fewer blocks than half the CUs a block gets TWO workgroups: the PARSER walks the tiles (it needs nothing but the
// compressed stream) and hands every tile's token list to the COPIER through a ring of RING slots in memory; the copier never
// walks.  Parse and copy of a block overlap (the parse was a fifth of a block's time), and one huge block -- the scalar
// decompress_into -- no longer leaves 255 CUs idle with 1.  Layout of DecompressArgs::pair_ws: 64 control bytes per block (u32:
// [k] slot k holds tile number [k] - 1, [3] tiles consumed, [4 + 3 k ..] slot k's tile start, token count, tile exit, [13] the
// copier has left), then
// RING slots of token lists per block.  Hand-over: MI355X_MICROARCH.md's valid forms (plain stores, barrier, one lane: agent
// release fence, s_waitcnt, relaxed flag store / one relaxed poll, agent acquire, barrier, plain loads); workgroups are
// dispatched in index order and a block's two are neighbours, so whatever is resident makes progress; every wait is bounded
// all the same (a block that gives up is decoded by the reference-order kernel).
constexpr uint32_t PAIR_RING = 3u;
constexpr uint32_t PAIR_CTRL = 64u;
constexpr uint32_t PAIR_TOKB = (2u * MAXSEQ + 255u) / 256u * 256u;      // (the production geometry's; the test geometry's lists are shorter)
enum : uint32_t { C_EXIT = 0, C_CUT = 1, C_TOTAL = 2, C_BAD = 3, C_G_SRC = 4, C_G_LIT = 5, C_G_ML = 6, C_G_OFF = 7, C_TIMEOUT = 8, C_BAD2 = 9,
                  C_NEEDPREV = 10, C_PREV = 11, C_G_IS = 12 };

#define PCD_DPP(v, ctrl, rmask) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), (ctrl), (rmask), 0xf, false))
__device__ __forceinline__ uint32_t wave_incl_add(uint32_t v) {
    v += PCD_DPP(v, 0x111, 0xf);     // row_shr:1
    v += PCD_DPP(v, 0x112, 0xf);     // row_shr:2
    v += PCD_DPP(v, 0x114, 0xf);     // row_shr:4
    v += PCD_DPP(v, 0x118, 0xf);     // row_shr:8
    v += PCD_DPP(v, 0x142, 0xa);     // row_bcast:15 -> rows 1, 3
    v += PCD_DPP(v, 0x143, 0xc);     // row_bcast:31 -> rows 2, 3
    return v;
}
__device__ __forceinline__ uint32_t bcast(uint32_t v, uint32_t l) { return (uint32_t)__shfl((int)v, (int)l, 64); }
__device__ __forceinline__ u32x4 ld16l(const lds_u8* p) { u32x4 v; __builtin_memcpy(&v, (const void*)p, 16); return v; }
__device__ __forceinline__ void st16l(lds_u8* p, const u32x4& v) { __builtin_memcpy((void*)p, &v, 16); }
__device__ __forceinline__ u32x4 ld16g(const uint8_t* p) { u32x4 v; __builtin_memcpy(&v, p, 16); return v; }
__device__ __forceinline__ void st16g(uint8_t* p, const u32x4& v) { __builtin_memcpy(p, &v, 16); }
__device__ __forceinline__ uint32_t byte_of(const u32x4& v, uint32_t j) {     // byte j of 16 (j uniform or not: selects)
    const uint32_t w = j < 8u ? (j < 4u ? v.x : v.y) : (j < 12u ? v.z : v.w);
    return (w >> (8u * (j & 3u))) & 0xFFu;
}

// n in 1..16 bytes of v, exactly, at any LDS address: 16, or 8 / 4 / 2 / 1-byte pieces
__device__ __forceinline__ void st_exact(lds_u8* d, const u32x4& v, uint32_t n) {
    if (n >= 16u) { st16l(d, v); return; }
    const bool n8 = (n & 8u) != 0u, n4 = (n & 4u) != 0u, n2 = (n & 2u) != 0u;
    const uint32_t w4 = n8 ? v.z : v.x;                               // the dword at byte offset (n & 8)
    const uint32_t wq = n8 ? (n4 ? v.w : v.z) : (n4 ? v.y : v.x);     // the dword at byte offset (n & 12)
    if (n8) { const uint64_t t = (uint64_t)v.x | ((uint64_t)v.y << 32); __builtin_memcpy((void*)d, &t, 8); }
    if (n4) __builtin_memcpy((void*)(d + (n & 8u)), &w4, 4);
    if (n2) { const uint16_t t = (uint16_t)wq; __builtin_memcpy((void*)(d + (n & 12u)), &t, 2); }
    if (n & 1u) d[n & 14u] = (uint8_t)(wq >> (n2 ? 16 : 0));
}

// the compressed stream: the tile's bytes from LDS, anything behind them (a long sequence's tail) from memory
template <class G>
struct Rd {
    const lds_u8* ct;
    const uint8_t* gin;
    uint32_t cbase;
    __device__ __forceinline__ uint32_t operator()(uint32_t pos) const {
        const uint32_t r = pos - cbase;
        return r < G::CT + G::CM ? (uint32_t)ct[r] : (uint32_t)gin[pos];
    }
    __device__ __forceinline__ uint32_t u32(uint32_t pos) const {      // pos + 4 <= ilen
        const uint32_t r = pos - cbase;
        uint32_t v;
        if (r + 4u <= G::CT + G::CM) __builtin_memcpy(&v, (const void*)(ct + r), 4);
        else __builtin_memcpy(&v, gin + pos, 4);
        return v;
    }
};

// The sequence whose token is at p, like parse_seq -- but the usual sequence (at most one length byte per length, not at the
// block's end, inside the staged tile) is decoded from aligned dwords of the LDS tile: ONE LDS round trip for up to 4 literals
// (token, offset and length byte lie in the 8 bytes at p), two otherwise, instead of a byte read per field.  A lane's walk is a
// chain of these -- its latency is the parse's critical path -- and a wavefront pays for the byte-wise path whenever ONE of its
// lanes takes it, so that path must be rare per lane (255-chains, the block's end).
template <class G>
__device__ __attribute__((noinline)) uint32_t seq_slow(const Rd<G>& rd, uint32_t ilen, uint32_t p, Seq& s) {
    return parse_seq(rd, ilen, p, s);
}
// the walk's slow path: where the sequence at p ends, offsets not looked at (lz4_pcd_common.h)
template <class G>
__device__ __attribute__((noinline)) uint32_t walk_slow(const Rd<G>& rd, uint32_t ilen, uint32_t p) {
    Seq s;
    return parse_seq<Rd<G>, false>(rd, ilen, p, s);
}
// A WALK's hop (the kernel's walk loop): where does the sequence whose token is at p end -- parse_seq<.., false>.  Token and
// literal length byte give that, except for the match length byte of a token whose match nibble is 15, which lies right before
// the NEXT token: so every hop reads the four bytes from p - 1 on (ONE LDS round trip; seq_at needs two) and first checks that
// the previous hop's length byte, if it assumed one, is not 255 (then the previous sequence is walked again, byte-wise).

// mark_addr: LDS byte address of a word that is fetched in the same round trip (the walk's "was this position marked before"),
// or 0; its value comes back in *mark_word.
template <class G>
__device__ __forceinline__ uint32_t seq_at(const Rd<G>& rd, uint32_t ilen, uint32_t staged, uint32_t p, Seq& s,
                                           uint32_t mark_addr = 0u, uint32_t* mark_word = nullptr) {
    // Straight-line code: every lane reads the 8 aligned bytes around its token and then the 8 around its offset, whether it
    // needs them or not -- both addresses lie inside the LDS tile for every token position below CT (a length byte adds at most
    // 270 bytes, the tile is staged with CM = 1 024 behind it).  Written with plain loads, hipcc moved each read into the
    // branch that uses it: four dependent round trips and ~170 instructions per sequence.
    const uint32_t r = p - rd.cbase;
    const uint32_t base = (uint32_t)(uintptr_t)rd.ct;
    uint64_t d01;
    uint32_t mk = 0u;
    if (mark_word != nullptr) asm volatile("ds_read_b32 %0, %1" : "=v"(mk) : "v"(mark_addr) : "memory");
    asm volatile("ds_read2_b32 %0, %1 offset1:1" : "=v"(d01) : "v"(base + (r & ~3u)) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(d01), "+v"(mk) :: "memory");
    if (mark_word != nullptr) *mark_word = mk;
    const uint32_t w0 = __builtin_amdgcn_alignbyte((uint32_t)(d01 >> 32), (uint32_t)d01, r & 3u);     // token and the literal length byte
    const uint32_t t = w0 & 0xFFu, lc = t >> 4, mlc = t & 15u;
    const uint32_t l15 = lc == 15u ? 1u : 0u, e1 = (w0 >> 8) & 0xFFu;
    const uint32_t lit = lc + (l15 ? e1 : 0u);                           // (one length byte: up to 269 literals)
    const uint32_t hdr = 1u + l15;
    const uint32_t rq = r + hdr + lit;
    const uint32_t q = p + hdr + lit;                                    // the offset's position
    uint64_t e01;
    asm volatile("ds_read2_b32 %0, %1 offset1:1" : "=v"(e01) : "v"(base + (rq & ~3u)) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(e01) :: "memory");
    const uint32_t x3 = __builtin_amdgcn_alignbyte((uint32_t)(e01 >> 32), (uint32_t)e01, rq & 3u);   // the 3 bytes at q
    const uint32_t off = x3 & 0xFFFFu, e = (x3 >> 16) & 0xFFu;
    // the usual sequence: at most one length byte per length; offset, a length byte and one more byte exist (no end-of-block
    // case) and are staged
    const bool usual = r + 8u <= staged && !(l15 && e1 == 255u) && q + 3u < ilen && rq + 8u <= staged && !(mlc == 15u && e == 255u);
    if (usual) {
        s.lit_src = p + hdr; s.lit = lit; s.off = off; s.ml = 4u + mlc + (mlc == 15u ? e : 0u);
        return off != 0u ? q + 2u + (mlc == 15u ? 1u : 0u) : X_ERR;      // (offset 0: what parse_seq returns, decompress.rs:168-173)
    }
    return seq_slow(rd, ilen, p, s);
}

template <class G>
struct Ctx {
    lds_u8* lds;
    const uint8_t* gin;
    uint8_t* gout;
    uint32_t ilen, cap;
    uint32_t tid, lane, wv;
    __device__ __forceinline__ lds_u8* ct() const { return lds + G::L_CT; }
    __device__ __forceinline__ lds_u32* marks() const { return (lds_u32*)(lds + G::L_MARK); }
    __device__ __forceinline__ lds_u32* ent() const { return (lds_u32*)(lds + G::L_ENT); }
    __device__ __forceinline__ lds_u32* ext() const { return (lds_u32*)(lds + G::L_EXT); }
    __device__ __forceinline__ lds_u32* nxt() const { return (lds_u32*)(lds + G::L_NXT); }
    __device__ __forceinline__ lds_u32* rch() const { return (lds_u32*)(lds + G::L_RCH); }
    __device__ __forceinline__ lds_u16* tok() const { return (lds_u16*)(lds + G::L_TOK); }
    __device__ __forceinline__ lds_u32* bst() const { return (lds_u32*)(lds + G::L_BST); }
    __device__ __forceinline__ lds_u32* mst() const { return (lds_u32*)(lds + G::L_MST); }
    __device__ __forceinline__ lds_u16* offs() const { return (lds_u16*)(lds + G::L_OFF); }
    __device__ __forceinline__ lds_u32* nmk() const { return (lds_u32*)(lds + G::L_TOK); }   // a walk's marks until it ends (the token list's space: free while a tile is parsed)
    __device__ __forceinline__ lds_u16* tb() const { return (lds_u16*)(lds + G::L_TB); }
    __device__ __forceinline__ lds_u32* done() const { return (lds_u32*)(lds + G::L_DONE); }
    __device__ __forceinline__ lds_u32* wsum() const { return (lds_u32*)(lds + G::L_WSUM); }
    __device__ __forceinline__ volatile lds_u32* ctl() const { return (volatile lds_u32*)(lds + G::L_CTL); }
    __device__ __forceinline__ lds_u8* win() const { return lds + G::L_WIN; }

    // exclusive prefix sum of v over the workgroup (thread order); *total = the sum.  Two barriers.
    __device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* total) const {
        const uint32_t incl = wave_incl_add(v);
        lds_u32* ws = wsum();
        if (lane == 63u) ws[wv] = incl;
        __syncthreads();
        uint32_t base = 0u, tot = 0u;
#pragma unroll
        for (uint32_t i = 0; i < G::NW; ++i) {
            const uint32_t s = ws[i];
            base += i < wv ? s : 0u;
            tot += s;
        }
        __syncthreads();
        *total = tot;
        return base + incl - v;
    }

    // the same for S values per thread, laid out slot by slot (all threads' slot 0 in thread order, then slot 1, ...): ex[u] = the
    // sum of everything before (u, thread).  Two barriers for all slots.
    __device__ __forceinline__ void block_excl_scan_slots(const uint32_t (&v)[G::S], uint32_t (&ex)[G::S]) const {
        uint32_t incl[G::S];
        lds_u32* ws = wsum();
#pragma unroll
        for (uint32_t u = 0; u < G::S; ++u) {
            incl[u] = wave_incl_add(v[u]);
            if (lane == 63u) ws[u * G::NW + wv] = incl[u];
        }
        __syncthreads();
        uint32_t before = 0u;                                         // the slots before u, whole
#pragma unroll
        for (uint32_t u = 0; u < G::S; ++u) {
            uint32_t base = 0u, tot = 0u;
#pragma unroll
            for (uint32_t i = 0; i < G::NW; ++i) {
                const uint32_t s = ws[u * G::NW + i];
                base += i < wv ? s : 0u;
                tot += s;
            }
            ex[u] = before + base + incl[u] - v[u];
            before += tot;
        }
        __syncthreads();
    }

    // ---- the tile [cbase, cbase + CT + CM) of the compressed stream into LDS (never a byte behind the block)
    __device__ __forceinline__ void load_tile(uint32_t cbase) const {
        const uint32_t avail = ilen - cbase;
        const uint32_t n = avail < G::CT + G::CM ? avail : G::CT + G::CM;
        const uint8_t* src = gin + cbase;
        const uint32_t n16 = n & ~15u;
        for (uint32_t o = 16u * tid; o < n16; o += 16u * G::T) st16l(ct() + o, ld16g(src + o));
        if (tid < n - n16) ct()[n16 + tid] = src[n16 + tid];
    }

    // ---- wavefront 0: follow the exits from part 0 by pointer jumping; live parts get their true entries
    __device__ __forceinline__ void resolve(uint32_t cbase, uint32_t parts) const {
        lds_u32 *nx = nxt(), *rc = rch(), *ex = ext(), *en = ent();
        uint32_t k[G::PPL], a[G::PPL], b2[G::PPL], r[G::PPL];
#pragma unroll
        for (uint32_t j = 0; j < G::PPL; ++j) {
            k[j] = lane + 64u * j;
            if (k[j] < parts) {
                const uint32_t x = ex[k[j]];
                const uint32_t q = (x - cbase) / G::P;
                nx[k[j]] = (x < X_ERR && q < parts) ? q : G::NP;      // NP: the chain leaves the tile here (or ends, or dies)
                rc[k[j]] = k[j] == 0u ? 1u : 0u;
            }
        }
        if (lane == 0u) { nx[G::NP] = G::NP; rc[G::NP] = 0u; }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // the usual tile: every part's chain lands in the next part (sequences are short against a part), the last one leaves the
        // tile -- every part is live and nothing has to be followed
        bool adjacent = true;
#pragma unroll
        for (uint32_t j = 0; j < G::PPL; ++j)
            if (k[j] < parts) adjacent = adjacent && nx[k[j]] == (k[j] + 1u < parts ? k[j] + 1u : G::NP);
        const bool chain = __all(adjacent);
        if (chain) {
#pragma unroll
            for (uint32_t j = 0; j < G::PPL; ++j) if (k[j] < parts) rc[k[j]] = 1u;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
#pragma unroll 1
        for (uint32_t round = 0; !chain && round < ceil_log2(G::NP) + 1u; ++round) {
            // every read of the round precedes every write of the round (LDS operations of a wavefront execute in order)
#pragma unroll
            for (uint32_t j = 0; j < G::PPL; ++j) {
                a[j] = G::NP; r[j] = 0u;
                if (k[j] < parts) { a[j] = nx[k[j]]; r[j] = rc[k[j]]; }
            }
#pragma unroll
            for (uint32_t j = 0; j < G::PPL; ++j) b2[j] = nx[a[j]];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (uint32_t j = 0; j < G::PPL; ++j) {
                if (k[j] < parts) {
                    if (r[j]) rc[a[j]] = 1u;
                    nx[k[j]] = b2[j];
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
#pragma unroll
        for (uint32_t j = 0; j < G::PPL; ++j) {
            if (k[j] < parts && rc[k[j]] != 0u) {
                const uint32_t x = ex[k[j]];
                const uint32_t q = (x - cbase) / G::P;
                if (x < X_ERR && q < parts) en[q] = x;                 // one live predecessor per live part: no conflict
                else ctl()[C_EXIT] = x;                                // the one live part whose chain leaves the tile
            }
        }
    }

    // ---- n bytes, any alignment, by ONE WAVEFRONT into the window: literals from the compressed stream
    __device__ __forceinline__ void wave_literals(uint32_t dstw, uint32_t src, uint32_t n) const {
        for (uint32_t o = 16u * lane; o < n; o += 1024u) {
            const uint32_t m = n - o < 16u ? n - o : 16u;
            const uint32_t p = src + o;
            if (p + 16u <= ilen) {
                st_exact(win() + dstw + o, ld16g(gin + p), m);
            } else {
                for (uint32_t j = 0; j < m; ++j) win()[dstw + o + j] = gin[p + j];
            }
        }
    }
    // n bytes from output position src to output position dst (src + n <= dst: no overlap) by ONE WAVEFRONT; the destination
    // lies in the window (base Lo), source bytes before Lo come from the written-back output
    __device__ __forceinline__ void wave_copy(uint32_t dst, uint32_t src, uint32_t n, uint32_t Lo) const {
        for (uint32_t o = 16u * lane; o < n; o += 1024u) {
            const uint32_t m = n - o < 16u ? n - o : 16u;
            const uint32_t p = src + o;
            lds_u8* d = win() + (dst + o - Lo);
            if (p >= Lo) {
                st_exact(d, ld16l(win() + (p - Lo)), m);               // (may read up to 15 bytes behind the source: unused, inside the window's slack)
            } else if (p + 16u <= Lo) {
                st_exact(d, ld16g(gout + p), m);
            } else {
                for (uint32_t j = 0; j < m; ++j) d[j] = p + j < Lo ? gout[p + j] : win()[p + j - Lo];
            }
        }
    }
    // a match of any offset and length at output position ms by ONE WAVEFRONT: non-overlapping steps of growing size -- after
    // `done` bytes (a multiple of the offset) the `done + off` bytes from ms - off on are final and periodic
    __device__ __forceinline__ void wave_match(uint32_t ms, uint32_t off, uint32_t ml, uint32_t Lo) const {
        uint32_t donem = 0u;
        while (donem < ml) {
            const uint32_t room = donem + off;
            const uint32_t n = ml - donem < room ? ml - donem : room;
            wave_copy(ms + donem, ms - off, n, Lo);
            donem += n;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");     // the next step reads what other lanes wrote in this one
            __builtin_amdgcn_wave_barrier();
        }
    }


    // ---- the matches of one slot of a batch (sequence i of the batch in this lane: s, at output position OP + exu): find the
    // producers, copy the ones that have none, poll for the others.  Not inlined: the kernel holds S copies of a batch's state,
    // and inlined S times this code needed more registers than a 1 024-thread workgroup has (124 spilled).
    __device__ __attribute__((noinline)) void match_slot(const Seq s, const uint32_t i, const bool hm, const uint32_t exu, uint32_t OP_, uint32_t Lo_,
                                                         uint32_t cnt_) const {
        const uint32_t OP = (uint32_t)__builtin_amdgcn_readfirstlane((int)OP_), Lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)Lo_),
                       cnt = (uint32_t)__builtin_amdgcn_readfirstlane((int)cnt_);
        const uint32_t ms = OP + exu + s.lit;
        uint32_t s0 = ms - s.off;                                          // source start (hm: off <= ms)
        uint32_t s1 = s0 + s.ml < ms ? s0 + s.ml : ms;                     // source end outside its own output
        // producers: the batch's sequences whose output holds [max(s0, OP), s1) -- lo..hi, all before mine
        uint32_t lo = 1u, hi = 0u;
        auto find = [&](bool want) {
            lo = 1u; hi = 0u;
            if (want && s1 > OP) {
                // largest j < cnt with bst[j] <= a: from the first sequence that starts in a's 64 bytes (tb), forward -- two or three
                // steps (a binary search over bst[] was 11 dependent LDS round trips, a twentieth of the kernel's time)
                const uint32_t a0 = s0 > OP ? s0 : OP, a1 = s1 - 1u;
                uint32_t jl = tb()[(a0 - OP) >> 6], jh = tb()[(a1 - OP) >> 6];
                for (;;) {
                    const bool ml_ = jl < cnt && bst()[jl] <= a0, mh_ = jh < cnt && bst()[jh] <= a1;
                    if (!ml_ && !mh_) break;
                    jl += ml_ ? 1u : 0u; jh += mh_ ? 1u : 0u;
                }
                jl -= 1u; jh -= 1u;                                        // (sequence 0 starts at OP <= a: never below 0)
                if (jl < i) {                                              // (a source inside my own literals has no producer)
                    lo = jl;
                    hi = jh < i ? jh : i - 1u;
                    // the last one only counts if the source reaches into its MATCH (its literals are placed already)
                    if (hi == jh && s1 <= mst()[hi]) { if (hi == lo) { lo = 1u; hi = 0u; } else hi -= 1u; }
                }
            }
        };
        find(hm);
        // RELINKING.  A match whose whole source lies inside the MATCH of one earlier sequence j of the batch copies bytes that j
        // copies from off_j further back: out[x] = out[x - off_j] for every byte of j's match, overlapping or not (decompress.rs:
        // 410-437).  So it can read there itself instead of waiting for j -- and its new source may be free (history, literals) or
        // wait for something earlier.  Chains of copies of copies are what a batch's match phase consists of: one round takes a
        // third of the levels away (JSON tiles 109 -> 74 per batch, log lines 28 -> 19; two rounds: 56 / 15) and costs one more
        // search.  Measured, 0 / 1 / 2 / 3 rounds: 256 JSON blocks 0.198 / 0.169 / 0.157 / 0.155 ms, 64 x 1 MiB JSON 2.18 / 1.74 /
        // 1.61 / 1.63, 256 x 4 MiB log blocks 5.47 / 5.51 / 5.88 / 6.31 (shallow chains: the searches cost more than the levels).
        // So one round always, and further ones only where they pay: while at least LZ4P_RELINK_MIN of the wavefront's 64 matches
        // would relink (the share falls round by round -- JSON 52 / 37 / 25 %, log lines 43 / 25 / 13 %, text 35 / 17 / 7 %: a
        // second round on most JSON wavefronts, rarely elsewhere).  Measured with a floor of 12 / 20 / 28 lanes against one fixed
        // round: 256 JSON blocks 0.138 / 0.140 / 0.143 against 0.150 ms, 64 x 1 MiB JSON 1.42 / 1.45 / 1.49 against 1.59, 256 x 4 MiB log
        // blocks 5.14 / 4.97 / 4.84 against 4.78.
#ifndef LZ4P_RELINK
#define LZ4P_RELINK 3
#endif
#ifndef LZ4P_RELINK_MIN
#define LZ4P_RELINK_MIN 28
#endif
#pragma unroll 1
        for (uint32_t rr = 0u; rr < LZ4P_RELINK; ++rr) {
            bool rl = hm && lo == hi && s.off >= s.ml;                     // one producer; I do not read my own output
            uint32_t mj = 0u, ej = 0u, oj = 0u;
            if (rl) { mj = mst()[lo]; ej = bst()[lo + 1u]; oj = offs()[lo]; }          // (lo < i: sequence lo + 1 exists)
            rl = rl && s0 >= mj && s1 <= ej;
            if ((uint32_t)__builtin_popcountll(__ballot(rl)) < (rr == 0u ? 1u : (uint32_t)LZ4P_RELINK_MIN)) break;
            if (rl) { s0 -= oj; s1 -= oj; }
            uint32_t lo2 = lo, hi2 = hi;
            find(rl);
            if (!rl) { lo = lo2; hi = hi2; }
        }
        const uint32_t off = ms - s0;                                      // the offset after relinking (>= off)
        // one lane, 16 bytes at a time: offset >= 16, up to 256 bytes, source entirely in the window or (up to 64 bytes)
        // entirely written back; everything else (periodic, long, straddling the window's start) is copied by the whole wavefront
        const bool dep = lo <= hi;
        const bool near = s0 >= Lo;
        const bool farok = s0 + ((s.ml + 15u) & ~15u) <= Lo;
        const bool inl = off >= 16u && (near ? s.ml <= 256u : (farok && s.ml <= 64u));   // (far: four loads in flight, no more)
        lds_u8* const dstp = win() + (ms - Lo);
        const lds_u8* const srcp = win() + (s0 - Lo);          // (only used where the source lies in the window)
        auto copy_coop = [&](bool want) {
            uint64_t cm = __ballot(want);
            while (cm != 0ull) {
                const uint32_t l = (uint32_t)__builtin_ctzll(cm);
                cm &= cm - 1ull;
                wave_match(bcast(ms, l), bcast(off, l), bcast(s.ml, l), Lo);
                                    }
        };
        auto publish = [&]() {     // my match's bytes are in the window: the DONE bit follows them (release)
            __hip_atomic_fetch_or(done() + (i >> 5), 1u << (i & 31u), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        };
        // matches without a producer in this batch (history, written-back output, own literals): all at once
        bool pending = hm;
        {
            const bool go = hm && !dep;
            if (go && inl) {
                if (near) {
                    // 64 bytes of loads before their stores where no byte of those 64 is the match's own output, else 16 (one
                    // loop for both: two loops would be executed one after the other by a wavefront that holds both kinds)
                    const uint32_t grp = (off >= 64u || off >= s.ml) ? 64u : 16u;
                    for (uint32_t o = 0u; o < s.ml; o += grp) {
                        u32x4 v[4];
#pragma unroll
                        for (uint32_t k = 0u; k < 4u; ++k) if (16u * k < grp && o + 16u * k < s.ml) v[k] = ld16l(srcp + o + 16u * k);
#pragma unroll
                        for (uint32_t k = 0u; k < 4u; ++k) {
                            const uint32_t at = o + 16u * k;
                            if (16u * k < grp && at < s.ml) st_exact(dstp + at, v[k], s.ml - at < 16u ? s.ml - at : 16u);
                        }
                    }
                } else {                                  // written-back output: every load is issued before the first store waits
                    const uint8_t* sp = gout + s0;
                    u32x4 v[4];
#pragma unroll
                    for (uint32_t k = 0u; k < 4u; ++k) if (16u * k < s.ml) v[k] = ld16g(sp + 16u * k);
#pragma unroll
                    for (uint32_t k = 0u; k < 4u; ++k) if (16u * k < s.ml) st_exact(dstp + 16u * k, v[k], s.ml - 16u * k < 16u ? s.ml - 16u * k : 16u);
                }
            }
            copy_coop(go && !inl);
            if (go) { publish(); pending = false; }
        }
        // The others poll their producers' DONE bits: [lo, hi] = bits lo & 31 .. of word wl up to bit hi & 31 of word wh.  This
        // loop is what the batch waits for -- a chain of d dependent matches costs d turns -- so a turn is as little code as
        // possible: both DONE words in one round trip, one copy loop (a match with producers reads the window, from OP on:
        // never the written-back output), 16 bytes per step in order (it may read its own output).  A wavefront without
        // open matches goes on (to the next slot, then to the barrier: off the issue slots).
        const uint32_t wl = lo >> 5, wh = dep ? hi >> 5 : wl;
        const uint32_t m1 = (0xFFFFFFFFu << (lo & 31u)) & (wl == wh ? 0xFFFFFFFFu >> (31u - (hi & 31u)) : 0xFFFFFFFFu);
        const uint32_t m2 = wl == wh ? m1 : 0xFFFFFFFFu >> (31u - (hi & 31u));
        const bool wide = dep && wh - wl > 1u;                   // (whole words between the two: a source of > 32 sequences)
        const volatile lds_u32* dn = (const volatile lds_u32*)done();
        const uint32_t nfull = s.ml >> 4, rem = s.ml & 15u;
        const bool inl2 = off >= 16u && s.ml <= 256u && near;
        uint32_t spins = 0u;
        while (__any(pending)) {
            bool ready = false;
            if (pending) {
                const uint32_t d1 = dn[wl], d2 = dn[wh];
                ready = (d1 & m1) == m1 && (d2 & m2) == m2;
                if (wide) for (uint32_t w = wl + 1u; ready && w < wh; ++w) ready = dn[w] == 0xFFFFFFFFu;
            }
            if (__any(ready)) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");     // the producers' bytes behind their DONE bits
                if (ready && inl2) {     // (16 bytes a step: grouping the loads as above makes this loop, and with it every level of the batch's dependency chains, a third slower)
                    for (uint32_t k = 0u; k < nfull; ++k) st16l(dstp + 16u * k, ld16l(srcp + 16u * k));
                    if (rem != 0u) st_exact(dstp + 16u * nfull, ld16l(srcp + 16u * nfull), rem);
                }
                copy_coop(ready && !inl2);
                if (ready) { publish(); pending = false; }
            } else {
                __builtin_amdgcn_s_sleep(LZ4P_SLEEP);
                if (++spins > (1u << 22)) { ctl()[C_TIMEOUT] = 1u; break; }   // (cannot happen: the lowest open match is always ready)
            }
        }
    }

    // ---- chained batches (Linked frames): the bytes before this block's start are written by the batch's earlier blocks, i.e. by
    // other workgroups.  Called by every thread; returns true once chain_done[b - 1] says "done" (all earlier blocks complete,
    // their bytes visible to this CU), false if the predecessor gave up or did not finish in time (-> the reference-order
    // kernel decodes this block after the launch).  Workgroups are dispatched in index order, so a predecessor is running or
    // done; the wait is bounded all the same (wall clock, 100 MHz).  Protocol: MI355X_MICROARCH.md "valid forms": producer plain
    // stores -> barrier -> one lane: agent release fence, s_waitcnt, relaxed flag store; consumer: one relaxed poll, one agent
    // acquire, barrier, plain loads.
    __device__ __forceinline__ bool wait_predecessor(const uint32_t* flag) const {
        if (tid == 0u) {
            const unsigned long long t0 = wall_clock64();
            uint32_t v = 0u;
            for (;;) {
                v = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (v != 0u) break;
                if (wall_clock64() - t0 > 1000000000ull) { v = 2u; break; }     // 10 s
                __builtin_amdgcn_s_sleep(32);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            ctl()[C_PREV] = v;
        }
        __syncthreads();
        const bool ok = ctl()[C_PREV] == 1u;
        __syncthreads();
        return ok;
    }

    // ---- the whole WORKGROUP, output memory to output memory / compressed stream to output memory (a sequence longer than the window)
    __device__ __forceinline__ void block_copy(uint8_t* dst, const uint8_t* src, uint32_t n) const {
        const uint32_t n16 = n & ~15u;
        for (uint32_t o = 16u * tid; o < n16; o += 16u * G::T) st16g(dst + o, ld16g(src + o));
        if (tid < n - n16) dst[n16 + tid] = src[n16 + tid];
    }
};

template <class G>
__global__ void __launch_bounds__(G::T) lz4_decompress_pcd_kernel(DecompressArgs a, int32_t redo_code) {
    constexpr uint32_t SPLAT_MAX = G::WIN - 32u < 4096u ? G::WIN - 32u : 4096u;       // periods the giant-match path repeats from LDS
    extern __shared__ __attribute__((aligned(16))) uint8_t pcd_lds[];
    const bool paired = a.pair_ws != nullptr;
    const uint32_t b = paired ? blockIdx.x >> 1 : blockIdx.x;
    const uint32_t role = paired ? 1u + (blockIdx.x & 1u) : 0u;     // 0: the block's only workgroup, 1: its parser, 2: its copier
    if (b >= a.n) return;
    Ctx<G> X;
    X.lds = (lds_u8*)pcd_lds;
    X.gin = a.in_base + a.in_off[b];
    X.gout = a.out_base + a.out_off[b];
    X.ilen = a.in_len[b];
    X.cap = a.out_cap[b];
    X.tid = threadIdx.x;
    X.lane = threadIdx.x & 63u;
    X.wv = threadIdx.x >> 6;
    const uint32_t tid = X.tid, lane = X.lane;
    volatile lds_u32* ctl = X.ctl();
    if (X.ilen == 0u) {                                           // decompress.rs:207-209: the reference-order kernel reports it
        if (tid == 0u && role != 1u) {
            a.status[b] = redo_code; a.out_len[b] = 0u;
            if (a.chain_done != nullptr) __hip_atomic_store(a.chain_done + b, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    uint32_t* const pcw = paired ? (uint32_t*)(a.pair_ws + (size_t)PAIR_CTRL * b) : nullptr;                      // this block's control words
    uint8_t* const ptok = paired ? a.pair_ws + (size_t)PAIR_CTRL * PCD_PAIR_MAX_BLOCKS + (size_t)PAIR_TOKB * PAIR_RING * b : nullptr;
    uint32_t tile_no = 0u;
    if (tid < 32u) ctl[tid] = 0u;
    // prefix mode (Linked frames): the sink already holds OP0 bytes that matches may refer to; in a chained batch they are being
    // written by the earlier blocks of the batch
    const uint32_t OP0 = a.out_pos != nullptr ? a.out_pos[b] : 0u;
    const uint32_t prev_b = a.chain_prev != nullptr ? a.chain_prev[b] : b - 1u;        // (block 0: 0xFFFFFFFF either way)
    const uint32_t* const prev_flag = (a.chain_done != nullptr && prev_b < b) ? a.chain_done + prev_b : nullptr;
    bool prev_ok = prev_flag == nullptr;      // the bytes before OP0 are final
    uint32_t cbase = 0u;       // the tile's first byte: a true token position
    uint32_t OP = OP0;         // output position: everything before it is written back
    uint32_t hist = 0u;        // window bytes [0, hist) hold output [OP - hist, OP)
    bool ended = false, bad = a.debug_giveup == b + 1u;      // (tests: a block that gives up without an error of its own)
    if (bad) ended = true;
    __syncthreads();
    PCD_PROF_DECL

    while (!ended) {
        uint32_t tile_exit = X_ERR, ntok = 0u;
        if (role == 2u) {
            // ---- copier of a pair: the tile comes parsed.  Wait for slot tile_no % RING, take the token list, give the slot back
            const uint32_t k = tile_no % PAIR_RING;
            if (tid == 0u) {
                const unsigned long long t0 = wall_clock64();
                uint32_t ok = 1u;
                while (__hip_atomic_load(pcw + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != tile_no + 1u) {
                    if (wall_clock64() - t0 > 1000000000ull) { ok = 0u; break; }      // 10 s
                    __builtin_amdgcn_s_sleep(8);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                ctl[C_G_SRC] = pcw[4u + 3u * k]; ctl[C_G_LIT] = pcw[5u + 3u * k]; ctl[C_G_ML] = pcw[6u + 3u * k];
                ctl[C_PREV] = ok;
            }
            __syncthreads();
            const bool ok = ctl[C_PREV] != 0u;
            cbase = ctl[C_G_SRC]; ntok = ctl[C_G_LIT]; tile_exit = ctl[C_G_ML];
            __syncthreads();                                       // (the control words are written again below)
            if (!ok || tile_exit == X_ERR || ntok > G::MAXSEQ) { bad = true; break; }
            X.load_tile(cbase);
            const uint8_t* src = ptok + (size_t)PAIR_TOKB * k;
            for (uint32_t o = 16u * tid; o < 2u * ntok; o += 16u * G::T) st16l((lds_u8*)X.tok() + o, ld16g(src + o));
            __syncthreads();
            if (tid == 0u) __hip_atomic_store(pcw + 3, tile_no + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (everybody's loads have landed: their values are in LDS)
        } else {
        // ================================================================ PARSE one tile
        X.load_tile(cbase);
        }
        const uint32_t span = X.ilen - cbase;
        const uint32_t parts = span >= G::CT ? G::NP : (span + G::P - 1u) / G::P;
        if (role != 2u) for (uint32_t w = tid; w < G::MW; w += G::T) { X.marks()[w] = 0u; X.nmk()[w] = 0u; }
        const uint32_t staged = span < G::CT + G::CM ? span : G::CT + G::CM;      // bytes of the block in the LDS tile
        const Rd<G> rd{X.ct(), X.gin, cbase};
        if (role != 2u) {
        uint32_t my_e = cbase + tid * G::P;
        if (tid < parts) { X.ent()[tid] = my_e; X.ext()[tid] = X_ERR; }
        bool dirty = tid < parts;
        if (tid == 0u) ctl[C_EXIT] = X_ERR;
        __syncthreads();
        bool settled = false;
        PCD_TICK(0) PCD_COUNT(16, 1)
        for (uint32_t it = 0u; it < G::MAX_ITERS; ++it) {
            bool exit_changed = false;
            if (dirty) {
                // Lane k walks part k from its entry.  A walk that lands on a position the part's PREVIOUS walk marked is
                // identical to it from there on: it stops, keeps those marks and the exit (the first walk of a tile, from an
                // assumed entry, meets no marks: the tile's are cleared).  So a part is walked once in full and then only as far
                // as its chains differ -- a few sequences.
                lds_u32* mk = X.marks() + tid * G::PW;
                lds_u32* nk = X.nmk() + tid * G::PW;                    // this walk's marks (zero between walks)
                const uint32_t pend = cbase + (tid + 1u) * G::P;
                // One hop = one LDS round trip and ~50 instructions: the four bytes from the byte before the token (hop_at's layout),
                // the part's old mark word in the same trip, the new mark set with an LDS OR (no per-word selects in registers),
                // conditions as integers.  What a hop costs IS the parse: a round of walks lasts as long as its slowest lane's chain
                // of hops (written with bool flags and the marks in registers: 120 instructions, 1 040 cycles a hop).
                // Everything unusual -- a length byte of 255, the block's or the staged bytes' end, a second length byte found
                // behind an assumed single one -- leaves through ONE wave-level test.
                const uint32_t ctb = (uint32_t)(uintptr_t)rd.ct, mkb = (uint32_t)(uintptr_t)X.marks(), nkb = (uint32_t)(uintptr_t)X.nmk();
                uint32_t p = my_e, prev_p = my_e, x = 0u, mrg = 0xFFFFFFFFu;   // mrg: where this walk fell into step with the old one (byte of the tile)
                uint32_t p15 = 0u;                                      // 1: the hop from prev_p assumed ONE match length byte, at p - 1
                bool act = true;
                const unsigned long long pr_w0 = PCD_NOW(); (void)pr_w0;
                while (act) {
                    const uint32_t r = p - cbase, ra = r - 1u, wofs = (r >> 5) << 2;
                    uint64_t d01;
                    uint32_t oldw;
                    asm volatile("ds_read_b32 %0, %1" : "=v"(oldw) : "v"(mkb + wofs) : "memory");
                    asm volatile("ds_read2_b32 %0, %1 offset1:1" : "=v"(d01) : "v"(ctb + (ra & ~3u)) : "memory");
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(d01), "+v"(oldw) :: "memory");
                    const uint32_t w = __builtin_amdgcn_alignbyte((uint32_t)(d01 >> 32), (uint32_t)d01, ra & 3u);   // byte 0: before the token, 1: token, 2: literal length byte
                    const uint32_t lc = (w >> 12) & 15u, e1 = (w >> 16) & 0xFFu;
                    const uint32_t l15 = lc == 15u ? 1u : 0u, b15 = (w & 0xF00u) == 0xF00u ? 1u : 0u;
                    const uint32_t q = p + 1u + l15 + lc + (l15 ? e1 : 0u);         // the offset's position
                    const uint32_t nx = q + 2u + b15;
                    const uint32_t bit = 1u << (r & 31u);
                    const bool over = p >= pend;                       // the chain has left the part: p is the exit once p - 1 is checked
                    const bool inl = r + 8u <= staged;
                    const bool prev255 = (w & 0xFFu) == 0xFFu;
                    const bool usual = !(l15 != 0u && e1 == 255u) && q + 3u < X.ilen;
                    const bool rare = !inl || (p15 != 0u && prev255) || (!over && !usual);
                    PCD_COUNT(24, 1)
                    if (__any(rare)) {
                        if (rare) {
                            const bool redo = p15 != 0u && (inl ? prev255 : rd(p - 1u) == 0xFFu);
                            p15 = 0u;
                            if (redo) {                                // p is not a token: the sequence before it is longer
                                p = walk_slow(rd, X.ilen, prev_p);
                                if (p >= X_ERR) { x = p; act = false; }
                            } else if (over) {
                                x = p; act = false;
                            } else if ((oldw & bit) != 0u) {
                                mrg = r; act = false;
                            } else {
                                __hip_atomic_fetch_or((lds_u32*)(uintptr_t)(nkb + wofs), bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                                const uint32_t n2 = (inl && usual) ? nx : walk_slow(rd, X.ilen, p);
                                prev_p = p;
                                if (n2 >= X_ERR) { x = n2; act = false; } else { p = n2; p15 = (inl && usual) ? b15 : 0u; }
                            }
                        }
                    }
                    if (!rare) {
                        if (over) {                                    // (the length byte before the exit is an ordinary one)
                            x = p; act = false;
                        } else if ((oldw & bit) != 0u) {
                            mrg = r; act = false;
                        } else {
                            __hip_atomic_fetch_or((lds_u32*)(uintptr_t)(nkb + wofs), bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                            prev_p = p;
                            p15 = b15;
                            p = nx;                                    // (usual: a real position, below X_ERR)
                        }
                    }
                }
                PCD_COUNT(25, PCD_NOW() - pr_w0)
                // the part's marks: this walk's, and behind the point where it fell into step the old walk's (whose exit stands)
                const bool merged = mrg != 0xFFFFFFFFu;
                const uint32_t mw = merged ? (mrg >> 5) - tid * G::PW : G::PW, mbit = mrg & 31u;
#pragma unroll
                for (uint32_t w = 0; w < G::PW; ++w) {
                    const uint32_t nw = nk[w], old = mk[w];
                    nk[w] = 0u;
                    mk[w] = w < mw ? nw : (w == mw ? (nw | (old & (0xFFFFFFFFu << mbit))) : old);
                }
                if (!merged) {
                    exit_changed = X.ext()[tid] != x;              // (a first walk: the slot holds X_ERR ... and an entry that really
                    X.ext()[tid] = x;                              //  leads to X_ERR again changes nothing: the chain dies there either way)
                    exit_changed = exit_changed || it == 0u;
                }
            }
            // no exit changed since the exits were last followed: every walk of this round fell into step with its part's old chain,
            // entries and live parts are what the last round found -- the tile is settled without following them again
            const int any_exit = __syncthreads_or(exit_changed ? 1 : 0);
            PCD_COUNT(it == 0u ? 21 : 22, PCD_NOW() - pr_t0) PCD_TICK(1) PCD_COUNT(17, 1)
            if (!any_exit) { settled = true; break; }
            if (X.wv == 0u) X.resolve(cbase, parts);
            __syncthreads();
            dirty = false;
            if (tid < parts) {
                const uint32_t e = X.ent()[tid];
                if (e != my_e) { my_e = e; dirty = true; }
            }
            const int any_dirty = __syncthreads_or(dirty ? 1 : 0);
            PCD_TICK(2)
            if (!any_dirty) { settled = true; break; }
        }
        tile_exit = ctl[C_EXIT];
        if (!settled) tile_exit = X_ERR;
        if (tile_exit != X_ERR) {
        // ---- the tile's sequences: set bits of the live parts, in order

            uint32_t word = 0u;
            if (tid < G::MW) {
                const uint32_t part = tid / G::PW;
                if (part < parts && X.rch()[part] != 0u) word = X.marks()[tid];
            }
            uint32_t at = X.block_excl_scan((uint32_t)__builtin_popcount(word), &ntok);
            while (word != 0u) {
                const uint32_t bit = (uint32_t)__builtin_ctz(word);
                word &= word - 1u;
                X.tok()[at++] = (uint16_t)(tid * 32u + bit);
            }
        }
        __syncthreads();
        PCD_TICK(3)
        if (role == 1u) {
            // ---- parser of a pair: hand the tile over (an X_ERR exit too: the copier gives the block to the reference-order kernel)
            const uint32_t k = tile_no % PAIR_RING;
            if (tid == 0u) {
                const unsigned long long t0 = wall_clock64();
                uint32_t ok = 1u;
                while (tile_no - __hip_atomic_load(pcw + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= PAIR_RING) {
                    if (__hip_atomic_load(pcw + 13, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u || wall_clock64() - t0 > 1000000000ull) { ok = 0u; break; }
                    __builtin_amdgcn_s_sleep(8);
                }
                ctl[C_PREV] = ok;
            }
            __syncthreads();
            if (ctl[C_PREV] == 0u) return;                          // the copier is gone (it has given the block up, or will)
            uint8_t* dst = ptok + (size_t)PAIR_TOKB * k;
            for (uint32_t o = 16u * tid; o < 2u * ntok; o += 16u * G::T) st16g(dst + o, ld16l((const lds_u8*)X.tok() + o));
            __syncthreads();                                       // every thread's stores precede the fence below
            if (tid == 0u) {
                pcw[4u + 3u * k] = cbase; pcw[5u + 3u * k] = ntok; pcw[6u + 3u * k] = tile_exit;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(pcw + k, tile_no + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (tile_exit >= X_ERR) return;                         // X_END: the last tile; X_ERR: nothing to parse behind it
            cbase = tile_exit;
            tile_no += 1u;
            __syncthreads();                                       // the tile's LDS is free
            continue;
        }
        }
        if (tile_exit == X_ERR) { bad = true; break; }             // (uniform)
        ended = tile_exit == X_END;

        // ================================================================ COPY: batches of consecutive sequences
        // A batch is up to S sequences per lane: sequence i of the batch belongs to lane i mod T, slot i / T -- the phases whose cost
        // is latency and barriers (token decode, prefix sum, cut, literal loads, write-back) are paid once for S x T sequences.
        constexpr uint32_t S = G::S, BN = G::T * G::S;
        uint32_t idx = 0u;
        while (idx < ntok) {
            const uint32_t m = ntok - idx < BN ? ntok - idx : BN;
            Seq sq[S];
            uint32_t len[S], lenc[S];
            bool perr = false;
#pragma unroll
            for (uint32_t u = 0; u < S; ++u) {
                const uint32_t i = u * G::T + tid;
                sq[u].lit_src = 0u; sq[u].lit = 0u; sq[u].ml = 0u; sq[u].off = 0u;
                len[u] = 0u;
                if (i < m) {
                    const uint32_t nx = seq_at(rd, X.ilen, staged, cbase + X.tok()[idx + i], sq[u]);
                    perr = perr || nx == X_ERR || (nx == X_END && !(ended && idx + i + 1u == ntok));
                    len[u] = sq[u].lit + sq[u].ml;                 // (both < 2^31)
                }
                lenc[u] = len[u] <= G::GIANT ? len[u] : G::WNEW + 1u;
            }
            if (tid == 0u) ctl[C_CUT] = m;
            // output position of every sequence: slot by slot, a slot's lanes in order (one pass of barriers for all slots)
            uint32_t ex[S];
            X.block_excl_scan_slots(lenc, ex);                       // (its barriers publish C_CUT)
#pragma unroll
            for (uint32_t u = 0; u < S; ++u) {
                const uint32_t i = u * G::T + tid;
                if (i < m && ex[u] + lenc[u] > G::WNEW) __hip_atomic_fetch_min((lds_u32*)(X.lds + G::L_CTL) + C_CUT, i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            if (perr) ctl[C_BAD] = 1u;
            __syncthreads();
            const uint32_t cnt = ctl[C_CUT];
            PCD_TICK(4)
            if (ctl[C_BAD] != 0u) { bad = true; break; }
            if (cnt == 0u) {
                // ---- a sequence longer than the window (or than Geo::GIANT): alone, by the whole workgroup, on the output itself.  Round 6: the
                // sequences BEHIND it that are giants too (a block of zeros is nothing else: one 48 KiB match per window of the encoder) follow
                // in the same pass -- their lanes hold them already; parsed again per giant, 85 times per 4 MiB block, the parse was 40 % of it
                for (uint32_t gi = 0u;;) {
                    if (tid == gi % G::T) {                         // sequence gi of this pass belongs to lane gi mod T, slot gi / T
                        Seq sv = sq[0];
                        uint32_t lc = lenc[0];
#pragma unroll
                        for (uint32_t u = 1; u < S; ++u)
                            if (gi / G::T == u) { sv = sq[u]; lc = lenc[u]; }
                        ctl[C_G_SRC] = sv.lit_src; ctl[C_G_LIT] = sv.lit; ctl[C_G_ML] = sv.ml; ctl[C_G_OFF] = sv.off;
                        ctl[C_G_IS] = (gi < m && lc > G::WNEW) ? 1u : 0u;
                    }
                    __syncthreads();
                    if (ctl[C_G_IS] == 0u) break;                   // (uniform) the next one fits a batch: back to the batch path, which parses from it on
                    const uint32_t g_src = ctl[C_G_SRC], g_lit = ctl[C_G_LIT], g_ml = ctl[C_G_ML], g_off = ctl[C_G_OFF];
                    if (g_lit > X.cap - OP) { bad = true; break; }                      // OutputTooSmall
                    X.block_copy(X.gout + OP, X.gin + g_src, g_lit);
                    OP += g_lit;
                    __syncthreads();                                                    // (workgroup-scope release / acquire: the bytes are visible)
                    if (g_ml != 0u) {
                        if (g_off > OP || g_ml > X.cap - OP) { bad = true; break; }     // OffsetOutOfBounds / OutputTooSmall
                        if (!prev_ok && OP - g_off < OP0) {                             // reads bytes of an earlier block of the chain
                            if (!X.wait_predecessor(prev_flag)) { bad = true; break; }
                            prev_ok = true;
                        }
                        uint32_t donem = 0u;
                        if (g_off <= SPLAT_MAX) {
                            // a short period (offset 1: a run of one byte, decompress_safe.rs:311-313): the period, and 16 bytes of its
                            // repetition, go to the (unused) window once; every thread then stores 16 bytes of the pattern per turn,
                            // read from the window at its position's phase -- stores only, instead of log2(ml / off) rounds of
                            // memory-to-memory copies with a barrier each (16 x 4 MiB of zeros: 6 GB/s in round 3)
                            const uint8_t* pat = X.gout + OP - g_off;
                            for (uint32_t o = tid; o < g_off + 16u; o += G::T) X.win()[o] = pat[o % g_off];
                            __syncthreads();
                            uint32_t phase = (16u * tid) % g_off;
                            const uint32_t hop = (16u * G::T) % g_off;
                            uint8_t* dstp = X.gout + OP;
                            for (uint32_t o = 16u * tid; o < g_ml; o += 16u * G::T) {
                                const u32x4 v = ld16l(X.win() + phase);
                                const uint32_t mrem = g_ml - o;
                                if (mrem >= 16u) st16g(dstp + o, v);
                                else for (uint32_t j = 0; j < mrem; ++j) dstp[o + j] = (uint8_t)byte_of(v, j);
                                phase += hop;
                                phase -= phase >= g_off ? g_off : 0u;
                            }
                            donem = g_ml;
                            __syncthreads();
                        }
                        while (donem < g_ml) {
                            const uint32_t room = donem + g_off;
                            const uint32_t n = g_ml - donem < room ? g_ml - donem : room;
                            X.block_copy(X.gout + OP + donem, X.gout + OP - g_off, n);
                            donem += n;
                            __syncthreads();
                        }
                        OP += g_ml;
                    }
                    hist = 0u;
                    idx += 1u;
                    gi += 1u;
                    PCD_TICK(9) PCD_COUNT(20, 1)
                    if (gi >= m) break;
                    __syncthreads();                                // (every thread has read the control words before they are written again)
                }
                if (bad) break;
                continue;
            }
            // ---- a batch of cnt sequences: [OP, OP + total) in the window behind the history
            const uint32_t Lo = OP - hist;
            bool has_m[S];
#pragma unroll
            for (uint32_t u = 0; u < S; ++u) {
                const uint32_t i = u * G::T + tid;
                const uint32_t ms = OP + ex[u] + sq[u].lit;        // where the match starts
                if (i + 1u == cnt) ctl[C_TOTAL] = ex[u] + len[u];
                if (i < cnt) {
                    X.bst()[i] = OP + ex[u]; X.mst()[i] = ms; X.offs()[i] = (uint16_t)sq[u].off;
                    // where does a position of the batch's output come from?  tb[k] = the first sequence that starts at or behind byte
                    // 64 k: the sequence behind me, for every 64-byte boundary inside me (or at my end)
                    if (i == 0u) X.tb()[0] = 0u;
                    for (uint32_t k = (ex[u] >> 6) + 1u; (k << 6) <= ex[u] + len[u]; ++k) X.tb()[k] = (uint16_t)(i + 1u);
                }
                has_m[u] = i < cnt && sq[u].ml != 0u;
                if (has_m[u] && sq[u].off > ms) ctl[C_BAD2] = 1u;  // OffsetOutOfBounds (decompress.rs:398-400)
                if (!prev_ok && has_m[u] && ms - sq[u].off < OP0) ctl[C_NEEDPREV] = 1u;   // a match reaches into an earlier block of the chain
                // DONE bits: set for sequences without a match (and for the slots behind the batch)
                const uint64_t nm = __ballot(!has_m[u]);
                if (lane == 0u) { X.done()[u * (G::T / 32u) + 2u * X.wv] = (uint32_t)nm; X.done()[u * (G::T / 32u) + 2u * X.wv + 1u] = (uint32_t)(nm >> 32); }
                // literals: short runs by their lane, long ones by the wavefront
                const uint32_t dstw = hist + ex[u];
                const bool mine = i < cnt && sq[u].lit != 0u;
                const bool shortl = mine && sq[u].lit <= 32u && sq[u].lit_src + 32u <= X.ilen;
                if (shortl) {
                    st_exact(X.win() + dstw, ld16g(X.gin + sq[u].lit_src), sq[u].lit < 16u ? sq[u].lit : 16u);
                    if (sq[u].lit > 16u) st_exact(X.win() + dstw + 16u, ld16g(X.gin + sq[u].lit_src + 16u), sq[u].lit - 16u);
                }
                uint64_t lm = __ballot(mine && !shortl);
                while (lm != 0ull) {
                    const uint32_t l = (uint32_t)__builtin_ctzll(lm);
                    lm &= lm - 1ull;
                    X.wave_literals(bcast(dstw, l), bcast(sq[u].lit_src, l), bcast(sq[u].lit, l));
                }
            }
            __syncthreads();                                       // literals placed, bst[] / DONE / C_TOTAL / C_BAD2 published
            PCD_TICK(5) PCD_COUNT(18, 1) PCD_COUNT(19, cnt)
            const uint32_t total = ctl[C_TOTAL];
            if (ctl[C_BAD2] != 0u || total > X.cap - OP) { bad = true; break; }    // ... / OutputTooSmall somewhere in the batch
            if (!prev_ok && ctl[C_NEEDPREV] != 0u) {
                if (!X.wait_predecessor(prev_flag)) { bad = true; break; }
                prev_ok = true;
            }
            // ---- matches, slot by slot: a slot's sequences only wait for sequences before them, so a wavefront that is through
            // with slot u goes on to slot u + 1 while others still poll (no barrier in between)
#pragma unroll
            for (uint32_t u = 0; u < S; ++u) X.match_slot(sq[u], u * G::T + tid, has_m[u], ex[u], OP, Lo, cnt);
            __syncthreads();
            PCD_TICK(7)
            if (ctl[C_TIMEOUT] != 0u) { bad = true; break; }
            // ---- write the batch back, slide the history
            for (uint32_t o = 16u * tid; o < total; o += 16u * G::T) {
                const u32x4 v = ld16l(X.win() + hist + o);
                const uint32_t mrem = total - o;
                if (mrem >= 16u) st16g(X.gout + OP + o, v);
                else for (uint32_t j = 0; j < mrem; ++j) X.gout[OP + o + j] = (uint8_t)byte_of(v, j);
            }
            const uint32_t have = hist + total;
            const uint32_t keep = have < G::HIST ? have : G::HIST;
            const uint32_t from = have - keep;
            if (from != 0u) {
                u32x4 v[G::KP];
#pragma unroll
                for (uint32_t j = 0; j < G::KP; ++j) {
                    const uint32_t o = 16u * (tid + G::T * j);
                    if (o < keep) v[j] = ld16l(X.win() + from + o);
                }
                __syncthreads();
#pragma unroll
                for (uint32_t j = 0; j < G::KP; ++j) {
                    const uint32_t o = 16u * (tid + G::T * j);
                    if (o < keep) st16l(X.win() + o, v[j]);
                }
            }
            OP += total;
            hist = keep;
            idx += cnt;
            __syncthreads();                                       // window and written-back output are consistent for the next batch
            PCD_TICK(8)
        }
        if (bad) break;
        cbase = tile_exit;
        tile_no += 1u;
        __syncthreads();                                           // the tile's LDS is free
    }
    PCD_PROF_FLUSH
    if (role == 2u && tid == 0u) __hip_atomic_store(pcw + 13, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (a parser still waiting for a slot goes home)
    if (a.chain_done != nullptr) {
        // my bytes are complete: say so once every earlier block has (a waiter wants ALL of its prefix), or pass the failure on
        __syncthreads();                                           // every thread's stores precede the fence below
        if (!bad && !prev_ok) bad = !X.wait_predecessor(prev_flag);
        if (tid == 0u) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(a.chain_done + b, bad ? 2u : 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (tid == 0u) {
        if (bad) { a.status[b] = redo_code; a.out_len[b] = 0u; }
        else { a.status[b] = 0; a.out_len[b] = OP - OP0; }
    }
}

template <class G>
static hipError_t launch_geo(const DecompressArgs& a, int32_t redo_code, hipStream_t s) {
    auto kern = lz4_decompress_pcd_kernel<G>;
    if (G::LDS_BYTES > 65536u) {   // the attribute is per device: remember which devices have it (per instantiation)
        static unsigned long long have = 0ull;   // benign race: setting it twice is harmless
        int dev = 0;
        (void)hipGetDevice(&dev);
        const unsigned long long bit = 1ull << (dev & 63);
        if (!(have & bit)) {
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES);
            if (e != hipSuccess) return e;
            have |= bit;
        }
    }
    hipLaunchKernelGGL(kern, dim3(a.pair_ws != nullptr ? 2u * a.n : a.n), dim3(G::T), G::LDS_BYTES, s, a, redo_code);
    return hipGetLastError();
}

}  // namespace pcd

// one workgroup per block; blocks it marks (status redo_code) are decoded again by launch_decompress (only_status = redo_code).
// geometry 1: the small geometry (2 KiB tiles, 64-byte parts, 128 lanes, 0.5 + 1 KiB window) that puts every kind of boundary
// inside small inputs -- tests only; 2 / 3: the medium-batch geometries (GeoMid256 / GeoMid512).
size_t decompress_pcd_pair_ws_bytes() {
    return (size_t)PCD_PAIR_MAX_BLOCKS * (pcd::PAIR_CTRL + (size_t)pcd::PAIR_TOKB * pcd::PAIR_RING);
}

hipError_t launch_decompress_pcd(const DecompressArgs& a, int32_t redo_code, hipStream_t s, int geometry) {
    if (a.n == 0u) return hipSuccess;
    if (a.pair_ws != nullptr && a.n > PCD_PAIR_MAX_BLOCKS) return hipErrorInvalidValue;
    if (a.dict_base != nullptr) return hipErrorInvalidValue;   // external dictionary: lz4_decompress.hip
    if (geometry == 2 || geometry == 3) {
        if (a.pair_ws != nullptr) return hipErrorInvalidValue;
        return geometry == 2 ? pcd::launch_geo<pcd::GeoMid256>(a, redo_code, s) : pcd::launch_geo<pcd::GeoMid512>(a, redo_code, s);
    }
    return geometry == 1 ? pcd::launch_geo<pcd::GeoTest>(a, redo_code, s) : pcd::launch_geo<pcd::GeoProd>(a, redo_code, s);
}

}  // namespace lz4flex_dev

#ifdef LZ4P_PROF
extern "C" int lz4flex_debug_pcd_prof(unsigned long long* vals, int reset) {
    if (reset) {
        unsigned long long z[32] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(lz4flex_dev::pcd::g_pcd_prof), z, sizeof z);
        return 0;
    }
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(vals, HIP_SYMBOL(lz4flex_dev::pcd::g_pcd_prof), 256);
    return 0;
}
#endif
