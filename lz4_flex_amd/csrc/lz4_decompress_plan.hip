// lz4_decompress_plan.hip -- the PLAN half of the plan / replay decoder: turns LZ4 blocks into the copy plans
// lz4_decompress_replay.hip executes (format and the record emitter: lz4_plan_common.h).  Everything of the reference's
// decode loop that is not a copy happens here: the token chain (src/block/decompress.rs:244-332), every bounds check
// (:346-348, :366-375, :398-402, :439-443), the output positions.  It diagnoses nothing: a block with ANY irregularity (an
// error, a sink that is too small, lengths or positions the 4-byte records cannot hold, a plan that outgrows its slot) is left
// marked for the reference-order kernel (lz4_decompress.hip), which decodes it again and names the exact error.
//
// ONE WAVEFRONT PER BLOCK, the 64 lanes are the parallelism inside the block.  The compressed stream is consumed in TILES of
// 4 KiB staged in LDS; a tile is cut into 64 PARTS of 64 bytes and lane k walks part k's token chain -- from an ASSUMED entry,
// the part's first byte (lane 0: the tile's true entry).  A chain started at a wrong byte falls into step with the true chain
// after a few sequences, and two chains that share a position are identical from there on.  So:
//   1. every lane walks its part (lz4_pcd_common.h parse_seq: the reference's checks), leaving 8-byte sequence descriptors in
//      LDS, a 64-bit mask of the token positions it visited, and where its chain LEAVES the part (its exit);
//   2. the exits are followed from the tile's entry (a scalar loop over lanes: v_readlane): that gives every part its true
//      entry -- if its predecessor's exit is true;
//   3. lanes whose walk did not start at their true entry walk again from it, but only until they land on a position of
//      their first walk (the descriptors behind it stand); if that changes an exit, 2. and 3. repeat (real data: once);
//   4. prefix sums over the lanes' decoded bytes give every part its output position; the lanes run the record emitter over
//      their sequences once to COUNT steps (the plan is a flat array: a part has to know where its steps go), prefix sums
//      again, and a second time to WRITE them.
// A part closes its last step where it ends (a step never holds pieces of two parts): 64-byte parts cost a JSON block 6 %
// more steps than the host model's serial emission.
// Round 5: compiled in -DLZ4FLEX_TOOLS builds only (decompress_variant 9 is refused by the product library): the plan kernel lost to the
// default dispatch on every shape measured; the replay kernel and the record format it feeds stay (tests/test_gpu_replay.py,
// the fused decoder's copy engine descends from it).
#ifdef LZ4FLEX_TOOLS
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lz4_device.h"
#include "lz4_pcd_common.h"
#include "lz4_plan_common.h"

namespace lz4flex_dev {
namespace plan {

typedef uint8_t __attribute__((address_space(3))) lds_u8;
typedef uint32_t __attribute__((address_space(3))) lds_u32;
typedef uint32_t __attribute__((ext_vector_type(4))) u32x4;
typedef uint32_t __attribute__((ext_vector_type(2))) u32x2;
using pcd::X_END;
using pcd::X_ERR;

constexpr uint32_t PT = 4096u;           // compressed bytes per tile
constexpr uint32_t PB = 64u;             // bytes per part
constexpr uint32_t NPART = PT / PB;      // 64: one per lane
constexpr uint32_t PMARGIN = 128u;       // bytes behind the tile staged with it (a walk's last sequence reads past its part; beyond: global memory)
constexpr uint32_t DCAP = 22u;           // descriptors per part: a sequence with a match is at least 3 bytes
constexpr uint32_t PCAP = 6u;            // ... and of a second walk before it lands on the first one (else the part is walked afresh)
constexpr uint32_t PLAN_SLOT_WORDS = 14336u; // words of the plan array per block (56 KiB: 2 x a JSON block's plan; a block whose plan outgrows it is irregular)
constexpr uint32_t MAX_TAIL_SLOT = 64u;  // tail records a block's slot has room for (a tail is the block's last < 16 compressed bytes' worth of pieces)
// LDS.  Lane k reads part k of the tile, its descriptor lists, over and over: every per-lane area is laid out at an ODD dword stride
// so that the 64 lanes hit 64 different banks (a tile stored flat puts lanes k and k + 4 on the same bank: 16-way conflicts on every
// read; the first version spent most of its time there).  The tile: 4 bytes of padding behind every 64-byte part; the descriptor
// lists: the two words of a descriptor in two arrays of (capacity + 1) dwords per lane.
constexpr uint32_t TILE_PARTS = (PT + PMARGIN) / PB;
constexpr uint32_t LDS_TILE = 0u;
constexpr uint32_t TILE_LDS = TILE_PARTS * (PB + 4u);
constexpr uint32_t MSTRIDE = DCAP + 1u, PSTRIDE = PCAP + 1u;          // dwords per lane
constexpr uint32_t LDS_MA = TILE_LDS, LDS_MB = LDS_MA + NPART * MSTRIDE * 4u, LDS_PA = LDS_MB + NPART * MSTRIDE * 4u, LDS_PB = LDS_PA + NPART * PSTRIDE * 4u;
constexpr uint32_t LDS_BYTES = LDS_PB + NPART * PSTRIDE * 4u;
static_assert(NPART == 64u && MSTRIDE % 2u == 1u && PSTRIDE % 2u == 1u && (PT + PMARGIN) % PB == 0u, "geometry");
__device__ __forceinline__ uint32_t tile_at(uint32_t r) { return r + 4u * (r / PB); }          // LDS offset of tile byte r

// a sequence, 8 bytes: a = token position relative to the tile (13 bits) | literal length << 13 (16 bits); b = offset | match length << 16
// (match length 0: the block's last sequence).  Literal and match lengths beyond 65 535 make a block irregular.
typedef u32x2 Desc;      // .x = a, .y = b

__device__ __forceinline__ uint32_t wave_incl_add(uint32_t v) {
#define LZ4P_DPP(v, ctrl, rmask) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), (ctrl), (rmask), 0xf, false))
    v += LZ4P_DPP(v, 0x111, 0xf);     // row_shr:1
    v += LZ4P_DPP(v, 0x112, 0xf);     // row_shr:2
    v += LZ4P_DPP(v, 0x114, 0xf);     // row_shr:4
    v += LZ4P_DPP(v, 0x118, 0xf);     // row_shr:8
    v += LZ4P_DPP(v, 0x142, 0xa);     // row_bcast:15 -> rows 1, 3
    v += LZ4P_DPP(v, 0x143, 0xc);     // row_bcast:31 -> rows 2, 3
#undef LZ4P_DPP
    return v;
}

// the compressed bytes: the staged window from LDS, anything else from memory
struct Reader {
    const lds_u8* tile;      // LDS copy of [t0, t0 + PT + PMARGIN)
    const uint8_t* g;
    uint32_t t0;
    __device__ __forceinline__ uint32_t operator()(uint32_t pos) const {
        const uint32_t r = pos - t0;
        return r < PT + PMARGIN ? (uint32_t)tile[tile_at(r)] : (uint32_t)g[pos];
    }
    __device__ __forceinline__ uint32_t u32(uint32_t pos) const { return (*this)(pos) | ((*this)(pos + 1u) << 8) | ((*this)(pos + 2u) << 16) | ((*this)(pos + 3u) << 24); }
};

// a lane's descriptor lists (main: a first walk's; pre: a second walk's, up to where it meets the first)
struct Lists {
    lds_u32* ma; lds_u32* mb; lds_u32* pa; lds_u32* pb;
    __device__ __forceinline__ Desc main(uint32_t i) const { return Desc{ma[i], mb[i]}; }
    __device__ __forceinline__ Desc pre(uint32_t i) const { return Desc{pa[i], pb[i]}; }
    __device__ __forceinline__ void set_main(uint32_t i, const Desc& d) const { ma[i] = d.x; mb[i] = d.y; }
    __device__ __forceinline__ void set_pre(uint32_t i, const Desc& d) const { pa[i] = d.x; pb[i] = d.y; }
};

struct CountSinkD {
    uint32_t n_steps, n_tail;
    __device__ __forceinline__ void step(uint32_t, uint32_t, uint32_t, uint32_t) { n_steps++; }
    __device__ __forceinline__ void tail(uint32_t) { n_tail++; }
};
struct StoreSinkD {
    uint32_t* words;         // the block's plan
    uint32_t* tailw;         // the block's tail records
    uint32_t at;             // next step
    uint32_t n_tail;
    // one 16-byte store per step, STEP-major; the block's layout (lz4_plan_common.h word_of: four steps x four lanes, lane-major)
    // is made by transposing whole groups of four steps afterwards -- four scattered 4-byte stores per step were a memory
    // transaction each, and half of the kernel's time
    __device__ __forceinline__ void step(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3) {
        const u32x4 v = u32x4{w0, w1, w2, w3};
#ifdef LZ4P_EXP_NOSTORE      // timing experiments only
        asm volatile("" :: "v"(v), "v"(words + 4u * at));
#else
        __builtin_memcpy(words + 4u * at, &v, 16);
#endif
        at++;
    }
    __device__ __forceinline__ void tail(uint32_t r) { tailw[n_tail++] = r; }
};

// groups [g0, g1) of four steps: step-major (as written) -> lane-major (as the replay kernel reads them), in place, a lane per group
__device__ __forceinline__ void transpose_groups(uint32_t* words, uint32_t g0, uint32_t g1, uint32_t lane) {
#ifdef LZ4P_EXP_NOTRANSPOSE  // timing experiments only
    return;
#endif
    for (uint32_t g = g0 + lane; g < g1; g += 64u) {
        u32x4 r0, r1, r2, r3;
        uint32_t* p = words + 16u * g;
        // (sc1: served by the L2.  The 128-byte line a group shares with its neighbour may sit in this CU's L1 from the neighbour's
        // transposition a tile ago, without the steps stored since)
        asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %4, off offset:16 sc1\n\t"
                     "global_load_dwordx4 %2, %4, off offset:32 sc1\n\tglobal_load_dwordx4 %3, %4, off offset:48 sc1\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(p) : "memory");
        const u32x4 c0 = u32x4{r0.x, r1.x, r2.x, r3.x}, c1 = u32x4{r0.y, r1.y, r2.y, r3.y}, c2 = u32x4{r0.z, r1.z, r2.z, r3.z},
                    c3 = u32x4{r0.w, r1.w, r2.w, r3.w};
        __builtin_memcpy(p, &c0, 16); __builtin_memcpy(p + 4, &c1, 16); __builtin_memcpy(p + 8, &c2, 16); __builtin_memcpy(p + 12, &c3, 16);
    }
}

struct PartState {
    uint32_t from;       // where the walk whose descriptors stand began (X_ERR: none yet)
    uint32_t exit;       // where the chain leaves the part: a position >= the part's end, X_END or X_ERR
    uint32_t main_exit;  // ... where the main list's chain does (the true chain's exit once it has met the main list)
    uint32_t cnt;        // descriptors in the main list
    uint32_t h;          // the true chain uses main[h ..)
    uint32_t np;         // ... behind pre[0 .. np)
    uint64_t marks;      // token positions of the main list, relative to the part's first byte
    uint32_t big;        // a length the descriptors cannot hold
};

// One walk of a part: from p (a position in the part) until the chain leaves the part.  FIRST: descriptors go to the main list, every
// position is marked.  Else: to the prefix list, until a marked position is reached (the main list stands from there); a walk that
// needs more than PCAP descriptors starts again as a FIRST walk.
template <bool FIRST>
__device__ __forceinline__ void walk_part(const Reader& rd, uint32_t ilen, uint32_t p, uint32_t part0, uint32_t part_end, const Lists& L, PartState& s) {
    const uint32_t entry = p;
    uint32_t n = 0u;
    uint64_t marks = 0ull;
    uint32_t exit_ = X_ERR;
    bool merged = false;
    for (;;) {
        if (p >= part_end) { exit_ = p; break; }
        if (!FIRST) {
            const uint64_t bit = 1ull << (p - part0);
            if (s.marks & bit) {                                  // the first walk passed here: its descriptors stand from this one on
                s.h = (uint32_t)__builtin_popcountll(s.marks & (bit - 1ull));
                merged = true;
                break;
            }
            if (n == PCAP) break;                                 // (no room: walk the part afresh, below)
        }
        pcd::Seq q;
        uint32_t nx;
        // The plain sequence -- literal length in the token or one more byte, match length in the token or one more byte, literals and
        // match bytes inside the staged window, 16 bytes of block left behind the token: everything parse_seq checks holds or is
        // checked here -- costs two LDS round trips: the 8 bytes at the token, the 8 bytes behind the literals, each out of three
        // aligned dwords.  Anything else goes through parse_seq -- rarely: a lane on that path holds up the other 63 (the first
        // version took it for 8 % of the sequences, i.e. in every hop of every wavefront, with four lanes working on average).
        {
            const uint32_t r = p - rd.t0;
            uint32_t ra = r & ~3u;
            uint32_t d0 = *(const lds_u32*)(rd.tile + tile_at(ra)), d1 = *(const lds_u32*)(rd.tile + tile_at(ra + 4u)),
                     d2 = *(const lds_u32*)(rd.tile + tile_at(ra + 8u));
            uint32_t sh = r & 3u;
            const uint32_t lo = __builtin_amdgcn_alignbyte(d1, d0, sh);
            const uint32_t t = lo & 0xFFu, mn = t & 15u;
            const uint32_t e1 = (lo >> 8) & 0xFFu;
            const uint32_t lit = (t >> 4) < 15u ? (t >> 4) : 15u + e1;
            const uint32_t lsrc = p + ((t >> 4) < 15u ? 1u : 2u);
            const uint32_t r1 = lsrc + lit - rd.t0;                       // where the offset is
            bool fast = !((t >> 4) == 15u && e1 == 255u) && r1 + 12u <= PT + PMARGIN && lsrc + lit + 16u <= ilen;
            ra = fast ? (r1 & ~3u) : 0u;
            d0 = *(const lds_u32*)(rd.tile + tile_at(ra)); d1 = *(const lds_u32*)(rd.tile + tile_at(ra + 4u));
            sh = r1 & 3u;
            const uint32_t w1 = __builtin_amdgcn_alignbyte(d1, d0, sh);
            const uint32_t off = w1 & 0xFFFFu, e2 = (w1 >> 16) & 0xFFu;
            fast = fast && off != 0u && !(mn == 15u && e2 == 255u);
            if (fast) {
                q.lit_src = lsrc; q.lit = lit; q.off = off; q.ml = 4u + mn + (mn == 15u ? e2 : 0u);
                nx = lsrc + lit + 2u + (mn == 15u ? 1u : 0u);
            } else {
                nx = pcd::parse_seq(rd, ilen, p, q);
            }
        }
        if (nx == pcd::X_ERR) { exit_ = X_ERR; break; }
        if (n == DCAP) { exit_ = X_ERR; break; }                  // (cannot happen: a sequence with a match is at least 3 bytes)
        if (q.lit > 0xFFFFu || q.ml > 0xFFFFu) s.big = 1u;
        const Desc d = Desc{(p - rd.t0) | (q.lit << 13), q.off | (q.ml << 16)};
        if (FIRST) L.set_main(n, d); else L.set_pre(n, d);
        marks |= 1ull << (p - part0);
        n++;
        if (nx == pcd::X_END) { exit_ = X_END; break; }
        p = nx;
    }
    if (FIRST) {
        s.from = entry; s.exit = exit_; s.main_exit = exit_; s.cnt = n; s.h = 0u; s.np = 0u; s.marks = marks;
    } else if (merged) {
        s.from = entry; s.np = n; s.exit = s.main_exit;           // (an earlier second walk may have left the part elsewhere)
    } else if (exit_ != X_ERR || n < PCAP) {
        // left the part (or failed) without meeting the first walk: the prefix list is the whole chain
        s.from = entry; s.exit = exit_; s.np = n; s.h = s.cnt;
        if (exit_ == X_ERR) { s.np = 0u; }
    } else {
        s.from = X_ERR;                                            // the caller walks the part afresh
    }
}

__device__ __forceinline__ uint32_t lit_src_of(uint32_t tok, uint32_t lit) { return tok + 1u + (lit < 15u ? 0u : 1u + (lit - 15u) / 255u); }

// the emitter over a part's true sequences.  off <= position is the reference's check :398-402.
template <class Sink>
__device__ __forceinline__ void emit_part(const PartState& s, const Lists& L, uint32_t t0, Emit& e, Sink& sink, uint32_t& bad) {
    const uint32_t total = s.np + (s.cnt - s.h);
    for (uint32_t i = 0u; i < total; ++i) {
        const Desc d = i < s.np ? L.pre(i) : L.main(s.h + (i - s.np));
        const uint32_t tok = t0 + (d.x & 0x1FFFu), lit = d.x >> 13, off = d.y & 0xFFFFu, ml = d.y >> 16;
        emit_literals(e, lit_src_of(tok, lit), lit, sink);
        if (ml != 0u) {
            if (off > e.op) { bad = 1u; return; }
            emit_match(e, off, ml, sink);
        }
        if (e.op > MAX_FIELD) { bad = 1u; return; }
    }
    emit_end(e, sink);                                             // a step never holds pieces of two parts
}

__global__ void __launch_bounds__(64) lz4_plan_kernel(PlanArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t plan_lds[];
    lds_u8* lds = (lds_u8*)plan_lds;
    const uint32_t lane = threadIdx.x;
    Lists L;
    L.ma = (lds_u32*)(lds + LDS_MA) + lane * MSTRIDE; L.mb = (lds_u32*)(lds + LDS_MB) + lane * MSTRIDE;
    L.pa = (lds_u32*)(lds + LDS_PA) + lane * PSTRIDE; L.pb = (lds_u32*)(lds + LDS_PB) + lane * PSTRIDE;
    for (uint32_t b = blockIdx.x; b < a.n; b += gridDim.x) {
        const uint8_t* gin = a.in_base + a.in_off[b];
        const uint32_t ilen = a.in_len[b];
        const uint32_t cap = a.out_cap[b];
        uint32_t* words = a.words + (size_t)a.slot_words * b;
        const uint32_t slot_steps = (a.slot_words - MAX_TAIL_SLOT) / G;      // steps the slot holds in front of its tail records
        uint32_t* tailw = words + (size_t)slot_steps * G;
        uint32_t bad = (ilen == 0u || ilen > MAX_FIELD) ? 1u : 0u;            // != 0: why the block has no plan (tools read it from the header's tail_op)            // (an empty block: decompress.rs:207-209, the reference-order kernel reports it)
        uint32_t entry = 0u;         // the next tile's first true token position
        uint32_t OP = 0u;            // decoded bytes in front of it
        uint32_t steps = 0u;         // steps written
        uint32_t gdone = 0u;         // groups of four steps already in their final layout
        uint32_t n_tail = 0u;
        bool ended = false;
        while (!bad && !ended) {
            const uint32_t t0 = entry / PT * PT;                              // (tiles a long literal run jumps over are never staged)
            // ---- stage the tile
            __builtin_amdgcn_s_barrier();                                      // (one wavefront: orders the LDS accesses of the previous turn)
            for (uint32_t o = 16u * lane; o < PT + PMARGIN; o += 16u * 64u) {
                u32x4 v = u32x4{0u, 0u, 0u, 0u};
                if (t0 + o + 16u <= ilen) __builtin_memcpy(&v, gin + t0 + o, 16);
                else {
                    uint32_t w[4] = {0u, 0u, 0u, 0u};
                    for (uint32_t k = 0u; k < 16u; ++k) if (t0 + o + k < ilen) w[k / 4u] |= (uint32_t)gin[t0 + o + k] << (8u * (k % 4u));
                    v = u32x4{w[0], w[1], w[2], w[3]};
                }
                __builtin_memcpy((void*)(lds + LDS_TILE + tile_at(o)), &v, 16);      // (16 bytes never straddle a part)
            }
            __builtin_amdgcn_s_barrier();
            Reader rd;
            rd.tile = lds + LDS_TILE; rd.g = gin; rd.t0 = t0;
            // ---- the parts: lane k owns [part0, part_end)
            const uint32_t part0 = t0 + PB * lane;
            const uint32_t part_end = part0 + PB < ilen ? part0 + PB : ilen;
            const bool has_part = part0 < ilen;
            PartState s;
            s.from = X_ERR; s.exit = X_ERR; s.main_exit = X_ERR; s.cnt = 0u; s.h = 0u; s.np = 0u; s.marks = 0ull; s.big = 0u;
            // ---- 1. first walks: the entry's lane from the tile's entry, the lanes behind it from their part's first byte
            const uint32_t entry_lane = (entry - t0) / PB;
            if (has_part && lane >= entry_lane) walk_part<true>(rd, ilen, lane == entry_lane ? entry : part0, part0, part_end, L, s);
#if defined(LZ4P_EXP_STOP) && LZ4P_EXP_STOP == 1      // timing experiments only: the phases of a tile, cut off one by one
            if (s.exit == X_END) ended = true;
            entry = (uint32_t)__builtin_amdgcn_readlane((int)s.exit, 63);
            if (entry == X_ERR || entry == X_END || t0 + PT >= ilen) ended = true; else entry = t0 + PT;
            continue;
#endif
            // ---- 2. / 3. which parts does the true chain visit, and where does it enter them?  Every part points at the part its
            // exit lies in (a part whose walk began elsewhere is taken to leave where that walk left: true once the chains have met);
            // the parts reachable from the entry's are found by pointer jumping (six rounds of ds_bpermute instead of a scalar walk
            // over up to 64 lanes), each takes the exit of the path part before it as its entry.  Lanes whose walk began elsewhere walk again; until
            // a whole pass finds nothing to do.
            uint32_t my_entry = X_ERR;       // this part's true entry (X_ERR: no sequence of the true chain begins here)
            uint32_t tile_exit = X_ERR;
            for (uint32_t round = 0u; round < NPART + 2u; ++round) {
                const bool inside = s.exit != X_END && s.exit != X_ERR && s.exit < t0 + PT;
                const uint32_t nxt = inside ? (s.exit - t0) / PB : 64u;               // 64: the chain leaves the tile (or ends, or fails) here
                // reach = the parts on the path from this one (itself included), jump = 2^i steps ahead
                uint64_t reach = 1ull << lane;
                uint32_t jump = nxt;
#pragma unroll
                for (uint32_t i = 0u; i < 6u; ++i) {
                    const uint32_t src = (jump < 64u ? jump : lane) * 4u;
                    const uint32_t rlo = (uint32_t)__builtin_amdgcn_ds_bpermute((int)src, (int)(uint32_t)reach);
                    const uint32_t rhi = (uint32_t)__builtin_amdgcn_ds_bpermute((int)src, (int)(uint32_t)(reach >> 32));
                    const uint32_t j2 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)src, (int)jump);
                    if (jump < 64u) { reach |= ((uint64_t)rhi << 32) | rlo; jump = j2; }
                }
                const uint64_t path = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(reach >> 32), (int)entry_lane) << 32) |
                                      (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)reach, (int)entry_lane);
                const bool on_path = ((path >> lane) & 1ull) != 0ull;
                // a part on the path is entered where the path part before it leaves (the chain only moves forward: the path part before
                // it is its predecessor)
                const uint64_t before = path & ((1ull << lane) - 1ull);
                const uint32_t pred = before != 0ull ? 63u - (uint32_t)__builtin_clzll(before) : lane;
                const uint32_t pulled = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(pred * 4u), (int)s.exit);
                my_entry = lane == entry_lane ? entry : (on_path && before != 0ull ? pulled : X_ERR);
                // the last part on the path says where the chain leaves the tile
                const uint32_t last = 63u - (uint32_t)__builtin_clzll(path);
                tile_exit = (uint32_t)__builtin_amdgcn_readlane((int)s.exit, (int)last);
                const bool need = my_entry != X_ERR && s.from != my_entry;
                if (!__any(need)) break;
                if (need) {
                    if (s.from != X_ERR && s.cnt != 0u) walk_part<false>(rd, ilen, my_entry, part0, part_end, L, s);
                    else s.from = X_ERR;
                    if (s.from == X_ERR) walk_part<true>(rd, ilen, my_entry, part0, part_end, L, s);
                }
                tile_exit = X_ERR;                                           // (not final: the next pass says)
            }
            // the chain must have come through: tile_exit is a position behind the tile, or the block's end
            if (tile_exit == X_ERR) { bad = 2u; break; }
            const bool live = my_entry != X_ERR;
            if (__any(live && s.big != 0u)) { bad = 3u; break; }
#if defined(LZ4P_EXP_STOP) && LZ4P_EXP_STOP == 2
            if (tile_exit == X_END) ended = true; else entry = tile_exit;
            continue;
#endif
            // ---- 4. output positions, step counts, steps
            uint32_t U = 0u;
            if (live) {
                const uint32_t total = s.np + (s.cnt - s.h);
                for (uint32_t i = 0u; i < total; ++i) {
                    const Desc d = i < s.np ? L.pre(i) : L.main(s.h + (i - s.np));
                    U += (d.x >> 13) + (d.y >> 16);
                }
            }
            const uint32_t incl = wave_incl_add(U);
            const uint32_t tileU = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            if ((uint64_t)OP + tileU > cap || (uint64_t)OP + tileU > MAX_FIELD) { bad = 4u; break; }   // (OutputTooSmall: the reference-order kernel names it)
            const uint32_t my_op = OP + incl - U;
            // the counting run.  A lane whose literals reach the block's last bytes opens the tail; the lanes behind it (and the
            // tiles behind this one) run in tail mode from their first piece: counted again
            Emit e;
            uint32_t lane_bad = 0u;
            CountSinkD cs; cs.n_steps = 0u; cs.n_tail = 0u;
            uint32_t tail_in = n_tail != 0u ? 1u : 0u;
            if (live) {
                emit_init(e, ilen); e.op = my_op; e.tail = tail_in;
                emit_part(s, L, t0, e, cs, lane_bad);
            }
            if (tail_in == 0u) {
                const uint64_t tm = __builtin_amdgcn_ballot_w64(live && cs.n_tail != 0u);
                if (tm != 0ull) {
                    const uint32_t opener = (uint32_t)__builtin_ctzll(tm);
                    if (live && lane > opener) {
                        tail_in = 1u;
                        cs.n_steps = 0u; cs.n_tail = 0u;
                        emit_init(e, ilen); e.op = my_op; e.tail = 1u;
                        emit_part(s, L, t0, e, cs, lane_bad);
                    }
                }
            }
            if (__any(lane_bad != 0u)) { bad = 5u; break; }
#if defined(LZ4P_EXP_STOP) && LZ4P_EXP_STOP == 3
            steps += cs.n_steps & 1u;
            if (tile_exit == X_END) ended = true; else entry = tile_exit;
            continue;
#endif
            const uint32_t sincl = wave_incl_add(cs.n_steps);
            const uint32_t tile_steps = (uint32_t)__builtin_amdgcn_readlane((int)sincl, 63);
            const uint32_t tincl = wave_incl_add(cs.n_tail);
            const uint32_t tile_tail = (uint32_t)__builtin_amdgcn_readlane((int)tincl, 63);
            if (steps + tile_steps + 1u + (END_TURNS + 1u) * TURN_STEPS > slot_steps || n_tail + tile_tail > MAX_TAIL_SLOT) { bad = 6u; break; }
            // the writing run
            if (live) {
                StoreSinkD ss; ss.words = words; ss.tailw = tailw; ss.at = steps + sincl - cs.n_steps; ss.n_tail = n_tail + tincl - cs.n_tail;
                emit_init(e, ilen); e.op = my_op; e.tail = tail_in;
                emit_part(s, L, t0, e, ss, lane_bad);
            }
            steps += tile_steps;
            n_tail += tile_tail;
            OP += tileU;
            // the groups of four steps that are complete now get their final layout (the lanes' stores first: other lanes read them)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            transpose_groups(words, gdone, steps / 4u, lane);
            gdone = steps / 4u;
            if (tile_exit == X_END) ended = true; else entry = tile_exit;
        }
        // ---- the block's end: K_END up to whole turns + END_TURNS, header, verdict
        const uint32_t turns = (steps + 1u + TURN_STEPS - 1u) / TURN_STEPS + END_TURNS;
        if (!bad) {
            // K_END: every word of these steps is the same, the layout does not matter; the group the last real step shares with them does
            const u32x4 endv = u32x4{END_REC, END_REC, END_REC, END_REC};
            for (uint32_t st = steps + lane; st < turns * TURN_STEPS; st += 64u) __builtin_memcpy(words + 4u * st, &endv, 16);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            transpose_groups(words, gdone, (steps + 3u) / 4u, lane);
        }
        if (lane == 0u) {
            // the bytes the steps move: everything but the tail's
            uint32_t tail_bytes = 0u;
            if (!bad) for (uint32_t t = 0u; t < n_tail; ++t) tail_bytes += rec_n(tailw[t]);
            BlockPlan bp;
            bp.in_off = a.in_off[b]; bp.out_off = a.out_off[b];
            bp.first_word = 0u; bp.tail_word = 0u; bp.tail_op = bad; bp.n_tail = 0u; bp.flags = 1u;
            if (!bad) {
                const uint64_t w0 = (uint64_t)a.slot_words * b;
                bp.first_word = (uint32_t)w0;
                bp.tail_word = (uint32_t)(w0 + (uint64_t)slot_steps * G);
                bp.tail_op = OP - tail_bytes;
                bp.n_tail = (uint16_t)n_tail;
                bp.flags = 0u;
            }
            a.plans[b] = bp;
            a.status[b] = bad ? a.redo_code : 0;
            a.out_len[b] = bad ? 0u : OP;
        }
    }
}

}  // namespace plan

size_t plan_slot_words() { return plan::PLAN_SLOT_WORDS; }

hipError_t launch_plan(const PlanArgs& a, hipStream_t s) {
    if (a.n == 0u) return hipSuccess;
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const uint32_t per_cu = 160u * 1024u / plan::LDS_BYTES;
    uint32_t grid = (uint32_t)cus * (per_cu > 8u ? 8u : per_cu);
    if (grid > a.n) grid = a.n;
    hipLaunchKernelGGL(plan::lz4_plan_kernel, dim3(grid), dim3(64), plan::LDS_BYTES, s, a);
    return hipGetLastError();
}

}  // namespace lz4flex_dev

// tools / tests: the plan kernel alone (the plans are then replayed by the host model, whose guards say what is wrong with them)
extern "C" int lz4flex_debug_plan(const void* in_base, const void* in_off, const void* in_len, const void* out_off, const void* out_cap, unsigned n,
                                  void* plans, void* words, void* out_len, void* status, void* stream) {
    lz4flex_dev::PlanArgs a;
    a.in_base = (const uint8_t*)in_base; a.in_off = (const uint64_t*)in_off; a.in_len = (const uint32_t*)in_len;
    a.out_off = (const uint64_t*)out_off; a.out_cap = (const uint32_t*)out_cap;
    a.plans = (lz4flex_dev::plan::BlockPlan*)plans; a.words = (uint32_t*)words; a.out_len = (uint32_t*)out_len; a.status = (int32_t*)status;
    a.n = n; a.slot_words = (uint32_t)lz4flex_dev::plan_slot_words(); a.redo_code = 0x7F000001;
    return (int)lz4flex_dev::launch_plan(a, (hipStream_t)stream);
}
extern "C" unsigned lz4flex_debug_plan_slot_words() { return (unsigned)lz4flex_dev::plan_slot_words(); }
#endif  // LZ4FLEX_TOOLS
