// lz4_decompress_split.hip -- batched LZ4 block decoder, PARSER / COPIER split ("v5").
//
// Same contract as lz4_decompress.hip (reference src/block/decompress.rs:201-449: result bytes, byte count,
// error variant and OutputTooSmall{expected,actual}, unsafe-flavour check order), blocks without dictionary /
// prefix.  What round 1's pipelined decoder (8 lanes per block, deleted) spent its issue slots on was the token parse:
// all 8 lanes ran the same parse (116 VALU + 75 SALU per step) to produce ONE sequence.  Here the two halves of the
// reference loop run in different wavefronts of a workgroup:
//
//   * PARSER wavefront: ONE LANE PER BLOCK (up to 64 blocks).  A lane walks its block's token chain
//     (token, literal length, offset, match length: decompress.rs:244-332,377-391) with every bounds check of
//     the reference, and pushes {literal source, literal length, match length, offset} records into the
//     block's LDS queue.  It never touches the output.  The compressed bytes around the parse position sit in a 64-byte
//     LDS RING per block (16-byte chunks enter it from a register loaded one step earlier; round 1 kept a 48-byte register
//     window, whose sliding and per-lane byte selection was half of a step's instructions); the last 48 bytes of a block
//     are staged in LDS (zero padded) so that nothing is read behind the block.
//     Everything that is not a plain sequence (255-chains, errors, the block's last sequence) goes through an
//     exact byte-wise path that follows decompress.rs line by line.
//   * COPIER wavefronts: a group of G lanes per block (large batches: 4 lanes x 16 bytes, four wavefronts for 64 blocks;
//     smaller ones: 8 lanes x 4 bytes).  A group pops records and executes them as pieces of up to G x WB bytes on the LDS
//     block's LDS output buffer (512 B of history, 16 B/lane coalesced write-back): literal pieces and far
//     match pieces are global loads issued three steps before their bytes are needed, near matches are LDS -> LDS.  Every
//     4 steps a group snapshots the queue's tail and publishes its head; once per iteration it checks the buffer space for
//     the iteration's pieces and touches the next 128-byte line of the compressed stream ahead of the parser, which keeps
//     the parser's chunk loads out of HBM latency.
//
// The parser is per-lane scalar code: lz4_split_parser.h also compiles for the host (tests/sim/), where it is checked
// against the oracle.  Measurements, what bounds the kernel and the variants that were tried and dropped: DESIGN.md
// section 5.2.
//
// LDS per block (LayoutBig): 1 984 B output buffer + 16 x 16 B queue + 16 B head/tail + 80 B tail copy + 16 B sink, then
// (64 B aligned) the 80 B ring = 2 496 B; 64 blocks = 156 KiB of the CU's 160 KiB.  LayoutSmall (1 024 B buffer, 8
// records) only exists for the host simulation of a short queue.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lz4_device.h"
#include "lz4_split_parser.h"

namespace lz4flex_dev {
#ifdef LZ4FLEX_PROFILE_PHASES
// [0] parser wave cycles, [1] parser wave-steps, [2..6] lane sums: live steps, queue-full, window bubbles, exact path, records
// [8] copier wave cycles, [9] copier wave 4-step iterations, [10] service visits, [11] group-steps with a piece,
// [12] group-steps idle on an empty queue, [13] group-steps blocked/done, [14] cycles in service
__device__ unsigned long long g_split_prof[16];
#define SP_ADD(k, v) atomicAdd(&g_split_prof[k], (unsigned long long)(v))
#endif
namespace v5 {

// =====================================================================================================
// COPIER: G lanes = one block
// =====================================================================================================
template <uint32_t WB> struct Word;
template <> struct Word<4> { using type = uint32_t; };
template <> struct Word<16> { using type = u32x4; };

// G lanes per block, WB bytes per lane and piece, NS pieces in flight (a piece's global load is issued NS - 1 steps before its
// bytes are stored); L = the block's LDS layout
template <class L, uint32_t G, uint32_t WB, uint32_t NS>
struct Copier {
    static constexpr uint32_t PIECE = G * WB;
    static_assert((WB == 4u || WB == 16u) && NS >= 2u && NS * PIECE + 64u + 64u <= L::FLUSH_AT && PIECE <= L::OUT_H, "geometry");
    static constexpr uint32_t OUT_CAP = L::OUT_CAP, OUT_H = L::OUT_H, FLUSH_AT = L::FLUSH_AT, TAIL_OFF = L::TAIL_OFF;
    using word_t = typename Word<WB>::type;
    const uint8_t* gin;
    uint8_t* gout;
    lds_u8* lout;         // the block's LDS output buffer
    QueueT<L> q;
    uint32_t g;
    uint32_t ilen;
    uint32_t op, L0, F;   // LDS output holds positions [L0, op); [0, F) is written back
    uint32_t lit_src, lit_rem, ml_rem, moff;
    const uint8_t* gin_ld;   // bases of the per-step load: gin / gout, or g_pad when the block is too small to read 4 bytes from
    const uint8_t* gout_ld;
    uint32_t ilen_w;         // ilen - WB (0 with gin_ld = g_pad): the last position a literal load may start at
    uint32_t head, head_pub, snap;
    u32x4 e;                 // the record at `head` (complete iff head != snap)
    uint32_t pf_next;        // next compressed position whose line has not been touched yet
    uint32_t pf_v, pf_acc;
    uint32_t blocked, done;
#ifdef LZ4FLEX_PROFILE_PHASES
    uint32_t pr_piece, pr_idle, pr_blocked;
#endif
    enum : uint32_t { K_NONE = 0, K_MAINT = 1, K_RARE = R_RARE, K_CAREFUL = R_CAREFUL, K_FINISH = R_FINISH, K_LONG = 8 };
    static constexpr uint32_t LONG_LIT = 1024u, LONG_KEEP = 256u;     // literal runs from LONG_LIT bytes on: all but the last LONG_KEEP (+ < 64) bytes go memory to memory
    static_assert(OUT_H + 16u + 64u <= LONG_LIT - LONG_KEEP - 64u, "the window is rebuilt from the run itself");
    struct Slot { uint32_t n, dst, msrc, glob; word_t v; };
    static __device__ __forceinline__ word_t ldw(const uint8_t* p) { word_t v; __builtin_memcpy(&v, p, WB); return v; }

    __device__ __forceinline__ uint32_t out_space() const { return L0 + OUT_CAP - OUT_SLACK - op; }
    __device__ __forceinline__ void st32l(uint32_t off, uint32_t v) const { __builtin_memcpy((void*)(lout + off), &v, 4); }
    __device__ __forceinline__ uint32_t ld32o(uint32_t off) const { return ld32l(lout + off); }

    __device__ __forceinline__ void flush_slide() {
        const uint32_t fnew = op & ~15u;
        for (uint32_t p = F + 16u * g; p < fnew; p += 16u * G) {
            const u32x4 v = *reinterpret_cast<const u32x4 __attribute__((address_space(3)))*>(lout + (p - L0));
            __builtin_memcpy(gout + p, &v, 16);
        }
        F = fnew;
        const uint32_t new_l0 = F > OUT_H ? F - OUT_H : 0u;   // multiple of 16
        if (new_l0 > L0) {
            const uint32_t shift = new_l0 - L0;
            const uint32_t keep = op - new_l0;
            for (uint32_t i = 16u * g; i < keep; i += 16u * G) {
                const u32x4 v = *reinterpret_cast<const u32x4 __attribute__((address_space(3)))*>(lout + shift + i);
                *reinterpret_cast<u32x4 __attribute__((address_space(3)))*>(lout + i) = v;
            }
            L0 = new_l0;
        }
    }
    __device__ __forceinline__ void final_flush() {
        const uint32_t fnew = op & ~15u;
        for (uint32_t p = F + 16u * g; p < fnew; p += 16u * G) {
            const u32x4 v = *reinterpret_cast<const u32x4 __attribute__((address_space(3)))*>(lout + (p - L0));
            __builtin_memcpy(gout + p, &v, 16);
        }
        for (uint32_t p = fnew + g; p < op; p += G) gout[p] = lout[p - L0];
        F = op;
    }
    // A long literal run (incompressible data: a 64 KiB block is ONE run; round 3: 3.2 ms per GiB through 64-byte pieces and the LDS
    // buffer) goes memory to memory, 16 bytes per lane and four loads in flight, up to LONG_KEEP bytes before its end; the rest -- the end
    // of the block may be there -- goes the usual way.  The LDS buffer is written back first and rebuilt behind the bulk from the
    // run's own bytes (512 bytes of history + the granule that is not complete), with the invariants flush_slide() leaves.
    // The copy is a function of its own (not inlined: its registers would be the kernel's -- inlined, the hot loop spilled and
    // JSON blocks took 1.74 instead of 1.64 ms).
    // The WHOLE WAVEFRONT copies one block's run at a time, 16 bytes per lane and load, four loads in flight (the wavefront's other
    // blocks cannot step while one of them is served anyway: with only the block's own four lanes copying, a block with several
    // long runs held its fifteen neighbours up for 30 us each time).  src / dst / n are wave-uniform.
    static __device__ __attribute__((noinline)) void wave_bulk_copy(const uint8_t* src, uint8_t* dst, uint32_t n_, uint32_t lane) {
        typedef __attribute__((address_space(1))) uint8_t g_u8;      // (named: a pointer that crossed a call is flat)
        typedef u32x4 __attribute__((aligned(1))) u32x4_u;
        typedef __attribute__((address_space(1))) u32x4_u g_u32x4_u;
        const uint64_t sa = ((uint64_t)__builtin_amdgcn_readfirstlane((int)((uint64_t)src >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uint64_t)src);
        const uint64_t da = ((uint64_t)__builtin_amdgcn_readfirstlane((int)((uint64_t)dst >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uint64_t)dst);
        const uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane((int)n_);
        const g_u8* s1 = (const g_u8*)sa;
        g_u8* d1 = (g_u8*)da;
        auto ld = [&](uint32_t o) -> u32x4 { return *reinterpret_cast<const g_u32x4_u*>(s1 + o); };
        auto st = [&](uint32_t o, const u32x4& v) { *reinterpret_cast<g_u32x4_u*>(d1 + o) = v; };
        uint32_t i = 16u * lane;
        for (; i + 3072u < n; i += 4096u) {
            const u32x4 v0 = ld(i), v1 = ld(i + 1024u), v2 = ld(i + 2048u), v3 = ld(i + 3072u);
            st(i, v0); st(i + 1024u, v1); st(i + 2048u, v2); st(i + 3072u, v3);
        }
        for (; i < n; i += 1024u) st(i, ld(i));
    }
    // ... and a block's LDS buffer behind its run: 512 bytes of history + the granule that is not complete, from the run's own bytes
    static __device__ __attribute__((noinline)) void bulk_rebuild(lds_u8* lds_dst, const uint8_t* hist, uint32_t have, uint32_t g) {
        typedef __attribute__((address_space(1))) uint8_t g_u8;
        typedef u32x4 __attribute__((aligned(1))) u32x4_u;
        typedef __attribute__((address_space(1))) u32x4_u g_u32x4_u;
        const g_u8* h1 = (const g_u8*)hist;
        for (uint32_t k = 16u * g; k < have; k += 16u * G)          // (the last granule reads a few of the run's remaining bytes: there are LONG_KEEP of them)
            *reinterpret_cast<u32x4 __attribute__((address_space(3)))*>(lds_dst + k) = *reinterpret_cast<const g_u32x4_u*>(h1 + k);
    }
    // every lane of the wavefront calls this (want: this lane's block has a run to move)
    __device__ void bulk_literals(bool want) {
        if (want) final_flush();
        const uint32_t n = want ? (lit_rem - LONG_KEEP) & ~63u : 0u;
        const uint64_t sp = (uint64_t)(gin + lit_src), dp = (uint64_t)(gout + op);
        const uint32_t lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        uint64_t m = __ballot(want && g == 0u);
        while (m != 0ull) {
            const uint32_t l = (uint32_t)__builtin_ctzll(m);
            m &= m - 1ull;
            const uint64_t s = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(sp >> 32), (int)l) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)sp, (int)l);
            const uint64_t d = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(dp >> 32), (int)l) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)dp, (int)l);
            wave_bulk_copy((const uint8_t*)s, (uint8_t*)d, (uint32_t)__builtin_amdgcn_readlane((int)n, (int)l), lane);
        }
        if (want) {
            const uint32_t op2 = op + n, f2 = op2 & ~15u, l2 = f2 > OUT_H ? f2 - OUT_H : 0u;
            const uint32_t have = op2 - l2;                         // <= OUT_H + 15 <= n: every byte of it is a byte of this run
            bulk_rebuild(lout, gin + lit_src + n - have, have, g);
            op = op2; lit_src += n; lit_rem -= n;
            F = f2; L0 = l2;
        }
    }
    // literals with exact source bounds (the block's last literals end at its last byte)
    __device__ void generic_literals(uint32_t s, uint32_t n) {
        while (n != 0u) {
            uint32_t space = out_space();
            if (space < 64u && space < n) { flush_slide(); space = out_space(); }
            const uint32_t m = n < space ? n : space;
            const uint32_t dst = op - L0;
            for (uint32_t i = 4u * g; i < m; i += 4u * G) {
                if (s + i + 4u <= ilen) {
                    st32l(dst + i, ld32(gin + s + i));
                } else {
                    for (uint32_t k = i; k < m; ++k) lout[dst + k] = gin[s + k];
                }
            }
            s += m; op += m; n -= m;
        }
    }
    // any offset (periodic copies included), any length: decompress.rs:410-437
    __device__ void generic_match(uint32_t offset, uint32_t n) {
        while (n != 0u) {
            uint32_t space = out_space();
            if (space < 64u && space < n) { flush_slide(); space = out_space(); }
            const uint32_t m = n < space ? n : space;
            const uint32_t dst = op - L0;
            const uint32_t src = op - offset;
            if (offset >= 4u * G) {
                for (uint32_t i = 4u * g; i < m; i += 4u * G) {
                    const uint32_t p = src + i;
                    const uint32_t v = (p >= L0) ? ld32o(p - L0) : ld32(gout + p);   // older bytes are written back: p + 4 <= L0 + 3 < F
                    st32l(dst + i, v);
                }
            } else if (offset == 1u) {
                const uint32_t v = (uint32_t)lout[dst - 1u] * 0x01010101u;
                for (uint32_t i = 4u * g; i < m; i += 4u * G) st32l(dst + i, v);
            } else {
                // periodic: the `offset` bytes before op are in LDS (offset < 32 <= op - L0 or L0 == 0)
                const lds_u8* pat = lout + dst - offset;
                uint32_t idx = (4u * g) % offset;
                const uint32_t stp = (4u * G) % offset;
                for (uint32_t i = 4u * g; i < m; i += 4u * G) {
                    uint32_t j = idx, v = 0u;
#pragma unroll
                    for (uint32_t k = 0; k < 4u; ++k) {
                        v |= (uint32_t)pat[j] << (8u * k);
                        j = (j + 1u == offset) ? 0u : j + 1u;
                    }
                    st32l(dst + i, v);
                    idx += stp;
                    if (idx >= offset) idx -= offset;
                }
            }
            op += m; n -= m;
        }
    }

    // Every 4 steps: snapshot the queue's tail (records below it are complete), publish the head.
    __device__ __forceinline__ void snapshot() {
        snap = q.tail();
        e = q.get(head);   // re-read AFTER the snapshot: the copy fetched at the end of the last step may predate the record
        if (g == 0u && head != head_pub) q.set_head(head);
        head_pub = head;
    }
    // Once per NS-step iteration: make sure NS pieces fit into the output buffer, touch the next line of the compressed
    // stream ahead of the parser.
    __device__ __forceinline__ void iteration_begin() {
        // (a long literal run is noticed HERE, once per iteration, some pieces after its record was popped: the same test at the pop,
        // one compare and one select per step, made JSON blocks 6 % slower -- 1.72 instead of 1.62 ms)
#ifdef LZ4S_EXP_NOLONG    // (timing experiments: round 3's line)
        if ((done | blocked) == 0u && out_space() < NS * PIECE + 64u) blocked = K_MAINT;
#else
        if ((done | blocked) == 0u) blocked = lit_rem >= LONG_LIT ? (uint32_t)K_LONG : (out_space() < NS * PIECE + 64u ? (uint32_t)K_MAINT : (uint32_t)K_NONE);
#endif
        pf_acc += pf_v;                                   // the touch issued one iteration ago (long complete)
        const bool pf = pf_next < lit_src + PF_AHEAD;
        const uint32_t pos = pf_next + (128u / G) * g;
        pf_v = ld32(gin_ld + (pos < ilen_w ? pos : ilen_w));
        pf_next = pf ? pf_next + 128u : pf_next;
    }
    // Front end: pop a record when the current one is finished, cut one piece, issue its global load.
    __device__ __forceinline__ void fe_step(Slot& s) {
        const bool ok = (done | blocked) == 0u;
        const bool pop = ok && (lit_rem | ml_rem) == 0u && head != snap;
        lit_src = pop ? e.x : lit_src;
        lit_rem = pop ? e.y : lit_rem;
        ml_rem = pop ? e.z : ml_rem;
        moff = pop ? (e.w & 0xFFFFu) : moff;
        const uint32_t kind = pop ? e.w >> 16 : 0u;      // R_RARE / R_CAREFUL / R_FINISH == K_*: generic code in service()
        blocked |= kind;
        head = pop ? head + 1u : head;
        e = q.get(head);                                  // next record (complete iff head != snap); its latency overlaps this step
        const bool go = ok && kind == 0u;
        const bool isl = lit_rem != 0u;
        // a match piece never reads bytes of its own: whole words up to the offset, or (words wider than 4 bytes) the offset's
        // bytes in the group's first lane.  Offsets below 4 are R_RARE records.
        const uint32_t pm = moff >= PIECE ? PIECE : ((WB == 4u || moff >= WB) ? (moff & ~(WB - 1u)) : moff);
        const uint32_t rem = isl ? lit_rem : ml_rem;
        const uint32_t lim = isl ? PIECE : pm;
        const uint32_t n = go ? (rem < lim ? rem : lim) : 0u;
        const uint32_t msrc = op - moff;
#ifdef LZ4S_EXP_NOFAR    // timing experiments only (tools; the output is wrong): far matches read LDS like near ones
        const bool far = false;
#else
        const bool far = !isl && msrc < L0;
#endif
        // one load per step and lane: literal bytes, far match bytes (already written back: msrc + PIECE <= L0 + PIECE - 1 < F),
        // or nothing useful (position 0; a load only where a piece needs one: 1.94 instead of 1.65 ms, hipcc waits for a
        // conditional load where it is issued).  Literal positions are clamped so that no lane reads behind the block.
        const uint32_t lpos = lit_src + WB * g;
        const uint32_t off = isl ? (lpos < ilen_w ? lpos : ilen_w) : (far ? msrc + WB * g : 0u);
#ifdef LZ4S_EXP_NOGLOB   // ... no global load at all in a step
        s.v = word_t{}; asm volatile("" :: "v"(off));
#else
        s.v = ldw((isl ? gin_ld : gout_ld) + off);
#endif
        s.n = n;
        s.dst = op - L0;
#ifdef LZ4S_EXP_NOFAR
        s.msrc = (msrc - L0) & 1023u;
#else
        s.msrc = far ? 0u : msrc - L0;                    // LDS source (always read; 0 when unused)
#endif
        s.glob = (isl || far) ? 1u : 0u;
        lit_src = isl ? lit_src + n : lit_src;
        lit_rem = isl ? lit_rem - n : lit_rem;
        ml_rem = isl ? ml_rem : ml_rem - n;
        op += n;
#ifdef LZ4FLEX_PROFILE_PHASES
        pr_piece += n != 0u; pr_idle += ok && !pop && n == 0u; pr_blocked += !ok;
#endif
    }
    __device__ __forceinline__ void be_step(const Slot& s) {
#ifdef LZ4S_EXP_NOBE     // ... no LDS traffic in the back end
        asm volatile("" :: "v"(s.n), "v"(s.dst), "v"(s.msrc), "v"(s.glob), "v"(s.v));
        return;
#endif
        word_t x;
        __builtin_memcpy(&x, (const void*)(lout + (s.glob ? 0u : s.msrc) + WB * g), WB);
        if (s.glob) x = s.v;
        if (WB * g < s.n) __builtin_memcpy((void*)(lout + s.dst + WB * g), &x, WB);
    }
    __device__ __forceinline__ void service() {
        {
            const bool want = done == 0u && lit_rem >= LONG_LIT && (blocked == K_LONG || blocked == K_CAREFUL || blocked == K_FINISH);
            if (__any(want)) bulk_literals(want);              // (the whole wavefront: see wave_bulk_copy)
        }
        if (done) return;
        if (out_space() < FLUSH_AT) flush_slide();
        if (blocked == K_RARE) {   // a periodic match; the record's literals (if any) go first
            generic_literals(lit_src, lit_rem);
            lit_rem = 0u;
            generic_match(moff, ml_rem);
            ml_rem = 0u;
        } else if (blocked == K_CAREFUL || blocked == K_FINISH) {
            generic_literals(lit_src, lit_rem);
            lit_rem = 0u;
            if (blocked == K_FINISH) { final_flush(); done = 1u; }
        }
        blocked = K_NONE;
    }
    __device__ __forceinline__ void run() {
        Slot sl[NS];
#ifdef LZ4FLEX_PROFILE_PHASES
        pr_piece = pr_idle = pr_blocked = 0u;
        const unsigned long long t_begin = __builtin_readcyclecounter();
        unsigned long long t_service = 0ull;
        uint32_t iters = 0u, services = 0u;
#endif
        if (__all(done != 0u)) return;   // a wavefront without blocks (batch tail)
        for (;;) {
#pragma unroll
            for (uint32_t k = 1u; k < NS; ++k) { sl[k].n = 0u; sl[k].dst = 0u; sl[k].msrc = 0u; sl[k].glob = 0u; sl[k].v = word_t{}; }
            do {
                const uint32_t before = head + op;
#pragma unroll
                for (uint32_t k = 0u; k < NS; ++k) {
                    if (k % 4u == 0u) snapshot();
                    if (k == 0u) iteration_begin();
                    fe_step(sl[k]); be_step(sl[(k + 1u) % NS]);
                }
                if (!__any(head + op != before)) __builtin_amdgcn_s_sleep(2);   // nothing to do in the whole wave: yield issue slots
#ifdef LZ4FLEX_PROFILE_PHASES
                iters++;
#endif
            } while (!__any(blocked != K_NONE));
#pragma unroll
            for (uint32_t k = 1u; k < NS; ++k) be_step(sl[k]);
#ifdef LZ4FLEX_PROFILE_PHASES
            const unsigned long long ts = __builtin_readcyclecounter();
#endif
            service();
#ifdef LZ4FLEX_PROFILE_PHASES
            t_service += __builtin_readcyclecounter() - ts;
            services++;
#endif
            if (__all(done != 0u)) break;
        }
        if (g == 0u && head != head_pub) q.set_head(head);
        if (pf_acc + pf_v == 0x9E3779B9u && ilen == 0xFFFFFFFFu) gout[0] = 1;   // keeps the touch loads alive; never true
#ifdef LZ4FLEX_PROFILE_PHASES
        if (g == 0u) { SP_ADD(11, pr_piece); SP_ADD(12, pr_idle); SP_ADD(13, pr_blocked); }
        if (threadIdx.x % 64u == 0u) {
            SP_ADD(8, __builtin_readcyclecounter() - t_begin); SP_ADD(9, iters); SP_ADD(10, services); SP_ADD(14, t_service);
        }
#endif
    }
};

// NB blocks per workgroup: NB*G/64 copier wavefronts and the parser wavefront (NB lanes in use).  G copier lanes per
// block move WB bytes each per piece, NS pieces are in flight.
// Measured on the configs[1] workload (tools/dec_variants.py, 16 384 blocks, same run, ms): 8 x 4 B 1.75; 8 x 16 B 1.83;
// 4 x 16 B 1.65 (NS = 3 / 4 / 6 / 8: 1.69 / 1.65 / 1.70 / 1.69); 2 x 16 B 1.71.  A 16-byte word costs the instructions of
// a 4-byte one, the pieces are twice as long (13 % fewer of them: most are cut by the sequence, not by the piece size) and
// half the copier wavefronts issue them.  8-byte words (8 x 8 B 4.0, 4 x 8 B 4.4) are not an option: an unaligned 8-byte
// LDS access is several times slower than a 4- or a 16-byte one.  Two or four blocks per parser lane (independent chains
// in one instruction stream) 2.94 / 14.4 ms (round 1): the compiler serialises them.
// Wavefronts of a workgroup are dealt to the four SIMDs in turn, and the parser -- the serial chain everything waits for --
// gets SIMD 3 to itself: with eight copier wavefronts (8 x 4 B; ISO) the workgroup is launched with twelve, wavefront 3 is
// the parser and 7, 10, 11 end at once; with four (4 x 16 B) it is launched with eight, the copiers are wavefronts 0, 1, 2, 4
// and 5, 6, 7 end at once; with two, wavefronts 0, 1 of four.
#ifndef LZ4S_ISO_WAVES
#define LZ4S_ISO_WAVES 12
#endif
template <uint32_t CW, uint32_t G> constexpr bool split_iso() { return G == 8u && CW == 8u && LZ4S_ISO_WAVES != 0; }
template <uint32_t CW, uint32_t G> constexpr uint32_t split_waves() {
    if (G != 8u) return CW == 4u ? 8u : (CW == 2u ? 4u : CW + 1u);
    return split_iso<CW, G>() ? (uint32_t)LZ4S_ISO_WAVES : CW + 1u;
}

template <class L, uint32_t NB, uint32_t G, uint32_t WB, uint32_t NS>
__global__ void __launch_bounds__((64 * split_waves<NB * G / 64, G>())) lz4_decompress_split_kernel(DecompressArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t dyn_lds[];
    lds_u8* lds = (lds_u8*)dyn_lds;
    constexpr uint32_t CW = NB * G / 64u;
    static_assert(NB <= 64 && (NB * G) % 64 == 0, "geometry");
    const uint32_t pw = threadIdx.x / 64u;
    uint32_t wave = pw;                                    // role: < CW copier, == CW parser
    // (the padding wavefronts only exist to place the others on the SIMDs: they take part in the workgroup's one barrier -- a
    // wavefront that ends before a barrier the others wait at is outside what HIP defines -- and end right behind it)
    if (G != 8u) {
        if (CW == 4u) {
            if (pw >= 5u) { __syncthreads(); return; }
            wave = pw == 3u ? CW : (pw < 3u ? pw : 3u);
        } else if (CW == 2u) {
            if (pw == 2u) { __syncthreads(); return; }
            wave = pw == 3u ? CW : pw;
        }
    } else if (split_iso<CW, G>()) {
#if LZ4S_ISO_WAVES == 12
        if (pw == 7u || pw >= 10u) { __syncthreads(); return; }
        wave = pw == 3u ? CW : (pw < 3u ? pw : (pw < 7u ? pw - 1u : pw - 2u));
#elif LZ4S_ISO_WAVES == 9      // experiment: the parser shares SIMD 3 with one copier
        wave = pw == 3u ? CW : (pw < 3u ? pw : pw - 1u);
#elif LZ4S_ISO_WAVES == 10     // experiment: parser + wavefront 7 idle: SIMDs 0, 1, 2 carry 3, 3, 2 copiers (as 12) 
        if (pw == 7u) { __syncthreads(); return; }
        wave = pw == 3u ? CW : (pw < 3u ? pw : (pw < 7u ? pw - 1u : pw - 2u));
#endif
    }
    const uint32_t lane = threadIdx.x % 64u;
    const uint32_t first = blockIdx.x * NB;
    if (wave < CW) {
        // ---- copier group: set up the block's queue and tail copy, then run
        const uint32_t j = wave * (64u / G) + lane / G;
        const uint32_t b = first + j;
        const bool valid = b < a.n;
        Copier<L, G, WB, NS> c;
        constexpr uint32_t BLK_LDS = L::BLK_LDS, TAIL_OFF = L::TAIL_OFF;
        c.g = lane % G;
        c.lout = lds + j * BLK_LDS;
        c.q.blk = c.lout;
        c.gin = valid ? a.in_base + a.in_off[b] : g_pad;
        c.gout = valid ? a.out_base + a.out_off[b] : nullptr;
        c.ilen = valid ? a.in_len[b] : 0u;
        const uint32_t tstart = c.ilen > TAILB ? c.ilen - TAILB : 0u;
        for (uint32_t i = c.g; i < TAIL_BUF; i += G) c.lout[TAIL_OFF + i] = (tstart + i < c.ilen) ? c.gin[tstart + i] : (uint8_t)0;
        if (c.g == 0u) { c.q.set_head(0u); c.q.set_tail(0u); }
        c.op = 0u; c.L0 = 0u; c.F = 0u;
        c.lit_src = 0u; c.lit_rem = 0u; c.ml_rem = 0u; c.moff = 0u; c.head = 0u; c.head_pub = 0u; c.snap = 0u; c.pf_next = 0u;
        c.pf_v = 0u; c.pf_acc = 0u;
        c.gin_ld = c.ilen >= WB ? c.gin : g_pad;
        c.ilen_w = c.ilen >= WB ? c.ilen - WB : 0u;
        c.gout_ld = (valid && a.out_cap[b] >= WB) ? c.gout : g_pad;
        c.blocked = Copier<L, G, WB, NS>::K_NONE;
        c.done = valid ? 0u : 1u;
        __syncthreads();
        c.run();
    } else {
        // ---- parser: lane j owns block first + j
        const uint32_t j = lane;
        const uint32_t b = first + j;
        const bool valid = j < NB && b < a.n;
        ParserT<L> p;
        p.q.blk = lds + (j < NB ? j : 0u) * L::BLK_LDS;
        p.init_window(valid ? a.in_base + a.in_off[b] : g_pad, valid ? a.in_len[b] : 0u);
        p.rare_below = 4u;           // the copier cuts match pieces at the offset (words, or the offset's bytes in one lane)
        p.lit_slack = WB - 1u;
        p.cap = valid ? a.out_cap[b] : 0u;
        p.ip = 0u; p.op = 0u; p.tok_over = 0u; p.qtail = 0u;
        p.status = 0; p.expected = 0u;
        p.done = valid ? 0u : 1u;
#ifdef LZ4FLEX_SPLIT_DEBUG
        p.dbg_ip = 0xFFFFFFFFu; p.dbg_w0 = 0u; p.dbg_w1 = 0u; p.dbg_kb = 0u;
#endif
        p.prime();
        __syncthreads();
        // the token chain is the critical path of the workgroup: the parser wavefront issues ahead of the copiers on its SIMD
        // (priority 0: 2.80 ms instead of 2.36 ms)
        __builtin_amdgcn_s_setprio(3);
        if (valid && p.ilen == 0u) p.fail(LZ4FLEX_DEV_E_EXPECTED_ANOTHER_BYTE);   // :207-209
#ifdef LZ4FLEX_PROFILE_PHASES
        p.pr_steps = p.pr_noroom = p.pr_bubble = p.pr_slow = p.pr_pushed = 0u;
        const unsigned long long t_begin = __builtin_readcyclecounter();
        uint32_t wsteps = 0u;
        while (!__all(p.done != 0u)) { p.step(); wsteps++; }
        if (lane == 0u) { SP_ADD(0, __builtin_readcyclecounter() - t_begin); SP_ADD(1, wsteps); }
        SP_ADD(2, p.pr_steps); SP_ADD(3, p.pr_noroom); SP_ADD(4, p.pr_bubble); SP_ADD(5, p.pr_slow); SP_ADD(6, p.pr_pushed);
#else
        // A lane only finishes (or fails) inside exact_step, so the hot loop below needs no "done" test and no slow-path
        // code: it runs fast steps until some lane's sequence leaves the fast path.
        // Only a few lanes at a time are within 48 bytes of their block's end (and read the tail copy instead of the ring):
        // the step without any such lane carries no code for it.
        while (!__all(p.done != 0u)) {
            do {
                p.window();
                if (__any(p.tailmode)) {
                    asm volatile("" ::: "memory");        // keep this a branch: hipcc otherwise merges both bodies into one with selects
                    p.patch_tail();
                    p.template parse<true>();
                } else {
                    p.template parse<false>();
                }
            } while (!__any(p.slow));
            if (p.slow) p.exact_step();
        }
#endif
        if (valid) {
            a.status[b] = p.status;
            a.out_len[b] = p.status == 0 ? p.op : 0u;
            if (a.detail) {
#ifdef LZ4FLEX_SPLIT_DEBUG
                a.detail[2u * b] = ((uint64_t)p.dbg_ip << 32) | p.dbg_kb;
                a.detail[2u * b + 1u] = ((uint64_t)p.dbg_w1 << 32) | p.dbg_w0;
#else
                a.detail[2u * b] = p.status == LZ4FLEX_DEV_E_OUTPUT_TOO_SMALL ? p.expected : 0u;
                a.detail[2u * b + 1u] = p.status == LZ4FLEX_DEV_E_OUTPUT_TOO_SMALL ? (uint64_t)p.cap : 0u;
#endif
            }
        }
    }
}

template <class L, uint32_t NB, uint32_t G, uint32_t WB, uint32_t NS>
static hipError_t launch_cfg(const DecompressArgs& a, hipStream_t s) {
    const uint32_t grid = (a.n + NB - 1u) / NB;
    const size_t lds = (size_t)NB * L::BLK_LDS;
    auto kern = lz4_decompress_split_kernel<L, NB, G, WB, NS>;
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
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64u * split_waves<NB * G / 64u, G>()), lds, s, a);
    return hipGetLastError();
}

}  // namespace v5

// geometry of the 64-blocks-per-workgroup launch (the large batches): copier lanes per block, bytes per lane, pieces in flight
#ifndef LZ4S_G
#define LZ4S_G 4
#endif
#ifndef LZ4S_WB
#define LZ4S_WB 16
#endif
#ifndef LZ4S_NS
#define LZ4S_NS 4
#endif

// blocks_per_wg: 8, 16, 32 or 64 (0 = the largest that still gives every CU a workgroup).  Round 2, JSON tiles: 4 096 blocks
// (16 per workgroup) 1.31 ms, 8 192 (32) 1.55, 16 384 (64) 1.79, 32 768 (two rounds) 3.86.
hipError_t launch_decompress_split(const DecompressArgs& a, hipStream_t s, int blocks_per_wg) {
    if (a.n == 0u) return hipSuccess;
    if (a.dict_base != nullptr || a.out_pos != nullptr) return hipErrorInvalidValue;   // dictionary / prefix: v1 kernel
    // 0 = the default: 64 blocks per workgroup whatever the batch size.  The kernel takes as long as a workgroup's slowest block
    // (1.62 ms for JSON blocks), and rounds 2 and 3 spread smaller batches over more, smaller workgroups (8 / 16 / 32 blocks each, the older
    // 8-lanes-of-4-bytes copier): 8 193 ... 16 383 blocks then ran in two rounds or several workgroups per CU -- 1.92 ... 2.15 ms where 64
    // per workgroup needs 1.62 ... 1.66 on half-empty chips (tools/split_geometry.py, profiles/r04_decoder_shapes.txt)
    if (blocks_per_wg == 0) blocks_per_wg = 64;
    switch (blocks_per_wg) {
        case 64: return v5::launch_cfg<v5::LayoutBig, 64, LZ4S_G, LZ4S_WB, LZ4S_NS>(a, s);
        case 32: return v5::launch_cfg<v5::LayoutBig, 32, 8, 4, 4>(a, s);
        case 16: return v5::launch_cfg<v5::LayoutBig, 16, 8, 4, 4>(a, s);
        case 8: return v5::launch_cfg<v5::LayoutBig, 8, 8, 4, 4>(a, s);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace lz4flex_dev

#ifdef LZ4FLEX_PROFILE_PHASES
extern "C" int lz4flex_debug_phase_split(unsigned long long* vals, int reset) {
    if (reset) {
        unsigned long long z[16] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(lz4flex_dev::g_split_prof), z, sizeof z);
        return 0;
    }
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(vals, HIP_SYMBOL(lz4flex_dev::g_split_prof), 128);
    return 0;
}
#endif
