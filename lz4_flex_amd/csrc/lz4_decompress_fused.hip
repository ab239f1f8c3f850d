// lz4_decompress_fused.hip -- batched LZ4 block decoder, PARSER -> CUTTER -> QUADS in one workgroup of 64 blocks ("v6").
//
// Same contract as lz4_decompress_split.hip (reference src/block/decompress.rs:201-449: result bytes, byte count, error variant and
// OutputTooSmall{expected,actual}, unsafe-flavour check order; blocks without dictionary / prefix), same parser
// (lz4_split_parser.h: one lane per block, every bounds check of the reference).  What changed is everything behind the parser's
// queue (lz4_fused_common.h says why):
//   * CUTTER wavefront, one lane per block: cuts the parser's sequence records into pieces (<= 64 bytes, never reading what they
//     write, never crossing the ring's end), one 4-byte word per piece into the block's piece queue;
//   * four QUAD wavefronts, four lanes per block: per step a quad takes the pieces at the head of its queue that fit four lanes and do
//     not depend on each other (pack_step: every lane for itself), requests memory sources -- literals, matches further back than
//     the ring -- with an exec-masked 16-byte global load LOOKAHEAD steps before the step executes (inline assembly with hand-counted
//     waits, lz4_decompress_replay.hip's scheme), and executes: one LDS read for near sources, four ordered 16-byte LDS writes into
//     the block's 1 KiB output ring; complete 64-byte lines of the output leave the ring every fourth step.  A literal run that may
//     touch the block's last byte, or that is longer than the parser's window, is a SPECIAL: the quad stops fetching behind it, copies
//     it through the ring with exact bounds when its step executes, and goes on.
// Wavefronts of a workgroup are dealt to the four SIMDs in turn: the parser (wavefront 3) and the cutter (2) have a SIMD each,
// the quads (0, 1, 4, 5) share two; wavefronts 6 and 7 only exist to make that placement and end at the first barrier.
// Blocks of 512 KiB or more (compressed or sink) are left with status `redo_code` for the reference-order kernel, like the blocks
// the other fast decoders mark.
// Round 6: compiled in -DLZ4FLEX_TOOLS builds only (decompress_variant 12 is refused by the product library): level with the split decoder on JSON,
// behind it on incompressible data and runs -- an experiment, and the product ships none by default.
#ifdef LZ4FLEX_TOOLS
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lz4_device.h"
#include "lz4_fused_common.h"

namespace lz4flex_dev {
namespace fused {

constexpr uint32_t NB = 64u;             // blocks per workgroup
constexpr uint32_t L = LOOKAHEAD;
typedef __attribute__((address_space(1))) uint8_t g_u8;

__device__ __attribute__((aligned(64))) uint8_t g_fused_pad[64];   // the sink / source of lane groups without a block
#ifdef LZ4F_PROF     // tools: [0] parser cycles (lane 0 of every workgroup), [1] parser steps, [2] quad cycles, [3] quad turns, [4] quad turns with steps (lane sums / 4),
                     // [5] emitter cycles, [6] emitter iterations, [7] lane-iterations with a piece placed, [8] ... that could not write a step (queue full), [9] ... with nothing to do
__device__ unsigned long long g_fused_prof[16];
#define FP_ADD(k, v) atomicAdd(&g_fused_prof[k], (unsigned long long)(v))
#endif

// what a lane keeps of a step between its fetch and its execution: n [31:27] | kind [26:25] | rel [24:19] | field [18:0]
// (kind K_END: a resting step -- padding, or field = SP_*: a special step, whose literal run is v.x = position, v.y = length)
constexpr uint32_t N_SHIFT = 27u, KIND_SHIFT = 25u, REL_SHIFT = 19u;
constexpr uint32_t KIND_MASK = 3u << KIND_SHIFT;
struct Slot {
    uint32_t r;     // the lane's job
    u32x4 v;        // its bytes, if they come from memory (K_LIT, K_FAR)
};
// see lz4_decompress_replay.hip: loads under an execution mask with hand-counted waits ("lz4f-load" / "lz4f-wait <registers>":
// lz4_flex_amd/build.py checks on the shipped ISA that nothing touches the registers in between; -DLZ4F_PLAIN_LOADS is the fallback)
__device__ __forceinline__ void slot_load(u32x4& dst, const uint8_t* p, uint64_t mask) {
#ifdef LZ4F_PLAIN_LOADS
    if (__builtin_amdgcn_inverse_ballot_w64(mask)) __builtin_memcpy(&dst, p, 16);
#else
    uint64_t save;
    asm volatile("s_mov_b64 %1, exec\n\ts_mov_b64 exec, %3\n\tglobal_load_dwordx4 %0, %2, off ; lz4f-load\n\ts_mov_b64 exec, %1"
                 : "+v"(dst), "=&s"(save) : "v"(p), "s"(mask) : "memory");
#endif
}
template <int N>
__device__ __forceinline__ void slot_wait(u32x4& v) {
#ifndef LZ4F_PLAIN_LOADS
    asm volatile("s_waitcnt vmcnt(%1) ; lz4f-wait %0" : "+v"(v) : "n"(N) : "memory");
#endif
}

struct Quad {
    const uint8_t* in_b;     // the compressed block
    const uint8_t* out_b;    // the block's sink (far sources)
    uint8_t* out_wr;         // the block's sink + 16 g (write-back)
    lds_u8* blk;             // the block's LDS area
    lds_u8* ring;            // its output ring
    uint32_t g, g16;
    uint32_t ilen;
    uint64_t lane_is[G];     // execution masks: lane g of every quad
    uint32_t op;             // output position
    uint32_t F;              // lines below F are in memory (a multiple of 64)
    uint32_t pi;             // next word of the piece queue the front end takes
    uint32_t opf;            // output position of the next step the front end packs
    uint32_t hold;           // a special is in the slots: nothing is fetched behind it until it is served
    uint32_t done;
    uint32_t sp, sp_src, sp_len;     // the special step of this turn (0: none)

    // fetch: the pieces at the head of the queue -> a step -> this lane's job; request its bytes, if they come from memory (an
    // instruction whose mask is empty still counts).  pt = the queue's tail (a snapshot)
    __device__ __forceinline__ void front(Slot& s, uint32_t pt) {
        const lds_u8* pq = blk + Layout::PQ_OFF;
        const uint32_t w0 = *reinterpret_cast<const lds_vu32*>(pq + 4u * (pi & (PQ - 1u)));
        const uint32_t w1 = *reinterpret_cast<const lds_vu32*>(pq + 4u * ((pi + 1u) & (PQ - 1u)));
        const uint32_t w2 = *reinterpret_cast<const lds_vu32*>(pq + 4u * ((pi + 2u) & (PQ - 1u)));
        const uint32_t w3 = *reinterpret_cast<const lds_vu32*>(pq + 4u * ((pi + 3u) & (PQ - 1u)));
        const uint32_t have = pt - pi;
        const uint32_t avail = (hold | done) != 0u ? 0u : (have < 4u ? have : 4u);
        const LaneJob J = pack_step(w0, w1, w2, w3, avail, opf, g);
        const bool special = J.special != 0u;
        pi += J.take;
        opf += J.total + (special ? w2 : 0u);
        hold = special ? 1u : hold;
        const uint32_t r = special ? (K_END << KIND_SHIFT) | J.special : (J.n << N_SHIFT) | (J.kind << KIND_SHIFT) | (J.rel << REL_SHIFT) | (J.src & MASK);
        const uint8_t* p = (J.kind == K_FAR ? out_b : in_b) + J.src;
        const bool need = J.n != 0u && (J.kind - 1u) < 2u;
        s.v.x = special ? w1 : s.v.x;         // (before the load: nothing touches the slot's registers between the load and its wait)
        s.v.y = special ? w2 : s.v.y;
        slot_load(s.v, p, __builtin_amdgcn_ballot_w64(need));
        s.r = r;
    }
    // execute a step: the lanes read their sources (near ones from the ring) at once, then write in lane order, 16 bytes each: what a
    // lane writes beyond its n bytes is overwritten by the next lane (by the next step behind the last one)
    __device__ __forceinline__ void back(Slot& s) {
        const uint32_t r = s.r;
        const bool rest = (r & KIND_MASK) == KIND_MASK;
        const uint32_t n = rest ? 0u : r >> N_SHIFT;
        slot_wait<(int)L - 1>(s.v);          // marked loads since this slot's: the other L - 1 slots (stores and a special step's traffic only make it stricter)
        u32x4 x = s.v;
        const bool active = n != 0u;
        if (active && (r & KIND_MASK) == 0u) __builtin_memcpy(&x, (const void*)(ring + (r & MASK)), 16);
        lds_u8* dst = ring + ((op + ((r >> REL_SHIFT) & 63u)) & MASK);
        const uint64_t act = __builtin_amdgcn_ballot_w64(active);
        uint64_t save;
        asm volatile("s_mov_b64 %0, exec\n\t"
                     "s_and_b64 exec, %3, %4\n\tds_write_b128 %1, %2\n\t"
                     "s_and_b64 exec, %3, %5\n\tds_write_b128 %1, %2\n\t"
                     "s_and_b64 exec, %3, %6\n\tds_write_b128 %1, %2\n\t"
                     "s_and_b64 exec, %3, %7\n\tds_write_b128 %1, %2\n\t"
                     "s_mov_b64 exec, %0"
                     : "=&s"(save) : "v"(dst), "v"(x), "s"(act), "s"(lane_is[0]), "s"(lane_is[1]), "s"(lane_is[2]), "s"(lane_is[3]) : "memory");
        // the step's bytes: the sum of the quad's four n
        uint32_t t = n + (uint32_t)__builtin_amdgcn_mov_dpp((int)n, 0xB1, 0xF, 0xF, true);     // quad_perm [1,0,3,2]
        t += (uint32_t)__builtin_amdgcn_mov_dpp((int)t, 0x4E, 0xF, 0xF, true);                   // quad_perm [2,3,0,1]
        op += t;
        // a special step is served at the end of the turn (nothing but resting steps follows it for two turns)
        const uint32_t code = rest ? (r & 3u) : 0u;
        sp = code != 0u ? code : sp;
        sp_src = code != 0u ? x.x : sp_src;
        sp_len = code != 0u ? x.y : sp_len;
    }
    // the 64-byte lines of the output that are complete leave the ring, 16 aligned bytes per lane
    __device__ __forceinline__ void flush() {
        while (F + PIECE <= op) {
            const u32x4 y = *reinterpret_cast<const u32x4 __attribute__((address_space(3)))*>(ring + g16 + (F & MASK));
            __builtin_memcpy(out_wr + F, &y, 16);
            F += PIECE;
        }
    }
    // a literal run with exact bounds, through the ring: chunks that end at the ring's end, 16-byte units dealt to the quad's lanes
    // (a unit whose 16 bytes would leave the compressed block is read byte by byte), every chunk's complete lines written back at once
    // (inlined: a member function that is called takes the whole Quad to memory)
    __device__ __forceinline__ void serve() {
        uint32_t src = sp_src, n = sp_len;
        while (n != 0u) {
            const uint32_t room = W - 128u - (op - F);                      // (op - F < 64 behind a flush: room >= 832)
            uint32_t chunk = n < room ? n : room;
            const uint32_t to_wrap = W - (op & MASK);
            chunk = chunk < to_wrap ? chunk : to_wrap;
            for (uint32_t o = 16u * g; o < chunk; o += 16u * G) {
                u32x4 v = {0u, 0u, 0u, 0u};
                if (src + o + 16u <= ilen) {
                    __builtin_memcpy(&v, in_b + src + o, 16);
                } else {
                    uint8_t t[16];
                    for (uint32_t k = 0; k < 16u; ++k) t[k] = (src + o + k < ilen) ? in_b[src + o + k] : (uint8_t)0;
                    __builtin_memcpy(&v, t, 16);
                }
                // (a unit's bytes behind the chunk's end land on ring positions that hold nothing a near source may still name, or in the pad)
                *reinterpret_cast<u32x4 __attribute__((address_space(3)))*>(ring + ((op + o) & MASK)) = v;
            }
            op += chunk; src += chunk; n -= chunk;
            flush();
        }
        if (sp == SP_FINISH) {
            // what is still in the ring leaves it: lines, then bytes (exact writes: the sink may end here)
            flush();
            uint8_t* sink = out_wr - g16;                                                  // (out_wr = sink + 16 g)
            for (uint32_t k = F + g; k < op; k += G) sink[k] = ring[k & MASK];
            F = op;
            done = 1u;
        }
        sp = 0u;
        hold = 0u;
    }
};

__global__ void __launch_bounds__(512) lz4_decompress_fused_kernel(DecompressArgs a, int32_t redo_code) {
    extern __shared__ __attribute__((aligned(16))) uint8_t dyn_lds[];
    lds_u8* lds = (lds_u8*)dyn_lds;
    constexpr uint32_t BLK_LDS = Layout::BLK_LDS, TAIL_OFF = Layout::TAIL_OFF;
    const uint32_t pw = threadIdx.x / 64u, lane = threadIdx.x % 64u;
    const uint32_t first = blockIdx.x * NB;
    if (pw >= 6u) { __syncthreads(); return; }                      // (placement only; they take part in the workgroup's one barrier)
    if (pw != 2u && pw != 3u) {
        // ---- quads: set up the block's queues and the parser's tail copy, then run
        const uint32_t qw = pw < 2u ? pw : pw - 2u;
        const uint32_t j = qw * 16u + lane / G;
        const uint32_t b = first + j;
        const bool valid = b < a.n && a.in_len[b < a.n ? b : 0u] <= MAX_FIELD && a.out_cap[b < a.n ? b : 0u] <= MAX_FIELD;
        Quad Q;
        Q.g = lane % G; Q.g16 = LANE_B * Q.g;
        Q.blk = lds + j * BLK_LDS;
        Q.ring = Q.blk + Layout::OUT_OFF;
        Q.in_b = valid ? a.in_base + a.in_off[b] : g_fused_pad;
        Q.out_b = valid ? a.out_base + a.out_off[b] : g_fused_pad;
        Q.out_wr = const_cast<uint8_t*>(Q.out_b) + Q.g16;
        Q.ilen = valid ? a.in_len[b] : 0u;
        Q.op = 0u; Q.F = 0u; Q.pi = 0u; Q.opf = 0u; Q.hold = 0u; Q.sp = 0u; Q.sp_src = 0u; Q.sp_len = 0u;
        Q.done = valid ? 0u : 1u;
        for (uint32_t k = 0; k < G; ++k) {
            Q.lane_is[k] = 0x1111111111111111ull << k;
            asm volatile("" : "+s"(Q.lane_is[k]));       // (kept in scalar registers: as immediates they are rebuilt at every use)
        }
        const uint32_t tstart = Q.ilen > v5::TAILB ? Q.ilen - v5::TAILB : 0u;
        for (uint32_t i = Q.g; i < v5::TAIL_BUF; i += G) Q.blk[TAIL_OFF + i] = (tstart + i < Q.ilen) ? Q.in_b[tstart + i] : (uint8_t)0;
        if (Q.g == 0u) {
            lds_vu32* ctl = reinterpret_cast<lds_vu32*>(Q.blk + Layout::CTL_OFF);
            ctl[0] = 0u; ctl[1] = 0u; ctl[2] = 0u; ctl[3] = 0u;
        }
        Slot sl[L];
#pragma unroll
        for (uint32_t i = 0; i < L; ++i) { sl[i].r = 0u; sl[i].v = u32x4{0u, 0u, 0u, 0u}; }
        __syncthreads();
#ifdef LZ4F_PROF
        const unsigned long long tq0 = __builtin_readcyclecounter();
        uint32_t pq_turns = 0u, pq_ok = 0u;
#endif
        for (;;) {
            // a turn: L steps leave the slots, L steps enter them (as far as the block's queue holds pieces)
            const uint32_t pt = *reinterpret_cast<lds_vu32*>(Q.blk + PIECE_TAIL);
            const uint32_t pi0 = Q.pi;
#pragma unroll
            for (uint32_t i = 0; i < L; ++i) {
                Q.back(sl[i]);
                if (i % FLUSH_EVERY == FLUSH_EVERY - 1u) Q.flush();
                Q.front(sl[i], pt);
            }
#ifdef LZ4F_PROF
            pq_turns++; pq_ok += Q.pi != pi0;
#endif
            if (Q.g == 0u) *reinterpret_cast<lds_vu32*>(Q.blk + PIECE_HEAD) = Q.pi;
            if (__any(Q.sp != 0u)) {
                if (Q.sp != 0u) Q.serve();
            }
            if (__all(Q.done != 0u)) break;
            if (!__any(Q.pi != pi0)) __builtin_amdgcn_s_sleep(4);          // nothing fetched in the whole wavefront: yield issue slots
        }
#ifndef LZ4F_PLAIN_LOADS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // requests still in flight own their registers until they land
#endif
#ifdef LZ4F_PROF
        if (Q.g == 0u) FP_ADD(4, pq_ok);
        if (lane == 0u) { FP_ADD(2, __builtin_readcyclecounter() - tq0); FP_ADD(3, pq_turns); }
#endif
    } else if (pw == 2u) {
        // ---- cutter: lane j owns block first + j
        const uint32_t j = lane, b = first + j;
        const bool valid = b < a.n && a.in_len[b < a.n ? b : 0u] <= MAX_FIELD && a.out_cap[b < a.n ? b : 0u] <= MAX_FIELD;
        Cutter cu;
        cu.init(lds + j * BLK_LDS);
        __syncthreads();
        __builtin_amdgcn_s_setprio(2);
        bool alive = valid;
#ifdef LZ4F_PROF
        const unsigned long long te0 = __builtin_readcyclecounter();
        uint32_t pe_it = 0u, pe_place = 0u, pe_full = 0u, pe_idle = 0u;
        while (__any(alive)) {
            const uint32_t op0 = cu.op, hd0 = cu.head;
            const bool full = alive && (cu.ptail - cu.piece_head()) > PQ - 3u;
            alive = cu.iterate(alive) && alive;
            pe_it++; pe_place += cu.op != op0; pe_full += full; pe_idle += alive && cu.op == op0 && cu.head == hd0 && !full;
        }
        FP_ADD(7, pe_place); FP_ADD(8, pe_full); FP_ADD(9, pe_idle);
        if (lane == 0u) { FP_ADD(5, __builtin_readcyclecounter() - te0); FP_ADD(6, pe_it); }
#else
        while (__any(alive)) alive = cu.iterate(alive) && alive;
#endif
    } else {
        // ---- parser: lane j owns block first + j (lz4_decompress_split.hip's, with the quads' word width)
        const uint32_t j = lane, b = first + j;
        const bool inb = b < a.n;
        const bool valid = inb && a.in_len[inb ? b : 0u] <= MAX_FIELD && a.out_cap[inb ? b : 0u] <= MAX_FIELD;
        v5::ParserT<Layout> p;
        p.q.blk = lds + j * BLK_LDS;
        p.init_window(valid ? a.in_base + a.in_off[b] : v5::g_pad, valid ? a.in_len[b] : 0u);
        p.rare_below = 0u;           // short periods are the cutter's business (doubling pieces)
        p.lit_slack = LANE_B - 1u;
        p.cap = valid ? a.out_cap[b] : 0u;
        p.ip = 0u; p.op = 0u; p.tok_over = 0u; p.qtail = 0u;
        p.status = 0; p.expected = 0u;
        p.done = valid ? 0u : 1u;
        p.prime();
        __syncthreads();
        __builtin_amdgcn_s_setprio(3);
        if (valid && p.ilen == 0u) p.fail(LZ4FLEX_DEV_E_EXPECTED_ANOTHER_BYTE);   // :207-209
#ifdef LZ4F_PROF
        const unsigned long long tp0 = __builtin_readcyclecounter();
        uint32_t pp_steps = 0u;
#endif
        while (!__all(p.done != 0u)) {
            do {
#ifdef LZ4F_PROF
                pp_steps++;
#endif
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
#ifdef LZ4F_PROF
        if (lane == 0u) { FP_ADD(0, __builtin_readcyclecounter() - tp0); FP_ADD(1, pp_steps); }
#endif
        if (inb) {
            a.status[b] = valid ? p.status : redo_code;
            a.out_len[b] = (valid && p.status == 0) ? p.op : 0u;
            if (a.detail) {
                a.detail[2u * b] = (valid && p.status == LZ4FLEX_DEV_E_OUTPUT_TOO_SMALL) ? p.expected : 0u;
                a.detail[2u * b + 1u] = (valid && p.status == LZ4FLEX_DEV_E_OUTPUT_TOO_SMALL) ? (uint64_t)p.cap : 0u;
            }
        }
    }
}

}  // namespace fused

hipError_t launch_decompress_fused(const DecompressArgs& a, int32_t redo_code, hipStream_t s) {
    if (a.n == 0u) return hipSuccess;
    if (a.dict_base != nullptr || a.out_pos != nullptr) return hipErrorInvalidValue;   // dictionary / prefix: other kernels
    const uint32_t grid = (a.n + fused::NB - 1u) / fused::NB;
    const size_t lds = (size_t)fused::NB * fused::Layout::BLK_LDS;
    auto kern = fused::lz4_decompress_fused_kernel;
    static unsigned long long have = 0ull;   // the attribute is per device (benign race: setting it twice is harmless)
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(have & bit)) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        have |= bit;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, s, a, redo_code);
    return hipGetLastError();
}

}  // namespace lz4flex_dev

#ifdef LZ4F_PROF
extern "C" int lz4flex_debug_fused_prof(unsigned long long* vals, int reset) {
    if (reset) {
        unsigned long long z[16] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(lz4flex_dev::fused::g_fused_prof), z, sizeof z);
        return 0;
    }
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(vals, HIP_SYMBOL(lz4flex_dev::fused::g_fused_prof), 128);
    return 0;
}
#endif

#endif  // LZ4FLEX_TOOLS
