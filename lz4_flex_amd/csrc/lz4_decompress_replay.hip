// lz4_decompress_replay.hip -- the REPLAY half of the plan / replay decoder: executes the copy plans the plan kernel
// (lz4_decompress_plan.hip) left, FOUR LANES PER BLOCK, 64 blocks per workgroup (one workgroup per CU: the batch shape
// of BASELINE configs[1], 16 384 blocks = 64 per CU).  Reference semantics: src/block/decompress.rs:334-437 (the copies of
// the decode loop); everything else of that loop -- token chain, lengths, checks -- happened in the plan kernel.
//
// A block's copies are a serial chain (a match may read what the previous sequence wrote), so this kernel's time is
// (records per block) x (time per record) whatever the batch size, and the only thing to optimise is the time per record:
//   * a step is four 4-byte records, one per lane; a lane's records of 24 steps arrive as 96 consecutive bytes -- no queue,
//     no LDS traffic for them, no pointer chasing;
//   * the plan kernel cut every copy so that a step needs no decision: a piece never reads what it writes, never wraps
//     the ring, 16-byte moves per lane never leave the buffers, and pieces that do not depend on each other share a step
//     (lz4_plan_common.h);
//   * the block's last 2 KiB of output live in an LDS ring (near matches: 65 % on JSON); a 64-byte line of the output goes
//     to memory in the step that completes it; far matches and literals are global loads issued LOOKAHEAD records before
//     their step;
//   * no error can occur here: irregular blocks have no plan (flags != 0) and are left to the reference-order kernel.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lz4_device.h"
#include "lz4_plan_common.h"

namespace lz4flex_dev {
namespace plan {

typedef uint32_t __attribute__((ext_vector_type(4))) u32x4;
typedef uint8_t __attribute__((address_space(3))) lds_u8;

// turns of K_END: what a lane group without a block (batch tail, irregular block) replays
#define LZ4R_E4 END_REC, END_REC, END_REC, END_REC
#define LZ4R_E24 LZ4R_E4, LZ4R_E4, LZ4R_E4, LZ4R_E4, LZ4R_E4, LZ4R_E4
#define LZ4R_TURN_END LZ4R_E24, LZ4R_E24, LZ4R_E24, LZ4R_E24
__device__ __attribute__((aligned(16))) uint32_t g_end_turns[(END_TURNS + 1u) * TURN_WORDS] = {LZ4R_TURN_END, LZ4R_TURN_END, LZ4R_TURN_END, LZ4R_TURN_END};
#undef LZ4R_TURN_END
#undef LZ4R_E24
#undef LZ4R_E4
static_assert(END_TURNS == 3u && TURN_WORDS == 96u, "g_end_turns");
__device__ __attribute__((aligned(64))) uint8_t g_replay_pad[64];   // the sink / source of lane groups without a block

#ifndef LZ4R_GROUPS_PER_WAVE
#define LZ4R_GROUPS_PER_WAVE 16
#endif
constexpr uint32_t GPW = LZ4R_GROUPS_PER_WAVE;     // lane groups per wavefront in use (16: every lane; measured with 8 -- twice the wavefronts, half the lanes each: 1.8 times slower, a step's cost is its instructions, not its lanes)
constexpr uint32_t NB = 4u * GPW;        // blocks per workgroup (four wavefronts)
static_assert(TURN_STEPS == 24u && TURN_STEPS % FLUSH_EVERY == 0u, "geometry");

// Memory sources are requested LOOKAHEAD steps before their use, by the lanes that need them only (a 16-byte load costs the
// CU's texture path the same whether its bytes are wanted or not, and every wavefront pays for every lane: the first
// version loaded in every lane at every step and spent a quarter of its time there).  hipcc cannot express that -- it waits
// for a conditional load where it is issued, and it counts the loads it knows about -- so the loads of this kernel's loop are
// inline assembly with hand-counted waits: "lz4r-load" under an execution mask (an instruction whose mask is empty still
// counts), "lz4r-wait <registers>" = s_waitcnt vmcnt(N) with N = the marked loads issued since (loads and stores the compiler
// adds in between only make the wait stricter).  Nothing may touch the destination registers between load and wait, which
// the compiler does not know: lz4_flex_amd/build.py checks exactly that on the ISA it ships (check_async_loads, also run by
// tests/test_isa_checks.py) and builds with -DLZ4R_PLAIN_LOADS -- every lane loads, the compiler waits -- if a toolchain
// ever breaks it.
struct Slot {
    uint32_t r;     // the lane's record
    u32x4 v;        // its bytes, if they come from memory (K_LIT, K_FAR)
};

__device__ __forceinline__ void slot_load(u32x4& dst, const uint8_t* p, uint64_t mask) {
#ifdef LZ4R_PLAIN_LOADS
    __builtin_memcpy(&dst, p, 16);
#else
    uint64_t save;
#ifndef LZ4R_LOAD_BITS
#define LZ4R_LOAD_BITS ""
#endif
    asm volatile("s_mov_b64 %1, exec\n\ts_mov_b64 exec, %3\n\tglobal_load_dwordx4 %0, %2, off" LZ4R_LOAD_BITS " ; lz4r-load\n\ts_mov_b64 exec, %1"
                 : "+v"(dst), "=&s"(save) : "v"(p), "s"(mask) : "memory");    // "+v": the slot keeps its registers from turn to turn (no copies of registers in flight)
#endif
}
template <int N>
__device__ __forceinline__ void slot_wait(u32x4& v) {
#ifndef LZ4R_PLAIN_LOADS
    asm volatile("s_waitcnt vmcnt(%1) ; lz4r-wait %0" : "+v"(v) : "n"(N) : "memory");
#endif
}
// two of a lane's records in flight
__device__ __forceinline__ void pair_issue(uint64_t& d, const uint32_t* p) {
#ifdef LZ4R_PLAIN_LOADS
    __builtin_memcpy(&d, p, 8);
#else
    asm volatile("global_load_dwordx2 %0, %1, off ; lz4r-load" : "+v"(d) : "v"(p) : "memory");
#endif
}
// ... have landed: the wait does not redefine the register in flight (an in-out operand lets the compiler copy the register
// BEFORE the wait it knows nothing about -- it did, for registers that stay alive behind it); the same statement hands the
// value on in another register
template <int N>
__device__ __forceinline__ uint64_t pair_take(const uint64_t& d) {
#ifdef LZ4R_PLAIN_LOADS
    return d;
#else
    uint64_t r;
    asm volatile("s_waitcnt vmcnt(%2) ; lz4r-wait %1\n\tv_mov_b64 %0, %1" : "=&v"(r) : "v"(d), "n"(N) : "memory");
    return r;
#endif
}

struct Lane {
    const uint8_t* in_b;     // the compressed block
    const uint8_t* out_b;    // the block's sink (far sources)
    uint8_t* out_wr;         // the block's sink + 16 g (write-back)
    lds_u8* ring;            // the block's ring
    uint32_t g16;            // 16 g
#ifdef LZ4R_PLAIN_LOADS
    const uint8_t* idle;
#endif
    uint64_t lane_is[G];     // execution masks: lane g of every group
    uint32_t op;             // output position
    uint32_t F;              // lines below F are in memory (a multiple of 64)

    // request the bytes of the lane's record, if they come from memory (an instruction whose mask is empty still counts)
    __device__ __forceinline__ void front(Slot& s, uint32_t r) {
        const uint32_t kind = (r >> KIND_SHIFT) & 3u;
#ifdef LZ4R_EXP_FARNEAR     // timing experiment (wrong output): far sources at most 4 KiB behind the block's start
        const uint32_t field = kind == K_FAR ? (r & 0xFFFu) : (r & MAX_FIELD);
#else
        const uint32_t field = r & MAX_FIELD;
#endif
        const uint8_t* p = (kind == K_FAR ? out_b : in_b) + field;
        const bool need = (kind - 1u) < 2u && (r >> N_SHIFT) != 0u;
#ifdef LZ4R_PLAIN_LOADS
        p = need ? p : idle;                 // every lane loads: a lane without a source reads its part of the plan that was fetched last
        __builtin_memcpy(&s.v, p, 16);
#elif defined(LZ4R_EXP_NOLOAD)      // timing experiments only (wrong output): no memory sources
        asm volatile("" :: "v"(p));
        s.v = u32x4{r, r, r, r};
#else
        slot_load(s.v, p, __builtin_amdgcn_ballot_w64(need));
#endif
        s.r = r;
    }
    // execute a step: the lanes read their sources (near ones from the ring) at once, then write in lane order, 16 bytes each:
    // what a lane writes beyond its n bytes is overwritten by the next lane (by the next step behind the last one)
    __device__ __forceinline__ void back(Slot& s) {
        const uint32_t r = s.r;
        const uint32_t n = r >> N_SHIFT;
        slot_wait<LOOKAHEAD - 1 + (int)(TURN_STEPS / 2u)>(s.v);   // marked loads since this slot's: the other LOOKAHEAD - 1 slots and a turn's record pairs
        u32x4 x = s.v;
        const bool active = n != 0u;
#ifndef LZ4R_EXP_NORING
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
#else
        asm volatile("" :: "v"(x));
#endif
        // the step's bytes: the sum of the group's four n
        uint32_t t = n + (uint32_t)__builtin_amdgcn_mov_dpp((int)n, 0xB1, 0xF, 0xF, true);     // quad_perm [1,0,3,2]
        t += (uint32_t)__builtin_amdgcn_mov_dpp((int)t, 0x4E, 0xF, 0xF, true);                   // quad_perm [2,3,0,1]
        op += t;
    }
    // every FLUSH_EVERY steps: the 64-byte lines of the output that are complete leave the ring, 16 aligned bytes per lane -- one
    // full-line write per 64 bytes of output (the first version stored every piece where it ended: the memory system counts
    // transactions, and so does the wavefront: a store instruction costs it as much as a load)
    __device__ __forceinline__ void flush() {
        while (F + PIECE <= op) {
#ifndef LZ4R_EXP_NORING
            const u32x4 y = *reinterpret_cast<const u32x4 __attribute__((address_space(3)))*>(ring + g16 + (F & MASK));
#else
            const u32x4 y = u32x4{F, op, F, op};
#endif
#ifndef LZ4R_EXP_NOSTORE
            __builtin_memcpy(out_wr + F, &y, 16);
#else
            asm volatile("" :: "v"(y));
#endif
            F += PIECE;
        }
    }
};

__global__ void __launch_bounds__(256) lz4_replay_kernel(ReplayArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t dyn_lds[];
    lds_u8* lds = (lds_u8*)dyn_lds;
    const uint32_t lane = threadIdx.x % 64u, g = threadIdx.x % G;
    const uint32_t j = (threadIdx.x / 64u) * GPW + (lane / G) % GPW;
    const uint32_t b = blockIdx.x * NB + j;
    const bool valid = b < a.n && lane / G < GPW;
    BlockPlan bp;
    bp.in_off = 0; bp.out_off = 0; bp.first_word = 0; bp.tail_word = 0; bp.tail_op = 0; bp.n_tail = 0; bp.flags = 1;
    if (valid) bp = a.plans[b];
    const bool live = valid && bp.flags == 0u;
    Lane L;
    L.g16 = LANE_B * g;
    L.in_b = live ? a.in_base + bp.in_off : g_replay_pad;
    L.out_b = live ? a.out_base + bp.out_off : g_replay_pad;
    L.out_wr = (live ? a.out_base + bp.out_off : g_replay_pad) + L.g16;
    L.ring = lds + j * RING_STRIDE;
    L.op = 0u; L.F = 0u;
    for (uint32_t k = 0; k < G; ++k) {
        L.lane_is[k] = 0x1111111111111111ull << k;
        asm volatile("" : "+s"(L.lane_is[k]));       // (kept in scalar registers: as immediates they are rebuilt at every use)
    }
    // the plan: four steps are 64 bytes, 16 per lane
    const uint32_t* lp = (live ? a.words + bp.first_word : g_end_turns) + 4u * g;
    Slot sl[LOOKAHEAD];
    for (uint32_t i = 0; i < LOOKAHEAD; ++i) sl[i].v = u32x4{0u, 0u, 0u, 0u};
    // The lane's records arrive four steps at a time (16 bytes per lane, 64 per lane group: one memory transaction per group
    // and four steps) in NP register pairs.  When the front end (which looks at the records LOOKAHEAD steps ahead of the back
    // end) reaches a pair, its records are taken out and the pair is refilled, in place, with the same steps of the turn
    // behind.  Marked loads between a refill and its use: the slots of LOOKAHEAD steps and the other NP - 1 pairs.
    constexpr uint32_t NP = TURN_STEPS / 2u;
    uint64_t d[NP];
    for (uint32_t i = 0; i < NP; ++i) d[i] = 0ull;
    uint64_t t0, t1;
    bool done;                   // the turn in the slots ends in K_END: the block's last step is in it (or behind us)
#ifdef LZ4R_PLAIN_LOADS
#define LZ4R_IDLE() L.idle = (const uint8_t*)lp;
#else
#define LZ4R_IDLE()
#endif
#define LZ4R_FRONT(i, w) L.front(sl[i], w);
#define LZ4R_STEP(i, w) L.back(sl[i]); if ((i) % FLUSH_EVERY == FLUSH_EVERY - 1u) L.flush(); LZ4R_FRONT(i, w)
#define LZ4R_PART(M, k) t0 = pair_take<LOOKAHEAD + (int)NP - 2>(d[2 * (k)]); t1 = pair_take<LOOKAHEAD + (int)NP - 2>(d[2 * (k) + 1]);     \
                        pair_issue(d[2 * (k)], lp + 16 * (k)); pair_issue(d[2 * (k) + 1], lp + 16 * (k) + 2);                         \
                        M(4 * (k), (uint32_t)t0) M(4 * (k) + 1, (uint32_t)(t0 >> 32)) M(4 * (k) + 2, (uint32_t)t1) M(4 * (k) + 3, (uint32_t)(t1 >> 32))
#define LZ4R_TURN(M) LZ4R_PART(M, 0) LZ4R_PART(M, 1) LZ4R_PART(M, 2) LZ4R_PART(M, 3) LZ4R_PART(M, 4) LZ4R_PART(M, 5)
    for (uint32_t k = 0; k < NP; ++k) pair_issue(d[k], lp + 16u * (k / 2u) + 2u * (k % 2u));
    LZ4R_IDLE()
    lp += TURN_WORDS;
#ifndef LZ4R_PLAIN_LOADS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    LZ4R_TURN(LZ4R_FRONT)                                // prologue: request the bytes of the first turn's records; w: the second turn's
    done = (sl[LOOKAHEAD - 1u].r & KIND_MASK) == KIND_MASK;
    for (uint32_t turn = 0u; turn < a.max_turns; ++turn) {   // (bounded: a damaged plan costs time, not the GPU)
        LZ4R_IDLE()
        lp += done ? 0u : TURN_WORDS;                    // (a block that is about to finish stays inside its K_END turns)
        LZ4R_TURN(LZ4R_STEP)
        if (__all(done)) break;
        done = (sl[LOOKAHEAD - 1u].r & KIND_MASK) == KIND_MASK;
    }
#ifndef LZ4R_PLAIN_LOADS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // requests still in flight own their registers until they land
#endif
#undef LZ4R_TURN
#undef LZ4R_PART
#undef LZ4R_STEP
#undef LZ4R_FRONT
#undef LZ4R_IDLE
    // what is still in the ring leaves it, then the tail: the block's last few pieces, byte by byte in memory (exact reads, exact
    // writes)
    if (live && g == 0u) {
        const uint8_t* in = a.in_base + bp.in_off;
        uint8_t* out = a.out_base + bp.out_off;
        for (uint32_t k = L.F; k < L.op; ++k) out[k] = L.ring[k & MASK];
        uint32_t op = L.op;
        for (uint32_t t = 0u; t < bp.n_tail; ++t) {
            const uint32_t r = a.words[bp.tail_word + t];
            const uint32_t n = rec_n(r), field = rec_field(r);
            if (rec_kind(r) == K_LIT) {
                for (uint32_t k = 0u; k < n; ++k) out[op + k] = in[field + k];
            } else {
                for (uint32_t k = 0u; k < n; ++k) out[op + k] = out[op + k - field];
            }
            op += n;
        }
    }
}

}  // namespace plan

hipError_t launch_replay(const ReplayArgs& a, hipStream_t s) {
    if (a.n == 0u) return hipSuccess;
    const uint32_t grid = (a.n + plan::NB - 1u) / plan::NB;
    const size_t lds = (size_t)plan::NB * plan::RING_STRIDE;
    auto kern = plan::lz4_replay_kernel;
    static unsigned long long have = 0ull;   // the attribute is per device (benign race: setting it twice is harmless)
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(have & bit)) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        have |= bit;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, a);
    return hipGetLastError();
}

}  // namespace lz4flex_dev

// tools / tests: replay plans that were compiled elsewhere (tests/sim/plan_model.cpp) -- the replay kernel alone
extern "C" int lz4flex_debug_replay(const void* in_base, void* out_base, const void* plans, const void* words, unsigned n, void* stream) {
    lz4flex_dev::ReplayArgs a;
    a.in_base = (const uint8_t*)in_base;
    a.out_base = (uint8_t*)out_base;
    a.plans = (const lz4flex_dev::plan::BlockPlan*)plans;
    a.words = (const uint32_t*)words;
    a.n = n;
    a.max_turns = 1u << 16;
    return (int)lz4flex_dev::launch_replay(a, (hipStream_t)stream);
}
