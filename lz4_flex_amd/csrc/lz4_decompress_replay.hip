// lz4_decompress_replay.hip -- the REPLAY half of the plan / replay decoder: executes the copy plans the plan kernel
// (lz4_decompress_plan.hip) left, FOUR LANES PER BLOCK, 64 blocks per workgroup (one workgroup per CU: the batch shape
// of BASELINE configs[1], 16 384 blocks = 64 per CU).  Reference semantics: src/block/decompress.rs:334-437 (the copies of
// the decode loop); everything else of that loop -- token chain, lengths, checks -- happened in the plan kernel.
//
// A block's copies are a serial chain (a match may read what the previous sequence wrote), so this kernel's time is
// (records per block) x (time per record) whatever the batch size, and the only thing to optimise is the time per record:
//   * a record is 4 bytes and one step: 16 of them arrive as one 64-byte line per group (16 bytes per lane), a record is
//     handed to the group's lanes by a DPP quad broadcast -- no queue, no LDS traffic, no pointer chasing;
//   * the plan kernel cut every copy so that a step needs no decision: a piece never reads what it writes, never wraps
//     the ring, and 16-byte moves per lane never leave the buffers (lz4_plan_common.h);
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

// four lines of K_END: what a lane group without a block (batch tail, irregular block) replays
#define LZ4R_E4 END_REC, END_REC, END_REC, END_REC
#define LZ4R_E24 LZ4R_E4, LZ4R_E4, LZ4R_E4, LZ4R_E4, LZ4R_E4, LZ4R_E4
__device__ __attribute__((aligned(16))) uint32_t g_end_lines[(END_LINES + 1u) * LINE_WORDS] = {LZ4R_E24, LZ4R_E24, LZ4R_E24, LZ4R_E24};
#undef LZ4R_E24
#undef LZ4R_E4
static_assert(END_LINES == 3u && LINE_WORDS == 24u, "g_end_lines");
__device__ __attribute__((aligned(64))) uint8_t g_replay_pad[64];   // the sink / source of lane groups without a block

#ifndef LZ4R_GROUPS_PER_WAVE
#define LZ4R_GROUPS_PER_WAVE 16
#endif
constexpr uint32_t GPW = LZ4R_GROUPS_PER_WAVE;     // lane groups per wavefront in use (16: every lane; measured with 8 -- twice the wavefronts, half the lanes each: 1.8 times slower, a step's cost is its instructions, not its lanes)
constexpr uint32_t NB = 4u * GPW;        // blocks per workgroup (four wavefronts)
constexpr uint32_t G = 4u;               // lanes per block
constexpr uint32_t LW = LINE_WORDS / G;  // words of a line per lane
static_assert(G * LANE_B == PIECE && LINE_WORDS == G * LW && LOOKAHEAD == LINE_WORDS && LW == 6u && LOOKAHEAD % FLUSH_EVERY == 0u, "geometry");

// Memory sources are requested LOOKAHEAD steps before their use, by the lanes that need them only (a 16-byte load costs the
// CU's texture path the same whether its bytes are wanted or not, and every wavefront pays for every lane: the first
// version loaded in every lane at every step and spent a quarter of its time there).  hipcc cannot express that -- it waits
// for a conditional load where it is issued, and it counts the loads it knows about -- so the loads of this kernel's loop are
// inline assembly with hand-counted waits: "lz4r-load" under an execution mask that always contains lane 0 (an instruction
// without any active lane might not count), "lz4r-wait <registers>" = s_waitcnt vmcnt(N) with N = the marked loads issued
// since (loads and stores the compiler adds in between only make the wait stricter).  Nothing may touch the destination
// registers between load and wait, which the compiler does not know: lz4_flex_amd/build.py checks exactly that on the ISA it
// ships (check_async_loads, also run by tests/test_isa_checks.py) and builds with -DLZ4R_PLAIN_LOADS -- every lane loads,
// the compiler waits -- if a toolchain ever breaks it.
struct Slot {
    uint32_t r;     // the record
    u32x4 v;        // its bytes, if they come from memory (K_LIT, K_FAR)
};

__device__ __forceinline__ void slot_load(u32x4& dst, const uint8_t* p, uint64_t mask) {
#ifdef LZ4R_PLAIN_LOADS
    __builtin_memcpy(&dst, p, 16);
#else
    uint64_t save;
    asm volatile("s_mov_b64 %1, exec\n\ts_mov_b64 exec, %3\n\tglobal_load_dwordx4 %0, %2, off ; lz4r-load\n\ts_mov_b64 exec, %1"
                 : "+v"(dst), "=&s"(save) : "v"(p), "s"(mask) : "memory");    // "+v": the slot keeps its registers from turn to turn (no copies of registers in flight)
#endif
}
template <int N>
__device__ __forceinline__ void slot_wait(u32x4& v) {
#ifndef LZ4R_PLAIN_LOADS
    asm volatile("s_waitcnt vmcnt(%1) ; lz4r-wait %0" : "+v"(v) : "n"(N) : "memory");
#endif
}
// a line of the plan in flight: LW words per lane
struct LineFlight { u32x4 a; uint64_t b; };
__device__ __forceinline__ void line_issue(LineFlight& l, const uint32_t* p) {
#ifdef LZ4R_PLAIN_LOADS
    __builtin_memcpy(&l.a, p, 16);
    __builtin_memcpy(&l.b, p + 4, 8);
#else
    asm volatile("global_load_dwordx4 %0, %2, off ; lz4r-load\n\tglobal_load_dwordx2 %1, %2, off offset:16 ; lz4r-load"
                 : "+v"(l.a), "+v"(l.b) : "v"(p) : "memory");
#endif
}
template <int N>
__device__ __forceinline__ void line_wait(LineFlight& l) {
#ifndef LZ4R_PLAIN_LOADS
    asm volatile("s_waitcnt vmcnt(%2) ; lz4r-wait %0 %1" : "+v"(l.a), "+v"(l.b) : "n"(N) : "memory");
#endif
}

template <uint32_t K>
__device__ __forceinline__ uint32_t quad_bcast(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, (int)(K | (K << 2) | (K << 4) | (K << 6)), 0xF, 0xF, true);
}

struct Lane {
    const uint8_t* in_l;     // the compressed block + 16 g
    const uint8_t* out_rd;   // the block's sink + 16 g (far sources)
    uint8_t* out_wr;         // the block's sink + 16 g
    lds_u8* ring_l;          // the block's ring + 16 g
    uint32_t g16;            // 16 g
#ifdef LZ4R_PLAIN_LOADS
    const uint8_t* idle;
#endif
    uint32_t op;             // output position
    uint32_t F;              // lines below F are in memory (a multiple of 64)

    // request the bytes of record r, in the lanes that will move them (an instruction whose mask is empty still counts)
    __device__ __forceinline__ void front(Slot& s, uint32_t r) {
        const uint32_t n = r >> N_SHIFT;
        const uint32_t kind = (r >> KIND_SHIFT) & 3u;
#ifdef LZ4R_EXP_FARNEAR     // timing experiment (wrong output): far sources at most 4 KiB behind the block's start
        const uint32_t field = kind == K_FAR ? (r & 0xFFFu) : (r & MAX_FIELD);
#else
        const uint32_t field = r & MAX_FIELD;
#endif
        const uint8_t* p = (kind == K_FAR ? out_rd : in_l) + field;
#ifdef LZ4R_PLAIN_LOADS
        p = (kind - 1u) < 2u ? (g16 < n ? p : p - g16) : idle;      // every lane loads: lanes the piece does not reach read where its first lane reads, records without a source their part of the plan line that was fetched last
        __builtin_memcpy(&s.v, p, 16);
#elif defined(LZ4R_EXP_NOLOAD)      // timing experiments only (wrong output): no memory sources
        asm volatile("" :: "v"(p));
        s.v = u32x4{r, r, r, r};
#else
        slot_load(s.v, p, __builtin_amdgcn_ballot_w64((kind - 1u) < 2u) & __builtin_amdgcn_ballot_w64(g16 < n));
#endif
        s.r = r;
    }
    // execute a record: near sources come from the ring, the bytes go to the ring
    __device__ __forceinline__ void back(Slot& s) {
        const uint32_t r = s.r;
        const uint32_t n = r >> N_SHIFT;
        const bool active = g16 < n;                 // (K_END: n = 0)
        slot_wait<LOOKAHEAD + 1>(s.v);               // marked loads since this slot's: the other LOOKAHEAD - 1 slots and the two halves of a line
        u32x4 x = s.v;
#ifndef LZ4R_EXP_NORING
        if (active && (r & KIND_MASK) == 0u) __builtin_memcpy(&x, (const void*)(ring_l + (r & MASK)), 16);
        if (active) __builtin_memcpy((void*)(ring_l + (op & MASK)), &x, 16);
#else
        asm volatile("" :: "v"(x));
#endif
        op += n;
    }
    // every FLUSH_EVERY steps: the 64-byte lines of the output that are complete leave the ring, 16 aligned bytes per lane -- one
    // full-line write per 64 bytes of output (the first version stored every piece where it ended: the memory system counts
    // transactions, and so does the wavefront: a store instruction costs it as much as a load)
    __device__ __forceinline__ void flush() {
        while (F + PIECE <= op) {
#ifndef LZ4R_EXP_NORING
            const u32x4 y = *reinterpret_cast<const u32x4 __attribute__((address_space(3)))*>(ring_l + (F & MASK));
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
    L.in_l = (live ? a.in_base + bp.in_off : g_replay_pad) + L.g16;
    L.out_rd = (live ? a.out_base + bp.out_off : g_replay_pad) + L.g16;
    L.out_wr = (live ? a.out_base + bp.out_off : g_replay_pad) + L.g16;
    L.ring_l = lds + j * RING_STRIDE + L.g16;
    L.op = 0u; L.F = 0u;
    // the plan: LW words per lane and line
    const uint32_t* lp = (live ? a.words + bp.first_word : g_end_lines) + LW * g;
    Slot sl[LOOKAHEAD];
    for (uint32_t i = 0; i < LOOKAHEAD; ++i) sl[i].v = u32x4{0u, 0u, 0u, 0u};
    // two lines of the plan in registers: the one whose records are being looked at, and the one in flight behind it.  They
    // swap roles every LOOKAHEAD steps, in two copies of the loop body: a single set of registers would have to be copied
    // when a line lands, and the compiler is free to place that copy before the wait.
    LineFlight fa, fb;
    fa.a = u32x4{0u, 0u, 0u, 0u}; fa.b = 0ull;
    fb.a = u32x4{0u, 0u, 0u, 0u}; fb.b = 0ull;
#define LZ4R_WORD(f, k) ((k) == 0 ? f.a.x : (k) == 1 ? f.a.y : (k) == 2 ? f.a.z : (k) == 3 ? f.a.w : (k) == 4 ? (uint32_t)f.b : (uint32_t)(f.b >> 32))
#define LZ4R_FRONT(f, i) L.front(sl[i], quad_bcast<(i) / LW>(LZ4R_WORD(f, (i) % LW)));
#define LZ4R_ALL(M, f) M(f, 0) M(f, 1) M(f, 2) M(f, 3) M(f, 4) M(f, 5) M(f, 6) M(f, 7) M(f, 8) M(f, 9) M(f, 10) M(f, 11) M(f, 12) M(f, 13) M(f, 14) \
                       M(f, 15) M(f, 16) M(f, 17) M(f, 18) M(f, 19) M(f, 20) M(f, 21) M(f, 22) M(f, 23)
#define LZ4R_STEP(f, i) L.back(sl[i]); if ((i) % FLUSH_EVERY == FLUSH_EVERY - 1u) L.flush(); LZ4R_FRONT(f, i)
    // a turn: `cur` (the line behind the one being executed) has had LOOKAHEAD steps to arrive; `nxt` (the one behind that)
    // starts now (a finished block stays inside its K_END lines)
#define LZ4R_TURN(cur, nxt)                                             \
    line_wait<LOOKAHEAD>(cur);                                          \
    LZ4R_IDLE()                                                         \
    lp += done ? 0u : LINE_WORDS;                                       \
    line_issue(nxt, lp);                                                \
    LZ4R_ALL(LZ4R_STEP, cur)                                            \
    done = (sl[LOOKAHEAD - 1u].r & KIND_MASK) == KIND_MASK;   /* the line just requested ends in K_END: the block's last record is behind us */ \
    if (__all(done_exec)) break;                                        \
    done_exec = done;
#ifdef LZ4R_PLAIN_LOADS
#define LZ4R_IDLE() L.idle = (const uint8_t*)lp;
#else
#define LZ4R_IDLE()
#endif
    bool done = false, done_exec = false;
    line_issue(fa, lp);
    line_wait<0>(fa);
    LZ4R_IDLE()
    lp += LINE_WORDS;
    line_issue(fb, lp);                                  // the second line: on its way while the first one's sources are requested
    LZ4R_ALL(LZ4R_FRONT, fa)                             // prologue: request the bytes of the first line's records
    for (;;) {
        LZ4R_TURN(fb, fa)
        LZ4R_TURN(fa, fb)
    }
#ifndef LZ4R_PLAIN_LOADS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // requests still in flight own their registers until they land
#endif
#undef LZ4R_TURN
#undef LZ4R_IDLE
#undef LZ4R_STEP
#undef LZ4R_ALL
#undef LZ4R_FRONT
#undef LZ4R_WORD
    // the bytes behind the last full line leave the ring, then the tail: the block's last few pieces, byte by byte in memory
    // (exact reads, exact writes)
    if (live && g == 0u) {
        const uint8_t* in = a.in_base + bp.in_off;
        uint8_t* out = a.out_base + bp.out_off;
        const lds_u8* ring = lds + j * RING_STRIDE;
        for (uint32_t k = L.F; k < L.op; ++k) out[k] = ring[k & MASK];
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
    return (int)lz4flex_dev::launch_replay(a, (hipStream_t)stream);
}
