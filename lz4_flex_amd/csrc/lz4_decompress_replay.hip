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
#define LZ4R_E28 LZ4R_E4, LZ4R_E4, LZ4R_E4, LZ4R_E4, LZ4R_E4, LZ4R_E4, LZ4R_E4
__device__ __attribute__((aligned(16))) uint32_t g_end_lines[(END_LINES + 1u) * LINE_WORDS] = {LZ4R_E28, LZ4R_E28, LZ4R_E28, LZ4R_E28};
#undef LZ4R_E28
#undef LZ4R_E4
static_assert(END_LINES == 3u && LINE_WORDS == 28u, "g_end_lines");
__device__ __attribute__((aligned(64))) uint8_t g_replay_pad[64];   // the sink / source of lane groups without a block

#ifndef LZ4R_GROUPS_PER_WAVE
#define LZ4R_GROUPS_PER_WAVE 16
#endif
constexpr uint32_t GPW = LZ4R_GROUPS_PER_WAVE;     // lane groups per wavefront in use (16: every lane; 8: half of them -- twice the wavefronts per block, the other SIMD-resident wavefront runs while one waits)
constexpr uint32_t NB = 4u * GPW;        // blocks per workgroup (four wavefronts)
constexpr uint32_t G = 4u;               // lanes per block
constexpr uint32_t LW = LINE_WORDS / G;  // words of a line per lane
static_assert(G * LANE_B == PIECE && LINE_WORDS == G * LW && LOOKAHEAD == LINE_WORDS && LW == 7u, "geometry");

struct Slot {
    uint32_t r;     // the record
    u32x4 v;        // its bytes, if they come from memory (K_LIT, K_FAR)
};
struct LineRegs { uint32_t w[LW]; };

template <uint32_t K>
__device__ __forceinline__ uint32_t quad_bcast(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, (int)(K | (K << 2) | (K << 4) | (K << 6)), 0xF, 0xF, true);
}

struct Lane {
    const uint8_t* in_l;     // the compressed block
    const uint8_t* idle;     // what a lane without a source reads: its own 16 bytes of the plan line that is being fetched anyway
    const uint8_t* out_rd;   // the block's sink (far sources)
    uint8_t* out_wr;         // the block's sink + 16 g
    lds_u8* ring_l;          // the block's ring + 16 g
    uint32_t g16;            // 16 g
    uint32_t g16d;           // 16 g until the block's K_END, then 0xFFFFFFFF: `g16d < n` is "this lane moves bytes of the piece"
    uint32_t op;             // output position
    uint32_t F;              // lines below F have left the ring (a multiple of 64)
    uint32_t pF;             // a line that was read from the ring in the previous step and goes to memory in this one
    uint32_t pend;
    u32x4 y;

    // request the bytes of record r (if they come from memory).  Lanes the piece does not reach read where the piece's first
    // lane reads, records without a load read the lane's part of the plan line in flight: neither costs a memory transaction.
    // (One 64-byte pad for every idle lane of the chip was 2.3 times slower than the whole kernel: same bank, same channel.)
    __device__ __forceinline__ void front(Slot& s, uint32_t r) {
        const uint32_t kind = r >> 30;
        const uint32_t n = ((r >> 24) & 63u) + 1u;
#ifdef LZ4R_EXP_FARNEAR     // timing experiment (wrong output): far sources at most 4 KiB behind the block's start
        const uint32_t field = kind == K_FAR ? (r & 0xFFFu) : (r & 0xFFFFFFu);
#else
        const uint32_t field = r & 0xFFFFFFu;
#endif
        const uint32_t lane_off = g16 < n ? g16 : 0u;
        const uint8_t* p = kind == K_FAR ? out_rd : in_l;
        p = (kind - 1u) < 2u ? p + (field + lane_off) : idle;
#ifdef LZ4R_EXP_NOLOAD      // timing experiments only (wrong output): no memory sources
        asm volatile("" :: "v"(p));
        s.v = u32x4{r, r, r, r};
#else
        __builtin_memcpy(&s.v, p, 16);
#endif
        s.r = r;
    }
    // execute a record: the read half (near sources come from the ring) ...
    __device__ __forceinline__ void back_read(const Slot& s, u32x4& x, uint32_t& n, bool& active) {
        const uint32_t r = s.r;
        n = r >= END_REC ? 0u : ((r >> 24) & 63u) + 1u;
        g16d = r >= END_REC ? 0xFFFFFFFFu : g16d;
        active = g16d < n;
        x = s.v;
#ifndef LZ4R_EXP_NORING
        if (active && r < (K_LIT << 30)) __builtin_memcpy(&x, (const void*)(ring_l + (r & MASK)), 16);
#endif
    }
    // ... the line the previous step took out of the ring goes to memory (one full-line write per 64 bytes of output instead
    // of one partial write per piece: the memory system counts transactions, not bytes) ...
    __device__ __forceinline__ void store_pending() {
#ifndef LZ4R_EXP_NOSTORE
        if (pend) __builtin_memcpy(out_wr + pF, &y, 16);
#endif
    }
    // ... and the write half: the bytes go to the ring; a 64-byte line this piece completed is read back, 16 aligned bytes per
    // lane, for the next step's store
    __device__ __forceinline__ void back_write(const u32x4& x, uint32_t n, bool active) {
#ifndef LZ4R_EXP_NORING
        if (active) __builtin_memcpy((void*)(ring_l + (op & MASK)), &x, 16);
#endif
        op += n;
        const uint32_t fl = op & ~(PIECE - 1u);
        pend = fl != F;              // pieces are at most PIECE bytes: exactly one line, [F, F + 64)
        if (pend) {
#ifndef LZ4R_EXP_NORING
            y = *reinterpret_cast<const u32x4 __attribute__((address_space(3)))*>(ring_l + (F & MASK));
#else
            y = x;
#endif
            pF = F;
            F = fl;
        }
    }
};

__device__ __forceinline__ void load_line(LineRegs& l, const uint32_t* p) { __builtin_memcpy(l.w, p, 4u * LW); }

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
    L.g16d = L.g16;
    L.in_l = live ? a.in_base + bp.in_off : g_replay_pad;
    L.out_rd = live ? a.out_base + bp.out_off : g_replay_pad;
    L.out_wr = (live ? a.out_base + bp.out_off : g_replay_pad) + L.g16;
    L.ring_l = lds + j * RING_STRIDE + L.g16;
    L.op = 0u; L.F = 0u; L.pF = 0u; L.pend = 0u; L.y = u32x4{0u, 0u, 0u, 0u};
    // the plan: LW words per lane and line
    const uint32_t* lp = (live ? a.words + bp.first_word : g_end_lines) + LW * g;
    Slot sl[LOOKAHEAD];
    LineRegs line, next;
    load_line(line, lp);
    load_line(next, lp + LINE_WORDS);
    L.idle = (const uint8_t*)lp;
    lp += 2u * LINE_WORDS;
    // prologue: request the bytes of the first line's records
#define LZ4R_FRONT(i) L.front(sl[i], quad_bcast<(i) / LW>(line.w[(i) % LW]))
#define LZ4R_ALL(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15) M(16) M(17) M(18) M(19) M(20) \
                    M(21) M(22) M(23) M(24) M(25) M(26) M(27)
#define LZ4R_PRO(i) LZ4R_FRONT(i);
    LZ4R_ALL(LZ4R_PRO)
    for (;;) {
        line = next;                             // the line behind the one being executed
        load_line(next, lp);                     // and the one behind that: on its way for LOOKAHEAD steps
        L.idle = (const uint8_t*)lp;
        lp += L.g16d == 0xFFFFFFFFu ? 0u : LINE_WORDS;   // (a finished block stays inside its K_END lines)
#define LZ4R_STEP(i) { u32x4 x; uint32_t n; bool act; L.back_read(sl[i], x, n, act); LZ4R_FRONT(i); L.store_pending(); L.back_write(x, n, act); }
        LZ4R_ALL(LZ4R_STEP)
        if (__all(L.g16d == 0xFFFFFFFFu)) break;
    }
    L.store_pending();
#undef LZ4R_STEP
#undef LZ4R_PRO
#undef LZ4R_ALL
#undef LZ4R_FRONT
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
