// capi.cpp -- host side of the C ABI declared in include/lz4flex_amd.h: context / device
// workspace management, the batched entry points (host- or device-resident buffers) and the
// lz4_flex-shaped scalar calls, which are 1-block batches through the same HIP kernels.
// There is no CPU codec in this library: every call that produces bytes launches a kernel.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/lz4flex_amd.h"
#include "lz4_device.h"
#include "lz4_plan_common.h"

using namespace lz4flex_dev;

static thread_local std::string g_last_error;
static thread_local int g_last_hip = 0;

static int hip_fail(hipError_t e, const char* what) {
    g_last_hip = (int)e;
    g_last_error = std::string(what) + ": " + hipGetErrorString(e);
    return -LZ4FLEX_E_HIP;
}
#define HIP_TRY(expr)                                   \
    do {                                                \
        hipError_t _e = (expr);                         \
        if (_e != hipSuccess) return hip_fail(_e, #expr); \
    } while (0)

// compress_variant -> launch_compress mode bits: 1 = encode_block + emitter wave (default), 3 = encode_block alone
static inline int comp_mode_bits(int v) { return v == 1 ? 0x600 : 0; }

struct lz4flex_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    uint8_t* d_arena = nullptr;   // device staging for MEM_HOST calls
    size_t arena_cap = 0;
    uint8_t* h_pin = nullptr;     // pinned host staging for descriptor / result arrays
    size_t pin_cap = 0;
    uint8_t* h_pay = nullptr;     // pinned host staging for compacted results of MEM_HOST compress batches
    size_t pay_cap = 0;
    int dec_lanes = 16;           // lanes per block, decode
    int comp_lanes = 8;           // lanes per block, encode
    int comp_mode = 0;            // 0 = throughput ("wave") encoder, own parse (default); 1 = reference-exact encoder (lz4_flex's bytes)
    int comp_variant = 1;         // reference-exact encoder: 1 = group encoder + emitter wave (default), 3 = group encoder alone
    uint8_t* pcd_ws = nullptr;    // workgroup decoder, small batches: a parser and a copier workgroup per block hand token lists over through this (lz4_device.h pair_ws); one per context, ordered across streams like wave_ws
    hipEvent_t pcd_done = nullptr;
    hipStream_t pcd_last = nullptr;
    bool pcd_used = false;
    int dec_pcd_pair = 1;         // 1: two workgroups per block for batches of few large blocks; 0: never, 2: whenever the batch is small enough (tests, measurements)
    uint32_t* chain_ws = nullptr; // chained decode batches (Linked frames): one "done" word per block, CHAIN_WS_BLOCKS of them; one per context, ordered across streams like wave_ws
    hipEvent_t chain_evt = nullptr;
    hipStream_t chain_last = nullptr;
    bool chain_used = false;
    void* wave_ws = nullptr;      // wave encoder workspace: wave_wgs persistent workgroups; allocated by lz4flex_ctx_create
    unsigned long long* wave_prof = nullptr;   // tools: per-role cycle counters of the wave encoder (lz4flex_debug_wave_prof)
    int wave_wgs = 0;
    // The workspace is ONE per context, and MEM_DEVICE batches are enqueued on the caller's stream: two batches on different
    // streams could overlap on the GPU and race on it.  Every wave launch records wave_done; a launch on another stream than
    // the previous one first makes its stream wait for it (same stream: already ordered).
    hipEvent_t wave_done = nullptr;
    hipStream_t wave_last = nullptr;
    bool wave_used = false;
    int dec_blocks_per_wg = 0;    // split decoder: blocks per workgroup (8/16/32/64), 0 = 64
    int comp_det = 0;             // "compress_deterministic": 1 = a block's bytes depend on the block and the settings alone (no sub-windows by batch size)
    int dec_level_chains = 1024;  // "decompress_level_chains": from this many Linked streams in one *_many call on, block k of every stream is one plain launch (frame_many.cpp); 0 = never
    int comp_sub = 0;             // throughput encoder, "compress_subwindows": 0 = by batch size, 1 = never, 2 / 4 = always that many sub-windows per block of <= 64 KiB
    int dec_variant = 0;          // 0 = by batch size, 1 = window in HBM/L2 (lz4_decompress.hip), 4 = parser / copier split (lz4_decompress_split.hip), 7 = a workgroup per block (lz4_decompress_pcd.hip; 8: its test geometry; 10 / 11: 256 / 512 lanes), 13 = a wavefront per block, a lane per sequence (lz4_decompress_seq.hip); tools builds: 9 = plan / replay, 12 = parser / emitter / quads
    int comp_sliding = 2;         // throughput encoder: the windows of a block longer than 64 KiB advance by 48 KiB (2: every window start has 16 KiB of history) or 32 KiB (1: round 4's bytes); 0 = by 64 KiB (round 3's bytes, fastest)
    int comp_carry_wait = 1;      // tests: 0 = a window of the throughput encoder that has to wait for its predecessor's carry gives up at once (the block then takes the second launch)
    int dec_second_pass = 1;      // tests: 0 leaves the blocks a first-pass decoder marked (status 0x7F000001) instead of decoding them again
    // plan / replay decoder (lz4_decompress_plan.hip, lz4_decompress_replay.hip): the copy plans of a batch, plan_slot_words() words per
    // block + a 32-byte header each.  Grows with the largest batch seen (a hipMalloc -- a device synchronisation -- in the first such
    // call and whenever a larger batch arrives; never shrinks); one per context, ordered across streams like wave_ws.
    uint8_t* plan_ws = nullptr;
    size_t plan_cap = 0;
    hipEvent_t plan_done = nullptr;
    hipStream_t plan_last = nullptr;
    bool plan_used = false;
    int chain_giveup = 0;         // tests: block chain_giveup - 1 of the next chained decode batches gives up without an error (the ordered second pass decodes it and everything behind it)
    // many frames at once (frame_many.cpp): device scratch of the calls that run to completion before they return (compressed staging,
    // descriptor arrays, staged host buffers).  Grow-only, a hipMalloc when a larger job arrives; freed with the context.
    void* many_ws[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t many_cap[4] = {0, 0, 0, 0};
    int fail_next_batch = 0;      // tests: the next N batch calls on this context fail before they launch anything (what an allocation failure looks like to the caller)
};

static constexpr uint32_t PCD_MAX_BLOCKS = DISPATCH_PCD_256;     // up to here the default dispatch takes the workgroup decoder
static constexpr uint32_t CHAIN_WS_BLOCKS = 65536u;      // blocks per chained decode batch (LZ4FLEX_MEM_CHAINED)

// the decoders for blocks without dictionary / prefix
// The "done" words of a chained batch are the context's: two chained batches on different streams would clear and poll the same
// flags -- a launch on another stream than the previous one waits for it first, every launch records the event (ADVICE r3 for the
// device path, r4 for the host path: ONE helper for both)
static hipError_t chain_ws_begin(lz4flex_ctx* c, uint32_t n, hipStream_t s) {
    if (c->chain_used && s != c->chain_last) { const hipError_t w = hipStreamWaitEvent(s, c->chain_evt, 0); if (w != hipSuccess) return w; }
    // (marked used BEFORE the kernels: a call that fails between here and its record still leaves work of this stream on the words)
    c->chain_last = s; c->chain_used = true;
    const hipError_t m = hipMemsetAsync(c->chain_ws, 0, 4ull * n, s);
    if (m != hipSuccess) return m;
    return hipEventRecord(c->chain_evt, s);
}
static hipError_t chain_ws_end(lz4flex_ctx* c, hipStream_t s) { return hipEventRecord(c->chain_evt, s); }

static hipError_t launch_decompress_fast(lz4flex_ctx* c, const DecompressArgs& a_, hipStream_t s, bool big_blocks = false) {
    DecompressArgs a = a_;
    // 0: by batch shape (lz4_device.h DISPATCH_*; tools/dec_shapes.py measures every decoder on every shape, profiles/r06_decoder_shapes.txt):
    //   * up to DISPATCH_PCD_256 blocks, or blocks known to be large: a whole WORKGROUP per block, token chain and copies parallel INSIDE the
    //     block (lz4_decompress_pcd.hip) -- the decoder whose time for a block is not the length of the block's chain: 256 / 512 JSON blocks
    //     0.14 / 0.23 ms, 256 x 4 MiB log blocks 4.7 ms, one 16 MiB block 14.6 ms;
    //   * up to DISPATCH_SEQ_MAX blocks: a WAVEFRONT per block, a lane per sequence (lz4_decompress_seq.hip, round 6: it replaced the wave
    //     decoder and its two-wavefront form, 0.74 / 1.01 ms for 1 024 / 4 096 JSON blocks): 768 / 2 304 / 4 096 / 8 192 / 12 288 JSON blocks
    //     0.32 / 0.41 / 0.53 / 1.00 / 1.44 ms, 4 096 text / log / incompressible blocks 0.88 / 0.49 / 0.30 ms;
    //   * larger batches: a LANE per block parses, four lanes copy (lz4_decompress_split.hip): 1.63 - 1.67 ms for any batch of 5 121 ...
    //     16 384 JSON blocks -- the chain of one block.
    int v = c->dec_variant != 0 ? c->dec_variant : ((a.n <= PCD_MAX_BLOCKS || big_blocks) ? 7 : (a.n <= DISPATCH_SEQ_MAX ? 13 : 4));
    const int geo_req = v == 10 ? 2 : (v == 11 ? 3 : 0);  // (an explicit geometry holds for prefix / chained batches too)
    // prefix mode (Linked frames): the workgroup decoder, or -- a batch whose prefixes are in memory already, not a CHAINED one -- the sequence decoder (round 6)
    if (a.out_pos != nullptr && v != 8 && !(v == 13 && a.chain_done == nullptr)) v = 7;
    if ((v == 12 || v == 13) && a.dict_base != nullptr) v = 4;
    // the workgroup decoder's geometry by batch size (lz4_decompress_pcd.hip GeoMid*: smaller workgroups, more of them per CU); large
    // blocks and chains keep the full workgroup (a chain is one block at a time, a large block wants the long tiles)
    int pcd_geo = v == 8 ? 1 : geo_req;
    if (c->dec_variant == 0 && v == 7 && !big_blocks && a.out_pos == nullptr && a.n > DISPATCH_PCD_1024) pcd_geo = a.n <= DISPATCH_PCD_512 ? 3 : 2;
    // several chains side by side (chain_prev): one block per chain is runnable at a time, so the CHAINS are what fills the chip --
    // tools/many_probe.py, 1 GiB of Linked JSON frames of 64 KiB blocks, 1 024 / 256 lanes per block: 256 chains 8.3 / 16.8 ms,
    // 1 024 chains 8.3 / 6.1, 4 096 chains 9.0 / 6.7, 64 chains 20.5 / 63
    if (c->dec_variant == 0 && v == 7 && a.chain_prev != nullptr && a.n_chains > DISPATCH_PCD_1024) pcd_geo = a.n_chains <= DISPATCH_PCD_512 ? 3 : 2;
#ifdef LZ4FLEX_TOOLS
    if (v == 9 && (uint64_t)a.n * plan_slot_words() > 0xFFFFFFFFull) v = 4;     // BlockPlan's word indices are 32-bit: such a batch takes the split decoder
    if (v == 9) {
        // plan / replay: every block is turned into a copy plan (one wavefront per block, everything that is parallel), then the
        // plans are replayed (four lanes per block, the serial rest); blocks without a plan (errors, sinks too small, oversized) go
        // to the reference-order kernel
        constexpr int32_t REDO = 0x7F000001;
        const size_t slot = plan_slot_words();
        const size_t need = (size_t)a.n * (slot * 4u + 32u) + 256u;
        if (need > c->plan_cap) {
            if (c->plan_used) { const hipError_t w = hipEventSynchronize(c->plan_done); if (w != hipSuccess) return w; }
            if (c->plan_ws) (void)hipFree(c->plan_ws);
            c->plan_ws = nullptr; c->plan_cap = 0;
            const hipError_t m = hipMalloc((void**)&c->plan_ws, need);
            if (m != hipSuccess) return m;
            c->plan_cap = need;
        }
        if (c->plan_used && s != c->plan_last) { const hipError_t w = hipStreamWaitEvent(s, c->plan_done, 0); if (w != hipSuccess) return w; }
        PlanArgs pa;
        pa.in_base = a.in_base; pa.in_off = a.in_off; pa.in_len = a.in_len; pa.out_off = a.out_off; pa.out_cap = a.out_cap;
        pa.plans = (plan::BlockPlan*)c->plan_ws;
        pa.words = (uint32_t*)(c->plan_ws + (((size_t)a.n * 32u + 255u) & ~(size_t)255u));
        pa.out_len = a.out_len; pa.status = a.status; pa.n = a.n; pa.slot_words = (uint32_t)slot; pa.redo_code = REDO;
        hipError_t e = launch_plan(pa, s);
        if (e != hipSuccess) return e;
        ReplayArgs ra;
        ra.in_base = a.in_base; ra.out_base = a.out_base; ra.plans = pa.plans; ra.words = pa.words; ra.n = a.n; ra.max_turns = (uint32_t)(slot / plan::TURN_WORDS) + 2u;
        e = launch_replay(ra, s);
        if (e != hipSuccess) return e;
        e = hipEventRecord(c->plan_done, s);
        if (e != hipSuccess) return e;
        c->plan_last = s; c->plan_used = true;
        if (a.detail) { const hipError_t z = hipMemsetAsync(a.detail, 0, 16ull * a.n, s); if (z != hipSuccess) return z; }
        if (!c->dec_second_pass) return hipSuccess;
        DecompressArgs r = a;
        r.only_status = REDO;
        return launch_decompress(r, c->dec_lanes, s);
    }
#endif
    if (v == 7 || v == 8 || v == 10 || v == 11) {
        // one block per workgroup; blocks it marks (errors, sinks too small) are decoded again in the reference's order
        constexpr int32_t REDO = 0x7F000001;
        bool pair = false;
        if (v >= 7 && v <= 8 && (c->dec_pcd_pair == 2 || (c->dec_pcd_pair == 1 && big_blocks)) && c->pcd_ws && a.n <= PCD_PAIR_MAX_BLOCKS) {
            // few LARGE blocks: a parser and a copier workgroup each (half the CUs would idle otherwise; a block of one tile has
            // nothing to overlap and would only pay the hand-over: 64 KiB JSON blocks 0.144 -> 0.156 ms).  2 = always (tests).  The hand-over workspace is the
            // context's: launches on different streams are ordered by an event, as the encoder's are
            if (c->pcd_used && s != c->pcd_last) { const hipError_t w = hipStreamWaitEvent(s, c->pcd_done, 0); if (w != hipSuccess) return w; }
            const hipError_t m = hipMemsetAsync(c->pcd_ws, 0, 64ull * a.n, s);
            if (m != hipSuccess) return m;
            a.pair_ws = c->pcd_ws;
            pair = true;
        }
        hipError_t e = launch_decompress_pcd(a, REDO, s, pcd_geo);
        if (e != hipSuccess) return e;
        if (pair) {
            e = hipEventRecord(c->pcd_done, s);
            if (e != hipSuccess) return e;
            c->pcd_last = s; c->pcd_used = true;
            a.pair_ws = nullptr;
        }
        if (!c->dec_second_pass) return hipSuccess;
        DecompressArgs r = a;
        r.only_status = REDO;
        // a chained batch's marked blocks depend on each other: in chain order, not side by side
        return r.chain_done ? launch_decompress_chain_redo(r, s) : launch_decompress(r, c->dec_lanes, s);
    }
    if (v == 13) {
        // a wavefront per block, a lane per sequence (lz4_decompress_seq.hip); irregular blocks go to the reference-order kernel
        constexpr int32_t REDO = 0x7F000001;
        const hipError_t e = launch_decompress_seq(a, REDO, s);
        if (e != hipSuccess) return e;
        if (!c->dec_second_pass) return hipSuccess;
        DecompressArgs r = a;
        r.only_status = REDO;
        return launch_decompress(r, c->dec_lanes, s);
    }
#ifdef LZ4FLEX_TOOLS
    if (v == 12) {
        // parser -> emitter -> quads (lz4_decompress_fused.hip); oversized blocks go to the reference-order kernel
        constexpr int32_t REDO = 0x7F000001;
        const hipError_t e = launch_decompress_fused(a, REDO, s);
        if (e != hipSuccess) return e;
        if (!c->dec_second_pass) return hipSuccess;
        DecompressArgs r = a;
        r.only_status = REDO;
        return launch_decompress(r, c->dec_lanes, s);
    }
#endif
    return launch_decompress_split(a, s, c->dec_blocks_per_wg);
}

// Reference-exact encoder, MEM_DEVICE batches without LZ4FLEX_MEM_BIG_BLOCKS: the u16-table kernel is only right for blocks
// of <= 64 KiB and the host cannot see the device-resident lengths, so a block that breaks the promise is flagged on the
// device (its bytes would be a valid block, but not lz4_flex's) instead of passing silently.
__global__ void lz4flex_flag_long_blocks_kernel(const uint32_t* in_len, uint32_t n, uint32_t* out_len, int32_t* status) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < n && in_len[b] > 65536u) { status[b] = LZ4FLEX_E_INVALID_ARG; out_len[b] = 0u; }
}

// the encoders for independent blocks: throughput mode (own parse, any block length) or the reference-exact one
static int launch_compress_any(lz4flex_ctx* c, const CompressArgs& a, bool big, hipStream_t s) {
    hipError_t le;
    if (c->comp_mode == 0) {
        if (!c->wave_ws || !c->wave_done) { g_last_error = "context without encoder workspace"; return -LZ4FLEX_E_INVALID_ARG; }
        if (c->wave_used && s != c->wave_last) HIP_TRY(hipStreamWaitEvent(s, c->wave_done, 0));
        CompressArgs aw = a;
        aw.slide = c->comp_sliding == 2 ? 49152u : (c->comp_sliding == 1 ? 32768u : 0u);
        // sub-windows (lz4_compress_wave.hip Item::sub): batches that leave at least half / three quarters of the persistent workgroups
        // without a block cut their blocks of <= 64 KiB into 2 / 3 / 4 items each
        aw.sub = c->comp_det ? 1u : c->comp_sub != 0 ? (uint32_t)c->comp_sub
                                  : (a.n * 4u <= (uint32_t)c->wave_wgs ? 4u : (a.n * 3u <= (uint32_t)c->wave_wgs ? 3u : (a.n * 2u <= (uint32_t)c->wave_wgs ? 2u : 1u)));
        le = launch_compress_wave(aw, c->wave_ws, c->wave_wgs, s, c->wave_prof, c->comp_carry_wait != 0);
        if (le == hipSuccess) {
            HIP_TRY(hipEventRecord(c->wave_done, s));
            c->wave_last = s;
            c->wave_used = true;
        }
    } else {
        le = launch_compress(a, c->comp_lanes | (big ? 0x100 : 0) | comp_mode_bits(c->comp_variant), s);
        if (le == hipSuccess && !big && a.n) {
            hipLaunchKernelGGL(lz4flex_flag_long_blocks_kernel, dim3((a.n + 255u) / 256u), dim3(256), 0, s, a.in_len, a.n, a.out_len, a.status);
            le = hipGetLastError();
        }
    }
    if (le != hipSuccess) return hip_fail(le, "kernel launch");
    return 0;
}

static int ensure_arena(lz4flex_ctx* c, size_t need) {
    if (need <= c->arena_cap) return 0;
    if (c->d_arena) { (void)hipFree(c->d_arena); c->d_arena = nullptr; c->arena_cap = 0; }
    size_t cap = std::max<size_t>(need + need / 4, 1u << 20);
    HIP_TRY(hipMalloc((void**)&c->d_arena, cap));
    c->arena_cap = cap;
    return 0;
}
static int ensure_pin(lz4flex_ctx* c, size_t need) {
    if (need <= c->pin_cap) return 0;
    if (c->h_pin) { (void)hipHostFree(c->h_pin); c->h_pin = nullptr; c->pin_cap = 0; }
    size_t cap = std::max<size_t>(need + need / 4, 1u << 16);
    HIP_TRY(hipHostMalloc((void**)&c->h_pin, cap, hipHostMallocDefault));
    c->pin_cap = cap;
    return 0;
}

static int ensure_pay(lz4flex_ctx* c, size_t need) {
    if (need <= c->pay_cap) return 0;
    if (c->h_pay) { (void)hipHostFree(c->h_pay); c->h_pay = nullptr; c->pay_cap = 0; }
    size_t cap = std::max<size_t>(need + need / 4, 1u << 20);
    HIP_TRY(hipHostMalloc((void**)&c->h_pay, cap, hipHostMallocDefault));
    c->pay_cap = cap;
    return 0;
}

// MEM_HOST batches whose results are sparse in the output span (compress: ~15 KB used of every 72 KB stride):
// pack the produced bytes densely on the device so that ONE transfer brings them to the host.
__global__ void __launch_bounds__(256) lz4flex_pack_results_kernel(const uint8_t* src, const uint64_t* src_off, const uint32_t* len,
                                                                   const int32_t* status, const uint64_t* dst_off, uint8_t* dst,
                                                                   uint32_t n) {
    const uint32_t b = blockIdx.x;
    if (b >= n || status[b] != 0) return;
    const uint8_t* s = src + src_off[b];
    uint8_t* d = dst + dst_off[b];
    const uint32_t m = len[b];
    for (uint32_t i = threadIdx.x; i < m; i += 256u) d[i] = s[i];
}

static int default_ctx(lz4flex_ctx** out);

extern "C" {

const char* lz4flex_version(void) { return "lz4flex-amd 0.3.0 (gfx950)"; }
#ifndef LZ4FLEX_BUILD_ID
#define LZ4FLEX_BUILD_ID "unstamped"
#endif
const char* lz4flex_build_id(void) { return LZ4FLEX_BUILD_ID; }
const char* lz4flex_last_error(void) { return g_last_error.c_str(); }

int lz4flex_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int lz4flex_ctx_create(lz4flex_ctx** out, int device) {
    if (!out) return -LZ4FLEX_E_INVALID_ARG;
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        g_last_error = "no HIP device available: the MI355X kernels are the only codec in this library";
        return -LZ4FLEX_E_NO_DEVICE;
    }
    if (device < 0) HIP_TRY(hipGetDevice(&device));
    if (device >= n) return -LZ4FLEX_E_INVALID_ARG;
    lz4flex_ctx* c = new (std::nothrow) lz4flex_ctx();
    if (!c) return -LZ4FLEX_E_NOMEM;
    c->device = device;
    if (const char* e = getenv("LZ4FLEX_COMPRESS_MODE")) c->comp_mode = (!strcmp(e, "exact") || !strcmp(e, "1")) ? 1 : 0;
    if (const char* e = getenv("LZ4FLEX_SLIDING_WINDOW")) {      // "0", "1" or "2", nothing else ("off" used to read as 0 silently: ADVICE r4)
        if (!strcmp(e, "0") || !strcmp(e, "1") || !strcmp(e, "2")) c->comp_sliding = e[0] - '0';
        else { g_last_error = "LZ4FLEX_SLIDING_WINDOW must be 0, 1 or 2"; delete c; return -LZ4FLEX_E_INVALID_ARG; }
    }
#ifdef LZ4FLEX_ALL_VARIANTS
    if (const char* e = getenv("LZ4FLEX_COMPRESS_VARIANT")) { const int v = atoi(e); if (v == 1 || v == 3) c->comp_variant = v; }
#endif
    if (const char* e = getenv("LZ4FLEX_DECOMPRESS_VARIANT")) { const int v = atoi(e); if (v == 0 || v == 1 || (v >= 4 && v <= 12 && v != 9)) c->dec_variant = v; }
    int prev = 0;
    (void)hipGetDevice(&prev);
    e = hipSetDevice(device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    // the throughput encoder's workspace (two 80 KiB workgroups per CU, 164 MiB on an MI355X) and its ordering event: here, not
    // inside the first compress call -- a hipMalloc in an "asynchronous" entry point is a device synchronisation and breaks
    // stream capture
    if (e == hipSuccess) {
        int cus = 0;
        e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
        if (e == hipSuccess) {
            c->wave_wgs = 2 * cus;
            e = hipMalloc(&c->wave_ws, compress_wave_workspace_bytes(c->wave_wgs));
        }
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->wave_done, hipEventDisableTiming);
    if (e == hipSuccess) e = hipMalloc((void**)&c->chain_ws, 4u * CHAIN_WS_BLOCKS);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->chain_evt, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->plan_done, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->pcd_done, hipEventDisableTiming);
    if (e == hipSuccess) e = hipMalloc((void**)&c->pcd_ws, decompress_pcd_pair_ws_bytes());
    (void)hipSetDevice(prev);
    if (e != hipSuccess) {
        const int rc = hip_fail(e, "ctx_create");
        lz4flex_ctx_destroy(c);
        return e == hipErrorOutOfMemory ? -LZ4FLEX_E_NOMEM : rc;
    }
    *out = c;
    return 0;
}

void lz4flex_ctx_destroy(lz4flex_ctx* c) {
    if (!c) return;
    if (c->d_arena) (void)hipFree(c->d_arena);
    if (c->wave_ws) (void)hipFree(c->wave_ws);
    if (c->wave_done) (void)hipEventDestroy(c->wave_done);
    if (c->chain_ws) (void)hipFree(c->chain_ws);
    if (c->chain_evt) (void)hipEventDestroy(c->chain_evt);
    if (c->plan_ws) (void)hipFree(c->plan_ws);
    if (c->plan_done) (void)hipEventDestroy(c->plan_done);
    if (c->pcd_ws) (void)hipFree(c->pcd_ws);
    if (c->pcd_done) (void)hipEventDestroy(c->pcd_done);
    if (c->wave_prof) (void)hipFree(c->wave_prof);
    for (void* p : c->many_ws) if (p) (void)hipFree(p);
    if (c->h_pin) (void)hipHostFree(c->h_pin);
    if (c->h_pay) (void)hipHostFree(c->h_pay);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

#ifdef LZ4FLEX_TOOLS
// variant builds for tools/ only (-DLZ4FLEX_TOOLS; not in the public header, not in the product library): enable != 0 starts /
// resets the wave encoder's per-role cycle counters of this context, vals (nullable) receives the 32 sums accumulated so far
int lz4flex_debug_wave_prof(lz4flex_ctx* c, int enable, unsigned long long* vals) {
    if (!c) return -LZ4FLEX_E_INVALID_ARG;
    if (vals && c->wave_prof) {
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(vals, c->wave_prof, 256, hipMemcpyDeviceToHost));
    }
    if (enable) {
        if (!c->wave_prof) HIP_TRY(hipMalloc((void**)&c->wave_prof, 256));
        HIP_TRY(hipMemset(c->wave_prof, 0, 256));
    } else if (c->wave_prof) {
        (void)hipFree(c->wave_prof);
        c->wave_prof = nullptr;
    }
    return 0;
}
#endif

int lz4flex_set_tuning(lz4flex_ctx* c, const char* key, int value) {
    if (!key) return -LZ4FLEX_E_INVALID_ARG;
    if (!c) { const int rc = default_ctx(&c); if (rc) return rc; }   // NULL: this thread's default context (the scalar calls)
    if (!strcmp(key, "decompress_lanes")) {
#ifdef LZ4FLEX_ALL_VARIANTS
        if (value != 8 && value != 16 && value != 32 && value != 64) return -LZ4FLEX_E_INVALID_ARG;
#else
        if (value != 16) return -LZ4FLEX_E_INVALID_ARG;          // the other widths exist in variant builds only (-DLZ4FLEX_ALL_VARIANTS)
#endif
        c->dec_lanes = value;
        return 0;
    }
    if (!strcmp(key, "compress_mode")) {
        if (value != 0 && value != 1) return -LZ4FLEX_E_INVALID_ARG;
        c->comp_mode = value;
        return 0;
    }
    if (!strcmp(key, "compress_subwindows")) {
        if (value < 0 || value > 4) return -LZ4FLEX_E_INVALID_ARG;
        c->comp_sub = value;
        return 0;
    }
    if (!strcmp(key, "compress_deterministic")) {
        if (value != 0 && value != 1) return -LZ4FLEX_E_INVALID_ARG;
        c->comp_det = value;
        return 0;
    }
    if (!strcmp(key, "decompress_level_chains")) {
        if (value < 0) return -LZ4FLEX_E_INVALID_ARG;
        c->dec_level_chains = value;
        return 0;
    }
    if (!strcmp(key, "decompress_blocks_per_wg")) {
        if (value != 0 && value != 8 && value != 16 && value != 32 && value != 64) return -LZ4FLEX_E_INVALID_ARG;
        c->dec_blocks_per_wg = value;
        return 0;
    }
    if (!strcmp(key, "decompress_variant")) {
        if (value != 0 && value != 1 && (value < 4 || value > 13)) return -LZ4FLEX_E_INVALID_ARG;
        if (value == 5 || value == 6) return -LZ4FLEX_E_INVALID_ARG;      // the wave decoder and its two-wavefront form: replaced by 13 in round 6
#ifndef LZ4FLEX_TOOLS
        if (value == 9 || value == 12) return -LZ4FLEX_E_INVALID_ARG;     // plan / replay, parser / emitter / quads: tools builds only
#endif
        c->dec_variant = value;
        return 0;
    }
    if (!strcmp(key, "decompress_second_pass")) {
        if (value != 0 && value != 1) return -LZ4FLEX_E_INVALID_ARG;
        c->dec_second_pass = value;
        return 0;
    }
    if (!strcmp(key, "decompress_pcd_pair")) {
        if (value < 0 || value > 2) return -LZ4FLEX_E_INVALID_ARG;
        c->dec_pcd_pair = value;
        return 0;
    }
    if (!strcmp(key, "compress_carry_wait")) {
        if (value != 0 && value != 1) return -LZ4FLEX_E_INVALID_ARG;
        c->comp_carry_wait = value;
        return 0;
    }
    if (!strcmp(key, "compress_sliding_window")) {
        if (value < 0 || value > 2) return -LZ4FLEX_E_INVALID_ARG;
        c->comp_sliding = value;
        return 0;
    }
    if (!strcmp(key, "compress_variant")) {
#ifdef LZ4FLEX_ALL_VARIANTS
        if (value != 1 && value != 3) return -LZ4FLEX_E_INVALID_ARG;
#else
        if (value != 1) return -LZ4FLEX_E_INVALID_ARG;           // 3 (no emitter wavefront) exists in variant builds only
#endif
        c->comp_variant = value;
        return 0;
    }
    if (!strcmp(key, "compress_lanes")) {
        if (value != 8 && value != 16) return -LZ4FLEX_E_INVALID_ARG;
        c->comp_lanes = value;
        return 0;
    }
    // Test hooks (fault injection; unsupported, see the header): only in a process that opted in with LZ4FLEX_TEST_HOOKS=1 (tests/conftest.py
    // does) -- the frame layer and every scalar call of a process share the default context, a hook set there by anybody else would make
    // unrelated users fail or take the slow redo path
    if (!strncmp(key, "debug_", 6)) {
        const char* e = getenv("LZ4FLEX_TEST_HOOKS");
        if (!e || strcmp(e, "1") != 0) return -LZ4FLEX_E_INVALID_ARG;
    }
    if (!strcmp(key, "debug_chain_giveup")) {
        if (value < 0) return -LZ4FLEX_E_INVALID_ARG;
        c->chain_giveup = value;
        return 0;
    }
    if (!strcmp(key, "debug_fail_next_batch")) {                 // tests/test_gpu_sharded_native.py: one rank's call-level failure
        if (value < 0) return -LZ4FLEX_E_INVALID_ARG;
        c->fail_next_batch = value;
        return 0;
    }
    return -LZ4FLEX_E_INVALID_ARG;
}

int lz4flex_get_tuning(lz4flex_ctx* c, const char* key) {
    if (!key) return -LZ4FLEX_E_INVALID_ARG;
    // the batch sizes at which the default decoder dispatch changes kernel or geometry (lz4_device.h), in ascending order; the
    // list ends where the key is refused.  tests/test_gpu_block.py builds its size matrix from it.
    if (!strncmp(key, "dispatch_threshold_", 19)) {
        const uint32_t t[] = {PCD_PAIR_MAX_BLOCKS, DISPATCH_PCD_1024, DISPATCH_PCD_512, DISPATCH_PCD_256, DISPATCH_SEQ_MAX, DISPATCH_SPLIT_FULL};
        const int i = atoi(key + 19);
        if (i < 0 || i >= (int)(sizeof t / sizeof t[0]) || (key[19] < '0' || key[19] > '9')) return -LZ4FLEX_E_INVALID_ARG;
        return (int)t[i];
    }
    // every decoder configuration this build can be pinned to, as variant * 1000 + parameter ("decompress_lanes" for variant 1,
    // "decompress_blocks_per_wg" for variant 4, else 0); the list ends where the key is refused.  ONE list: tests/test_gpu_block.py's
    // decoder matrix and tools/gpu_fuzz.py are generated from it (a decoder added here is tested there), no device needed.
    if (!strncmp(key, "decoder_config_", 15)) {
        const int t[] = {1016, 4008, 4032, 4064, 7000, 8000, 10000, 11000, 13000,
#ifdef LZ4FLEX_TOOLS
                         9000, 12000,
#endif
        };
        const int i = atoi(key + 15);
        if (i < 0 || i >= (int)(sizeof t / sizeof t[0]) || (key[15] < '0' || key[15] > '9')) return -LZ4FLEX_E_INVALID_ARG;
        return t[i];
    }
    if (!c) { const int rc = default_ctx(&c); if (rc) return rc; }
    if (!strcmp(key, "compress_mode")) return c->comp_mode;
    if (!strcmp(key, "compress_variant")) return c->comp_variant;
    if (!strcmp(key, "compress_carry_wait")) return c->comp_carry_wait;
    if (!strcmp(key, "compress_sliding_window")) return c->comp_sliding;
    if (!strcmp(key, "compress_subwindows")) return c->comp_sub;
    if (!strcmp(key, "compress_deterministic")) return c->comp_det;
    if (!strcmp(key, "compress_workgroups")) return c->wave_wgs;         // (what "compress_subwindows" 0 decides by: n * 4 <= this -> 4, n * 3 <= this -> 3, n * 2 <= this -> 2)
    if (!strcmp(key, "decompress_pcd_pair")) return c->dec_pcd_pair;
    if (!strcmp(key, "compress_lanes")) return c->comp_lanes;
    if (!strcmp(key, "decompress_variant")) return c->dec_variant;
    if (!strcmp(key, "decompress_second_pass")) return c->dec_second_pass;
    if (!strcmp(key, "decompress_blocks_per_wg")) return c->dec_blocks_per_wg;
    if (!strcmp(key, "decompress_level_chains")) return c->dec_level_chains;
    if (!strcmp(key, "decompress_lanes")) return c->dec_lanes;
    return -LZ4FLEX_E_INVALID_ARG;
}

int lz4flex_abi_version(void) { return 6; }

size_t lz4flex_get_maximum_output_size(size_t input_len) {
    return 16 + 4 + (size_t)((uint64_t)input_len * 110 / 100);
}

}  // extern "C"

// ------------------------------------------------------------------------------------------
// host-resident batches: mirror the caller's layout in the device arena, run, copy back
namespace {

struct Span { uint64_t lo = ~0ull, hi = 0; };
static Span span_of(const uint64_t* off, const uint32_t* len, uint32_t n) {
    Span s;
    for (uint32_t i = 0; i < n; i++) {
        s.lo = std::min<uint64_t>(s.lo, off[i]);
        s.hi = std::max<uint64_t>(s.hi, off[i] + len[i]);
    }
    if (n == 0 || s.lo > s.hi) { s.lo = 0; s.hi = 0; }
    return s;
}
static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct HostBatch {
    // device views
    uint8_t *d_in = nullptr, *d_out = nullptr, *d_dict = nullptr;
    uint64_t *d_in_off = nullptr, *d_out_off = nullptr, *d_dict_off = nullptr, *d_detail = nullptr;
    uint32_t *d_in_len = nullptr, *d_out_cap = nullptr, *d_out_len = nullptr, *d_flags = nullptr, *d_dict_len = nullptr,
             *d_out_pos = nullptr;
    int32_t* d_status = nullptr;
    Span in_span, out_span, dict_span;
};

}  // namespace

struct lz4flex_decompress_ext_ {
    const void* dict_base;
    const uint64_t* dict_off;
    const uint32_t* dict_len;
    const uint32_t* out_pos;
    const uint32_t* chain_prev;
    uint32_t n_chains;
};

// A host batch of a few small blocks (the scalar calls above all: compress_into / decompress_into are 1-block batches): ONE transfer up
// (descriptors + input, from page-locked staging), the kernels, ONE transfer down (results + output), one synchronisation -- the
// general path below costs two transfers each way from pageable memory and two synchronisations (0.16 / 0.38 ms per 64 KiB block in
// round 3).  Same semantics: a block that failed leaves the caller's bytes alone.
static constexpr uint32_t SMALL_BATCH_BLOCKS = 8u;
static constexpr size_t SMALL_BATCH_BYTES = 4u << 20;
static int run_host_small(lz4flex_ctx* c, bool compress, const uint8_t* in_base, const uint64_t* in_off, const uint32_t* in_len,
                          const uint32_t* flags, uint32_t n, uint8_t* out_base, const uint64_t* out_off, const uint32_t* out_cap,
                          uint32_t* out_len, int32_t* status, uint64_t* detail) {
    size_t in_bytes = 0, out_bytes = 0;
    for (uint32_t i = 0; i < n; i++) { in_bytes += align_up(in_len[i] + 16, 64); out_bytes += align_up((size_t)out_cap[i] + 16, 64); }
    // [up: in_off out_off in_len out_cap flags | input blocks]  [down: out_len status detail | output slots]
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o = align_up(o + bytes, 64); return at; };
    const size_t at_in_off = take(8ull * n), at_out_off = take(8ull * n), at_in_len = take(4ull * n), at_out_cap = take(4ull * n),
                 at_flags = take(4ull * n), at_in = take(in_bytes);
    const size_t up_bytes = o;
    const size_t at_out_len = take(4ull * n), at_status = take(4ull * n), at_detail = take(16ull * n), at_out = take(out_bytes);
    const size_t total = o;
    int rc;
    if ((rc = ensure_pin(c, total))) return rc;
    if ((rc = ensure_arena(c, total + 256))) return rc;
    uint8_t* hp = c->h_pin;
    uint8_t* d = c->d_arena;
    size_t ia = 0, oa = 0;
    for (uint32_t i = 0; i < n; i++) {
        ((uint64_t*)(hp + at_in_off))[i] = ia;
        ((uint64_t*)(hp + at_out_off))[i] = oa;
        ((uint32_t*)(hp + at_in_len))[i] = in_len[i];
        ((uint32_t*)(hp + at_out_cap))[i] = out_cap[i];
        ((uint32_t*)(hp + at_flags))[i] = flags ? flags[i] : 0u;
        if (in_len[i]) memcpy(hp + at_in + ia, in_base + in_off[i], in_len[i]);
        ia += align_up(in_len[i] + 16, 64);
        oa += align_up((size_t)out_cap[i] + 16, 64);
    }
    hipStream_t s = c->stream;
    HIP_TRY(hipMemcpyAsync(d, hp, up_bytes, hipMemcpyHostToDevice, s));
    if (compress) {
        CompressArgs a{};
        a.in_base = d + at_in; a.in_off = (const uint64_t*)(d + at_in_off); a.in_len = (const uint32_t*)(d + at_in_len);
        a.flags = flags ? (const uint32_t*)(d + at_flags) : nullptr;
        a.out_base = d + at_out; a.out_off = (const uint64_t*)(d + at_out_off); a.out_cap = (const uint32_t*)(d + at_out_cap);
        a.out_len = (uint32_t*)(d + at_out_len); a.status = (int32_t*)(d + at_status); a.n = n;
        bool big = false;
        for (uint32_t i = 0; i < n; i++) big |= in_len[i] > 65536u;
        if ((rc = launch_compress_any(c, a, big, s))) return rc;
    } else {
        DecompressArgs a{};
        a.in_base = d + at_in; a.in_off = (const uint64_t*)(d + at_in_off); a.in_len = (const uint32_t*)(d + at_in_len);
        a.out_base = d + at_out; a.out_off = (const uint64_t*)(d + at_out_off); a.out_cap = (const uint32_t*)(d + at_out_cap);
        a.out_len = (uint32_t*)(d + at_out_len); a.status = (int32_t*)(d + at_status); a.detail = (uint64_t*)(d + at_detail); a.n = n;
        bool big = false;
        for (uint32_t i = 0; i < n; i++) big |= in_len[i] > 131072u;
        const hipError_t le = c->dec_variant != 1 ? launch_decompress_fast(c, a, s, big) : launch_decompress(a, c->dec_lanes, s);
        if (le != hipSuccess) return hip_fail(le, "kernel launch");
    }
    HIP_TRY(hipMemcpyAsync(hp + at_out_len, d + at_out_len, total - at_out_len, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    oa = 0;
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t len = ((const uint32_t*)(hp + at_out_len))[i];
        const int32_t st = ((const int32_t*)(hp + at_status))[i];
        out_len[i] = len;
        status[i] = st;
        if (detail) { detail[2 * i] = compress ? 0 : ((const uint64_t*)(hp + at_detail))[2 * i]; detail[2 * i + 1] = compress ? 0 : ((const uint64_t*)(hp + at_detail))[2 * i + 1]; }
        if (st == 0 && len) memcpy(out_base + out_off[i], hp + at_out + oa, len);
        oa += align_up((size_t)out_cap[i] + 16, 64);
    }
    return 0;
}

static int run_host_batch(lz4flex_ctx* c, bool compress, const uint8_t* in_base, const uint64_t* in_off,
                          const uint32_t* in_len, const uint32_t* flags, uint32_t n, uint8_t* out_base,
                          const uint64_t* out_off, const uint32_t* out_cap, uint32_t* out_len, int32_t* status,
                          uint64_t* detail, const lz4flex_decompress_ext_* ext, bool chained = false) {
    if (n == 0) return 0;
    if (chained && (n > CHAIN_WS_BLOCKS || !ext || !ext->out_pos || ext->dict_base)) return -LZ4FLEX_E_INVALID_ARG;
    if (ext && ext->chain_prev) return -LZ4FLEX_E_INVALID_ARG;      // several chains in one batch: DEVICE batches only (a host batch stages ONE output region)
    int prev = 0;
    (void)hipGetDevice(&prev);
    HIP_TRY(hipSetDevice(c->device));
    struct Restore { int d; ~Restore() { (void)hipSetDevice(d); } } restore{prev};
    // LZ4FLEX_BLOCK_HISTORY: the bytes in front of a block travel with it
    uint64_t hist_lo = ~0ull;
    if (compress && flags)
        for (uint32_t i = 0; i < n; i++) {
            const uint64_t h = flags[i] >> 8;
            if (h > in_off[i]) return -LZ4FLEX_E_INVALID_ARG;
            if (h) hist_lo = std::min<uint64_t>(hist_lo, in_off[i] - h);
        }
    if (n <= SMALL_BATCH_BLOCKS && !ext && !chained && hist_lo == ~0ull) {
        size_t bytes = 0;
        for (uint32_t i = 0; i < n; i++) bytes += (size_t)in_len[i] + out_cap[i];
        if (bytes <= SMALL_BATCH_BYTES) return run_host_small(c, compress, in_base, in_off, in_len, flags, n, out_base, out_off, out_cap, out_len, status, detail);
    }

    HostBatch hb;
    hb.in_span = span_of(in_off, in_len, n);
    if (hist_lo < hb.in_span.lo) hb.in_span.lo = hist_lo;
    hb.out_span = span_of(out_off, out_cap, n);
    const bool has_dict = ext && ext->dict_base && ext->dict_off && ext->dict_len;
    const bool has_pos = ext && ext->out_pos;
    if (has_dict) hb.dict_span = span_of(ext->dict_off, ext->dict_len, n);
    const size_t in_bytes = (size_t)(hb.in_span.hi - hb.in_span.lo);
    const size_t out_bytes = (size_t)(hb.out_span.hi - hb.out_span.lo);
    const size_t dict_bytes = has_dict ? (size_t)(hb.dict_span.hi - hb.dict_span.lo) : 0;
    // descriptor block (pinned host mirror and device copy share one layout)
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o = align_up(o + bytes, 16); return at; };
    const size_t at_in_off = take(8ull * n), at_out_off = take(8ull * n), at_dict_off = take(has_dict ? 8ull * n : 0),
                 at_in_len = take(4ull * n), at_out_cap = take(4ull * n), at_flags = take(flags ? 4ull * n : 0),
                 at_dict_len = take(has_dict ? 4ull * n : 0), at_out_pos = take(has_pos ? 4ull * n : 0);
    const size_t desc_in_bytes = o;
    const size_t at_out_len = take(4ull * n), at_status = take(4ull * n), at_detail = take(16ull * n);
    const size_t desc_bytes = o;
    int rc;
    if ((rc = ensure_pin(c, desc_bytes))) return rc;
    const size_t a_in = 0, a_out = align_up(in_bytes + 64, 256), a_dict = a_out + align_up(out_bytes + 64, 256),
                 a_desc = a_dict + align_up(dict_bytes + 64, 256);
    if ((rc = ensure_arena(c, a_desc + desc_bytes + 256))) return rc;
    uint8_t* hp = c->h_pin;
    for (uint32_t i = 0; i < n; i++) {
        ((uint64_t*)(hp + at_in_off))[i] = in_off[i] - hb.in_span.lo;
        ((uint64_t*)(hp + at_out_off))[i] = out_off[i] - hb.out_span.lo;
        ((uint32_t*)(hp + at_in_len))[i] = in_len[i];
        ((uint32_t*)(hp + at_out_cap))[i] = out_cap[i];
        if (flags) ((uint32_t*)(hp + at_flags))[i] = flags[i];
        if (has_dict) {
            ((uint64_t*)(hp + at_dict_off))[i] = ext->dict_off[i] - hb.dict_span.lo;
            ((uint32_t*)(hp + at_dict_len))[i] = ext->dict_len[i];
        }
        if (has_pos) ((uint32_t*)(hp + at_out_pos))[i] = ext->out_pos[i];
    }
    uint8_t* d = c->d_arena;
    hipStream_t s = c->stream;
    if (in_bytes) HIP_TRY(hipMemcpyAsync(d + a_in, in_base + hb.in_span.lo, in_bytes, hipMemcpyHostToDevice, s));
    if (dict_bytes)
        HIP_TRY(hipMemcpyAsync(d + a_dict, (const uint8_t*)ext->dict_base + hb.dict_span.lo, dict_bytes,
                               hipMemcpyHostToDevice, s));
    if (has_pos && out_bytes) {
        // prefix mode: the sink already holds bytes [0, out_pos) that matches may reference; in a chained batch only what lies
        // before the FIRST block's position exists yet (the rest is what the batch produces)
        const size_t up = chained ? std::min<size_t>(out_bytes, (size_t)(out_off[0] - hb.out_span.lo) + ext->out_pos[0]) : out_bytes;
        if (up) HIP_TRY(hipMemcpyAsync(d + a_out, out_base + hb.out_span.lo, up, hipMemcpyHostToDevice, s));
    }
    HIP_TRY(hipMemcpyAsync(d + a_desc, hp, desc_in_bytes, hipMemcpyHostToDevice, s));
    uint8_t* dd = d + a_desc;
    hipError_t le;
    if (compress) {
        CompressArgs a{};
        a.in_base = d + a_in; a.in_off = (const uint64_t*)(dd + at_in_off); a.in_len = (const uint32_t*)(dd + at_in_len);
        a.flags = flags ? (const uint32_t*)(dd + at_flags) : nullptr;
        a.out_base = d + a_out; a.out_off = (const uint64_t*)(dd + at_out_off); a.out_cap = (const uint32_t*)(dd + at_out_cap);
        a.out_len = (uint32_t*)(dd + at_out_len); a.status = (int32_t*)(dd + at_status); a.n = n;
        bool big = false;
        for (uint32_t i = 0; i < n; i++) big |= in_len[i] > 65536u;
        if ((rc = launch_compress_any(c, a, big, s))) return rc;
        le = hipSuccess;
    } else {
        DecompressArgs a{};
        a.in_base = d + a_in; a.in_off = (const uint64_t*)(dd + at_in_off); a.in_len = (const uint32_t*)(dd + at_in_len);
        a.out_base = d + a_out; a.out_off = (const uint64_t*)(dd + at_out_off); a.out_cap = (const uint32_t*)(dd + at_out_cap);
        a.out_pos = has_pos ? (const uint32_t*)(dd + at_out_pos) : nullptr;
        a.dict_base = has_dict ? d + a_dict : nullptr;
        a.dict_off = has_dict ? (const uint64_t*)(dd + at_dict_off) : nullptr;
        a.dict_len = has_dict ? (const uint32_t*)(dd + at_dict_len) : nullptr;
        a.out_len = (uint32_t*)(dd + at_out_len); a.status = (int32_t*)(dd + at_status);
        a.detail = (uint64_t*)(dd + at_detail); a.n = n;
        bool big = false;                                    // host arrays are visible: blocks beyond 128 KiB compressed are "large"
        for (uint32_t i = 0; i < n; i++) big |= in_len[i] > 131072u;
        if (chained) {
            HIP_TRY(chain_ws_begin(c, n, s));
            a.chain_done = c->chain_ws;
            a.debug_giveup = (uint32_t)c->chain_giveup;
        }
        le = ((c->dec_variant != 1 || chained) && !has_dict) ? launch_decompress_fast(c, a, s, big) : launch_decompress(a, c->dec_lanes, s);
        if (chained) (void)chain_ws_end(c, s);               // (also behind a failed launch: whatever was enqueued is ordered)
    }
    if (le != hipSuccess) return hip_fail(le, "kernel launch");
    HIP_TRY(hipMemcpyAsync(hp + at_out_len, dd + at_out_len, desc_bytes - at_out_len, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    const uint32_t* r_len = (const uint32_t*)(hp + at_out_len);
    const int32_t* r_st = (const int32_t*)(hp + at_status);
    const uint64_t* r_det = (const uint64_t*)(hp + at_detail);
    uint64_t produced = 0;
    for (uint32_t i = 0; i < n; i++) {
        out_len[i] = r_len[i];
        status[i] = r_st[i];
        if (detail) { detail[2 * i] = compress ? 0 : r_det[2 * i]; detail[2 * i + 1] = compress ? 0 : r_det[2 * i + 1]; }
        if (r_st[i] == 0) produced += r_len[i];
    }
    // copy results back: dense outputs in one transfer, sparse ones block by block
    if (out_bytes && produced * 2 >= out_bytes && !has_pos) {
        // every successful block owns [out_off, out_off+len); failed blocks must leave the caller's bytes alone
        // ... and the one-shot copy covers the whole span, so the blocks must TILE it: with gaps between the callers' slots (a
        // padded stride) it would overwrite host bytes that belong to nobody with whatever the device arena holds
        // (round 6: the block that ENDS the span may be short -- a frame's last block, lz4flex_frame_decompress decoding straight into the
        // caller's buffer -- the transfer then stops at its last byte)
        bool all_ok = true;
        uint64_t cap_sum = 0, short_tail = 0;
        for (uint32_t i = 0; i < n; i++) {
            cap_sum += out_cap[i];
            if (r_st[i] != 0) { all_ok = false; break; }
            if (r_len[i] == out_cap[i]) continue;
            if (out_off[i] + out_cap[i] == hb.out_span.hi && short_tail == 0 && r_len[i] < out_cap[i]) short_tail = out_cap[i] - r_len[i];
            else { all_ok = false; break; }
        }
        all_ok = all_ok && cap_sum == (uint64_t)out_bytes;
        if (all_ok) {
            if (out_bytes > short_tail) HIP_TRY(hipMemcpyAsync(out_base + hb.out_span.lo, d + a_out, out_bytes - (size_t)short_tail, hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
            return 0;
        }
    }
    // sparse results, many blocks: pack them on the device into the (now free) input region, one transfer to pinned
    // memory, scatter on the host.  A per-block hipMemcpy to pageable memory costs ~10 us each.
    if (!has_pos && n >= 16u && produced != 0 && produced <= (uint64_t)align_up(in_bytes + 64, 256) &&
        ensure_pay(c, (size_t)produced) == 0) {
        uint64_t* h_dense = (uint64_t*)(hp + at_detail);          // reuse the detail slot (16 B per block >= 8 B)
        uint64_t at = 0;
        for (uint32_t i = 0; i < n; i++) { h_dense[i] = at; if (r_st[i] == 0) at += r_len[i]; }
        uint64_t* d_dense = (uint64_t*)(dd + at_detail);
        HIP_TRY(hipMemcpyAsync(d_dense, h_dense, 8ull * n, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(lz4flex_pack_results_kernel, dim3(n), dim3(256), 0, s, d + a_out,
                           (const uint64_t*)(dd + at_out_off), (const uint32_t*)(dd + at_out_len),
                           (const int32_t*)(dd + at_status), d_dense, d + a_in, n);
        const hipError_t pe = hipGetLastError();              // (reading it clears it: read once)
        if (pe != hipSuccess) return hip_fail(pe, "pack kernel launch");
        HIP_TRY(hipMemcpyAsync(c->h_pay, d + a_in, (size_t)produced, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        at = 0;
        for (uint32_t i = 0; i < n; i++) {
            if (r_st[i] != 0 || r_len[i] == 0) continue;
            memcpy(out_base + out_off[i], c->h_pay + at, r_len[i]);
            at += r_len[i];
        }
        return 0;
    }
    if (chained) {
        // one region, the blocks back to back behind the first one's position: one transfer up to the end of the last good block
        uint64_t lo = out_off[0] + ext->out_pos[0], hi = lo;
        for (uint32_t i = 0; i < n; i++)
            if (r_st[i] == 0) hi = std::max<uint64_t>(hi, out_off[i] + ext->out_pos[i] + r_len[i]);
        if (hi > lo) HIP_TRY(hipMemcpyAsync(out_base + lo, d + a_out + (lo - hb.out_span.lo), (size_t)(hi - lo), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        return 0;
    }
    for (uint32_t i = 0; i < n; i++) {
        if (r_st[i] != 0 || r_len[i] == 0) continue;
        const uint32_t pos = has_pos ? ext->out_pos[i] : 0u;
        HIP_TRY(hipMemcpyAsync(out_base + out_off[i] + pos, d + a_out + (out_off[i] - hb.out_span.lo) + pos, r_len[i],
                               hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(hipStreamSynchronize(s));
    return 0;
}

static int run_device_batch(lz4flex_ctx* c, bool compress, const void* in_base, const uint64_t* in_off,
                            const uint32_t* in_len, const uint32_t* flags, uint32_t n, void* out_base,
                            const uint64_t* out_off, const uint32_t* out_cap, uint32_t* out_len, int32_t* status,
                            uint64_t* detail, const lz4flex_decompress_ext_* ext, void* hip_stream, int big_hint, bool chained = false) {
    if (chained && (compress || n > CHAIN_WS_BLOCKS || !ext || !ext->out_pos || ext->dict_base)) return -LZ4FLEX_E_INVALID_ARG;
    if (!chained && ext && ext->chain_prev) return -LZ4FLEX_E_INVALID_ARG;
    hipStream_t s = (hipStream_t)hip_stream;   // DEVICE batches run on the caller's stream (NULL = HIP's null stream)
    hipError_t le;
    if (compress) {
        CompressArgs a{};
        a.in_base = (const uint8_t*)in_base; a.in_off = in_off; a.in_len = in_len; a.flags = flags;
        a.out_base = (uint8_t*)out_base; a.out_off = out_off; a.out_cap = out_cap; a.out_len = out_len;
        a.status = status; a.n = n;
        const int rc = launch_compress_any(c, a, big_hint != 0, s);
        if (rc) return rc;
        le = hipSuccess;
    } else {
        DecompressArgs a{};
        a.in_base = (const uint8_t*)in_base; a.in_off = in_off; a.in_len = in_len;
        a.out_base = (uint8_t*)out_base; a.out_off = out_off; a.out_cap = out_cap;
        a.out_pos = ext ? ext->out_pos : nullptr;
        a.dict_base = ext ? (const uint8_t*)ext->dict_base : nullptr;
        a.dict_off = ext ? ext->dict_off : nullptr;
        a.dict_len = ext ? ext->dict_len : nullptr;
        a.out_len = out_len; a.status = status; a.detail = detail; a.n = n;
        if (chained && n) {
            HIP_TRY(chain_ws_begin(c, n, s));
            a.chain_done = c->chain_ws;
            a.chain_prev = ext->chain_prev;
            a.n_chains = ext->n_chains;
            a.debug_giveup = (uint32_t)c->chain_giveup;
        }
        le = ((c->dec_variant != 1 || chained) && !a.dict_base) ? launch_decompress_fast(c, a, s, big_hint != 0) : launch_decompress(a, c->dec_lanes, s);
        if (chained && n) (void)chain_ws_end(c, s);
    }
    if (le != hipSuccess) return hip_fail(le, "kernel launch");
    return 0;
}

static thread_local lz4flex_ctx* g_default_ctx = nullptr;
static int default_ctx(lz4flex_ctx** out) {
    if (!g_default_ctx) {
        int rc = lz4flex_ctx_create(&g_default_ctx, -1);
        if (rc) return rc;
    }
    *out = g_default_ctx;
    return 0;
}

// frame_many.cpp's view of a context (lz4_device.h): the context or the calling thread's default one; its device; grow-only scratch
namespace lz4flex_dev {
int ctx_resolve(lz4flex_ctx** ctx) { return *ctx ? 0 : default_ctx(ctx); }
int ctx_device(lz4flex_ctx* c) { return c->device; }
hipStream_t ctx_stream(lz4flex_ctx* c) { return c->stream; }
int ctx_comp_mode(lz4flex_ctx* c) { return c->comp_mode; }
int ctx_scratch(lz4flex_ctx* c, int slot, size_t bytes, void** out) {
    if (slot < 0 || slot >= 4) return -LZ4FLEX_E_INVALID_ARG;
    if (bytes > c->many_cap[slot]) {
        if (c->many_ws[slot]) (void)hipFree(c->many_ws[slot]);
        c->many_ws[slot] = nullptr; c->many_cap[slot] = 0;
        const size_t want = bytes + bytes / 8 + 4096;
        const hipError_t e = hipMalloc(&c->many_ws[slot], want);
        if (e != hipSuccess) { (void)hip_fail(e, "scratch"); return e == hipErrorOutOfMemory ? -LZ4FLEX_E_NOMEM : -LZ4FLEX_E_HIP; }
        c->many_cap[slot] = want;
    }
    *out = c->many_ws[slot];
    return 0;
}
}  // namespace lz4flex_dev

extern "C" {

int lz4flex_compress_batch(lz4flex_ctx* ctx, const void* in_base, const uint64_t* in_off, const uint32_t* in_len,
                           const uint32_t* flags, uint32_t n, void* out_base, const uint64_t* out_off,
                           const uint32_t* out_cap, uint32_t* out_len, int32_t* status, int mem_kind,
                           void* hip_stream) {
    int rc;
    if (!ctx && (rc = default_ctx(&ctx))) return rc;
    if (n && (!in_off || !in_len || !out_off || !out_cap || !out_len || !status)) return -LZ4FLEX_E_INVALID_ARG;
    if (ctx->fail_next_batch > 0) { ctx->fail_next_batch--; g_last_error = "debug_fail_next_batch"; return -LZ4FLEX_E_HIP; }
    if (mem_kind == LZ4FLEX_MEM_HOST)
        return run_host_batch(ctx, true, (const uint8_t*)in_base, in_off, in_len, flags, n, (uint8_t*)out_base, out_off,
                              out_cap, out_len, status, nullptr, nullptr);
    if ((mem_kind & 0xFF) == LZ4FLEX_MEM_DEVICE)
        return run_device_batch(ctx, true, in_base, in_off, in_len, flags, n, out_base, out_off, out_cap, out_len, status,
                                nullptr, nullptr, hip_stream, (mem_kind & LZ4FLEX_MEM_BIG_BLOCKS) != 0);
    return -LZ4FLEX_E_INVALID_ARG;
}

int lz4flex_decompress_batch_ex(lz4flex_ctx* ctx, const void* in_base, const uint64_t* in_off, const uint32_t* in_len,
                                uint32_t n, void* out_base, const uint64_t* out_off, const uint32_t* out_cap,
                                uint32_t* out_len, int32_t* status, uint64_t* detail,
                                const lz4flex_decompress_ext* ext, int mem_kind, void* hip_stream) {
    int rc;
    if (!ctx && (rc = default_ctx(&ctx))) return rc;
    if (n && (!in_off || !in_len || !out_off || !out_cap || !out_len || !status)) return -LZ4FLEX_E_INVALID_ARG;
    if (ctx->fail_next_batch > 0) { ctx->fail_next_batch--; g_last_error = "debug_fail_next_batch"; return -LZ4FLEX_E_HIP; }
    lz4flex_decompress_ext_ e{};
    const bool chained = (mem_kind & LZ4FLEX_MEM_CHAINED) != 0;
    if (ext) {
        e.dict_base = ext->dict_base; e.dict_off = ext->dict_off; e.dict_len = ext->dict_len; e.out_pos = ext->out_pos;
        // chain_prev / n_chains (round 5) lie behind the four members every earlier caller knows: they are only looked at in the one
        // batch shape that uses them -- a caller built against the round-4 header hands over a 32-byte struct (lz4flex_abi_version)
        if (chained && (mem_kind & 0xFF) == LZ4FLEX_MEM_DEVICE) { e.chain_prev = ext->chain_prev; e.n_chains = ext->n_chains; }
    }
    if ((mem_kind & 0xFF) == LZ4FLEX_MEM_HOST)
        return run_host_batch(ctx, false, (const uint8_t*)in_base, in_off, in_len, nullptr, n, (uint8_t*)out_base, out_off,
                              out_cap, out_len, status, detail, ext ? &e : nullptr, chained);
    if ((mem_kind & 0xFF) == LZ4FLEX_MEM_DEVICE)
        return run_device_batch(ctx, false, in_base, in_off, in_len, nullptr, n, out_base, out_off, out_cap, out_len,
                                status, detail, ext ? &e : nullptr, hip_stream, (mem_kind & LZ4FLEX_MEM_BIG_BLOCKS) != 0, chained);
    return -LZ4FLEX_E_INVALID_ARG;
}

int lz4flex_decompress_batch(lz4flex_ctx* ctx, const void* in_base, const uint64_t* in_off, const uint32_t* in_len,
                             uint32_t n, void* out_base, const uint64_t* out_off, const uint32_t* out_cap,
                             uint32_t* out_len, int32_t* status, uint64_t* detail, int mem_kind, void* hip_stream) {
    return lz4flex_decompress_batch_ex(ctx, in_base, in_off, in_len, n, out_base, out_off, out_cap, out_len, status,
                                       detail, nullptr, mem_kind, hip_stream);
}

// ---- scalar, lz4_flex-shaped --------------------------------------------------------------
int64_t lz4flex_compress_into(const uint8_t* in, size_t in_len, uint8_t* out, size_t out_cap) {
    if (in_len > 0xFFFFFFFFull) return -LZ4FLEX_E_INVALID_ARG;
    // compress.rs:338-340: OutputTooSmall is decided up front, before anything is written
    if (out_cap < lz4flex_get_maximum_output_size(in_len)) return -LZ4FLEX_E_OUTPUT_TOO_SMALL;
    lz4flex_ctx* c;
    int rc = default_ctx(&c);
    if (rc) return rc;
    const uint64_t off0 = 0;
    const uint32_t len = (uint32_t)in_len, cap = (uint32_t)std::min<size_t>(out_cap, 0xFFFFFFFFull);
    uint32_t olen = 0;
    int32_t st = 0;
    static const uint8_t empty = 0;
    rc = run_host_batch(c, true, in ? in : &empty, &off0, &len, nullptr, 1, out, &off0, &cap, &olen, &st, nullptr, nullptr);
    if (rc) return rc;
    if (st) return -(int64_t)st;
    return (int64_t)olen;
}

int lz4flex_compress_chains(lz4flex_ctx* ctx, const void* in_base, const lz4flex_chain_block* blocks, uint32_t n_blocks,
                            const uint32_t* chain_first, const uint32_t* chain_count, uint32_t n_chains, void* out_base,
                            const uint64_t* out_off, const uint32_t* out_cap, uint32_t* out_len, int32_t* status,
                            uint32_t* tbl_state, int mem_kind, void* hip_stream) {
    static_assert(sizeof(lz4flex_chain_block) == 40, "ChainBlock layout");
    int rc;
    if (!ctx && (rc = default_ctx(&ctx))) return rc;
    if (n_chains == 0 || n_blocks == 0) return 0;
    if (!blocks || !chain_first || !chain_count || !out_off || !out_cap || !out_len || !status) return -LZ4FLEX_E_INVALID_ARG;
    if ((mem_kind & 0xFF) == LZ4FLEX_MEM_DEVICE) {
        hipError_t le = launch_compress_chain((const uint8_t*)in_base, blocks, chain_first, chain_count, n_chains,
                                              (uint8_t*)out_base, out_off, out_cap, out_len, status, tbl_state,
                                              (hipStream_t)hip_stream);
        return le == hipSuccess ? 0 : hip_fail(le, "chain kernel launch");
    }
    if (mem_kind != LZ4FLEX_MEM_HOST) return -LZ4FLEX_E_INVALID_ARG;
    for (uint32_t k = 0; k < n_chains; k++)                     // host arrays can be checked: a chain stays inside `blocks`
        if (chain_first[k] > n_blocks || chain_count[k] > n_blocks - chain_first[k]) return -LZ4FLEX_E_INVALID_ARG;
    lz4flex_ctx* c = ctx;
    int prev = 0;
    (void)hipGetDevice(&prev);
    HIP_TRY(hipSetDevice(c->device));
    struct Restore { int d; ~Restore() { (void)hipSetDevice(d); } } restore{prev};
    // spans of the caller's buffers
    uint64_t ilo = ~0ull, ihi = 0;
    for (uint32_t i = 0; i < n_blocks; i++) {
        ilo = std::min<uint64_t>(ilo, blocks[i].in_off);
        ihi = std::max<uint64_t>(ihi, blocks[i].in_off + blocks[i].in_len);
        if (blocks[i].dict_len) {
            ilo = std::min<uint64_t>(ilo, blocks[i].dict_off);
            ihi = std::max<uint64_t>(ihi, blocks[i].dict_off + blocks[i].dict_len);
        }
    }
    if (ilo > ihi) { ilo = 0; ihi = 0; }
    Span os = span_of(out_off, out_cap, n_blocks);
    const size_t in_bytes = (size_t)(ihi - ilo), out_bytes = (size_t)(os.hi - os.lo);
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o = align_up(o + bytes, 16); return at; };
    const size_t at_blocks = take(sizeof(lz4flex_chain_block) * (size_t)n_blocks), at_first = take(4ull * n_chains),
                 at_count = take(4ull * n_chains), at_out_off = take(8ull * n_blocks), at_out_cap = take(4ull * n_blocks);
    const size_t desc_in = o;
    const size_t at_out_len = take(4ull * n_blocks), at_status = take(4ull * n_blocks);
    const size_t desc_bytes = o;
    const size_t tbl_bytes = tbl_state ? 16384ull * n_chains : 0;
    if ((rc = ensure_pin(c, desc_bytes))) return rc;
    const size_t a_in = 0, a_out = align_up(in_bytes + 64, 256), a_desc = a_out + align_up(out_bytes + 64, 256),
                 a_tbl = a_desc + align_up(desc_bytes + 64, 256);
    if ((rc = ensure_arena(c, a_tbl + tbl_bytes + 256))) return rc;
    uint8_t* hp = c->h_pin;
    lz4flex_chain_block* hb = (lz4flex_chain_block*)(hp + at_blocks);
    for (uint32_t i = 0; i < n_blocks; i++) {
        hb[i] = blocks[i];
        hb[i].in_off -= ilo;
        hb[i].dict_off = blocks[i].dict_len ? blocks[i].dict_off - ilo : 0;
        ((uint64_t*)(hp + at_out_off))[i] = out_off[i] - os.lo;
        ((uint32_t*)(hp + at_out_cap))[i] = out_cap[i];
    }
    memcpy(hp + at_first, chain_first, 4ull * n_chains);
    memcpy(hp + at_count, chain_count, 4ull * n_chains);
    uint8_t* d = c->d_arena;
    hipStream_t s = c->stream;
    if (in_bytes) HIP_TRY(hipMemcpyAsync(d + a_in, (const uint8_t*)in_base + ilo, in_bytes, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d + a_desc, hp, desc_in, hipMemcpyHostToDevice, s));
    if (tbl_state) HIP_TRY(hipMemcpyAsync(d + a_tbl, tbl_state, tbl_bytes, hipMemcpyHostToDevice, s));
    uint8_t* dd = d + a_desc;
    hipError_t le = launch_compress_chain(d + a_in, dd + at_blocks, (const uint32_t*)(dd + at_first),
                                          (const uint32_t*)(dd + at_count), n_chains, d + a_out,
                                          (const uint64_t*)(dd + at_out_off), (const uint32_t*)(dd + at_out_cap),
                                          (uint32_t*)(dd + at_out_len), (int32_t*)(dd + at_status),
                                          tbl_state ? (uint32_t*)(d + a_tbl) : nullptr, s);
    if (le != hipSuccess) return hip_fail(le, "chain kernel launch");
    HIP_TRY(hipMemcpyAsync(hp + at_out_len, dd + at_out_len, desc_bytes - at_out_len, hipMemcpyDeviceToHost, s));
    if (tbl_state) HIP_TRY(hipMemcpyAsync(tbl_state, d + a_tbl, tbl_bytes, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    for (uint32_t i = 0; i < n_blocks; i++) {
        out_len[i] = ((const uint32_t*)(hp + at_out_len))[i];
        status[i] = ((const int32_t*)(hp + at_status))[i];
        if (status[i] == 0 && out_len[i])
            HIP_TRY(hipMemcpyAsync((uint8_t*)out_base + out_off[i], d + a_out + (out_off[i] - os.lo), out_len[i],
                                   hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(hipStreamSynchronize(s));
    return 0;
}

int64_t lz4flex_compress_into_with_dict(const uint8_t* in, size_t in_len, uint8_t* out, size_t out_cap,
                                        const uint8_t* dict, size_t dict_len) {
    if (in_len > 0x7FFFFFFFull) return -LZ4FLEX_E_INVALID_ARG;
    if (out_cap < lz4flex_get_maximum_output_size(in_len)) return -LZ4FLEX_E_OUTPUT_TOO_SMALL;   // compress.rs:338-340
    // compress_into_sink_with_dict, compress.rs:554-568: table kind from the UNtruncated dictionary length,
    // then init_dict keeps the last 64 KiB (:572-574)
    const bool h4 = dict_len + in_len < 65535u;
    if (dict_len > 65536u) { dict += dict_len - 65536u; dict_len = 65536u; }
    std::vector<uint8_t> buf(dict_len + in_len + 16);
    if (dict_len) memcpy(buf.data(), dict, dict_len);
    if (in_len) memcpy(buf.data() + dict_len, in, in_len);
    lz4flex_chain_block b{};
    b.in_off = dict_len; b.dict_off = 0; b.in_len = (uint32_t)in_len; b.in_pos = 0; b.dict_len = (uint32_t)dict_len;
    b.so = (uint32_t)dict_len; b.repos = 0; b.flags = (h4 ? 1u : 0u) | 2u;
    const uint32_t first = 0, count = 1;
    const uint64_t off0 = 0;
    const uint32_t cap = (uint32_t)std::min<size_t>(out_cap, 0xFFFFFFFFull);
    uint32_t olen = 0;
    int32_t st = 0;
    int rc = lz4flex_compress_chains(nullptr, buf.data(), &b, 1, &first, &count, 1, out, &off0, &cap, &olen, &st, nullptr,
                                     LZ4FLEX_MEM_HOST, nullptr);
    if (rc) return rc;
    if (st) return -(int64_t)st;
    return (int64_t)olen;
}

int64_t lz4flex_compress_prepend_size(const uint8_t* in, size_t in_len, uint8_t* out, size_t out_cap) {
    // compress.rs:624-634: 4-byte LE length, then the block
    if (in_len > 0xFFFFFFFFull) return -LZ4FLEX_E_INVALID_ARG;   // (nothing is written before the arguments are known to be fine)
    if (out_cap < 4 || out_cap - 4 < lz4flex_get_maximum_output_size(in_len)) return -LZ4FLEX_E_OUTPUT_TOO_SMALL;
    const uint32_t n = (uint32_t)in_len;
    out[0] = (uint8_t)n; out[1] = (uint8_t)(n >> 8); out[2] = (uint8_t)(n >> 16); out[3] = (uint8_t)(n >> 24);
    int64_t r = lz4flex_compress_into(in, in_len, out + 4, out_cap - 4);
    return r < 0 ? r : r + 4;
}

static int64_t decompress_common(const uint8_t* in, size_t in_len, uint8_t* out, size_t out_cap, const uint8_t* dict,
                                 size_t dict_len, bool use_dict, lz4flex_err_detail* detail) {
    if (in_len > 0xFFFFFFFFull || out_cap > 0xFFFFFFFFull || dict_len > 0xFFFFFFFFull) return -LZ4FLEX_E_INVALID_ARG;
    if (detail) memset(detail, 0, sizeof *detail);
    lz4flex_ctx* c;
    int rc = default_ctx(&c);
    if (rc) { if (detail) detail->hip_error = g_last_hip; return rc; }
    const uint64_t off0 = 0;
    const uint32_t len = (uint32_t)in_len, cap = (uint32_t)out_cap, dlen = (uint32_t)dict_len;
    uint32_t olen = 0;
    int32_t st = 0;
    uint64_t det[2] = {0, 0};
    static const uint8_t empty = 0;
    lz4flex_decompress_ext_ e{};
    e.dict_base = dict ? dict : &empty; e.dict_off = &off0; e.dict_len = &dlen; e.out_pos = nullptr;
    static uint8_t sink_dummy = 0;
    rc = run_host_batch(c, false, in ? in : &empty, &off0, &len, nullptr, 1, out ? out : &sink_dummy, &off0, &cap, &olen,
                        &st, det, use_dict ? &e : nullptr);
    if (rc) { if (detail) detail->hip_error = g_last_hip; return rc; }
    if (st) {
        if (detail) { detail->expected = det[0]; detail->actual = det[1]; }
        return -(int64_t)st;
    }
    return (int64_t)olen;
}

int64_t lz4flex_decompress_into(const uint8_t* in, size_t in_len, uint8_t* out, size_t out_cap,
                                lz4flex_err_detail* detail) {
    return decompress_common(in, in_len, out, out_cap, nullptr, 0, false, detail);
}

int64_t lz4flex_decompress_into_with_dict(const uint8_t* in, size_t in_len, uint8_t* out, size_t out_cap,
                                          const uint8_t* dict, size_t dict_len, lz4flex_err_detail* detail) {
    return decompress_common(in, in_len, out, out_cap, dict, dict_len, true, detail);
}

int lz4flex_xxh32_batch_device(const void* base, const uint64_t* off, const uint32_t* len, uint32_t n, uint32_t seed,
                               uint32_t* out, void* hip_stream) {
    if (n && (!base || !off || !len || !out)) return -LZ4FLEX_E_INVALID_ARG;
    hipError_t le = launch_xxh32_batch((const uint8_t*)base, off, len, n, seed, out, (hipStream_t)hip_stream);
    return le == hipSuccess ? 0 : hip_fail(le, "xxh32 kernel launch");
}

int lz4flex_frame_assemble_device(const void* src_base, const uint64_t* src_off, const uint32_t* in_len, const void* comp_base,
                                  const uint64_t* comp_off, const uint32_t* comp_len, uint32_t n, int block_checksums, void* seg,
                                  uint64_t* seg_off, void* scratch, void* hip_stream) {
    if (!seg_off || (n && (!src_base || !src_off || !in_len || !comp_base || !comp_off || !comp_len || !seg))) return -LZ4FLEX_E_INVALID_ARG;
    if (block_checksums && n && !scratch) return -LZ4FLEX_E_INVALID_ARG;
    uint8_t* sc = (uint8_t*)scratch;                                     // 16 bytes per block: u64 payload offset, u32 length, u32 XXH32
    hipError_t le = launch_frame_assemble((const uint8_t*)src_base, src_off, in_len, (const uint8_t*)comp_base, comp_off, comp_len, n,
                                          block_checksums, (uint8_t*)seg, seg_off, (uint64_t*)sc, (uint32_t*)(sc + 8ull * n),
                                          (uint32_t*)(sc + 12ull * n), (hipStream_t)hip_stream);
    return le == hipSuccess ? 0 : hip_fail(le, "frame assemble launch");
}

int lz4flex_copy_batch_device(const void* src_base, const uint64_t* src_off, const uint32_t* len, void* dst_base, const uint64_t* dst_off,
                              uint32_t n, void* hip_stream) {
    if (n && (!src_base || !src_off || !len || !dst_base || !dst_off)) return -LZ4FLEX_E_INVALID_ARG;
    hipError_t le = launch_copy_batch((const uint8_t*)src_base, src_off, len, (uint8_t*)dst_base, dst_off, n, (hipStream_t)hip_stream);
    return le == hipSuccess ? 0 : hip_fail(le, "copy batch launch");
}

int lz4flex_frame_walk_device(const void* frame, uint64_t frame_len, uint32_t header_len, int block_checksums, uint32_t block_size,
                              uint32_t max_blocks, uint64_t* payload_off, uint32_t* len_word, uint32_t* info, void* hip_stream) {
    if (!frame || !payload_off || !len_word || !info) return -LZ4FLEX_E_INVALID_ARG;
    hipError_t le = launch_frame_walk((const uint8_t*)frame, frame_len, header_len, block_checksums ? 4u : 0u, block_size, max_blocks,
                                      payload_off, len_word, info, (hipStream_t)hip_stream);
    return le == hipSuccess ? 0 : hip_fail(le, "frame walk launch");
}

// ---- CompressTable / compress_into_with_table, src/block/compress.rs:710-766 --------------------------------------
// The reference clears the table on every call: the handle only avoids re-allocating it and remembers its variant
// (Small = u16 entries + 4-byte hash, Large = u32 entries + 5-byte hash; upgraded, never downgraded).  Here the handle
// owns a context, i.e. the device workspace that is reused from call to call.
struct lz4flex_compress_table {
    lz4flex_ctx* ctx = nullptr;
    int large = 0;
};

lz4flex_compress_table* lz4flex_compress_table_new(int large) {
    lz4flex_compress_table* t = new (std::nothrow) lz4flex_compress_table();
    if (!t) return nullptr;
    t->large = large ? 1 : 0;
    if (lz4flex_ctx_create(&t->ctx, -1) != 0) { delete t; return nullptr; }
    return t;
}
void lz4flex_compress_table_free(lz4flex_compress_table* t) {
    if (!t) return;
    lz4flex_ctx_destroy(t->ctx);
    delete t;
}
int lz4flex_compress_table_is_large(const lz4flex_compress_table* t) { return t ? t->large : -LZ4FLEX_E_INVALID_ARG; }

int64_t lz4flex_compress_into_with_table(const uint8_t* in, size_t in_len, uint8_t* out, size_t out_cap, lz4flex_compress_table* t) {
    if (!t || in_len > 0xFFFFFFFFull) return -LZ4FLEX_E_INVALID_ARG;
    if (out_cap < lz4flex_get_maximum_output_size(in_len)) return -LZ4FLEX_E_OUTPUT_TOO_SMALL;   // compress.rs:338-340
    if (in_len >= 65535u) t->large = 1;                        // compress.rs:752-754: transparently upgraded, never downgraded
    lz4flex_ctx* dc = nullptr;
    int rc = default_ctx(&dc);
    if (rc) return rc;
    t->ctx->comp_mode = dc->comp_mode;                         // the thread's encoder choice (throughput / reference-exact)
    t->ctx->comp_variant = dc->comp_variant;
    const uint64_t off0 = 0;
    const uint32_t len = (uint32_t)in_len, cap = (uint32_t)std::min<size_t>(out_cap, 0xFFFFFFFFull);
    // Large: HashTable4K, cleared, stream offset 0 == the frame encoder's first block (LZ4FLEX_BLOCK_FRAME_FIRST);
    // Small: what compress_into picks for inputs below 65 535 bytes
    const uint32_t flags = t->large ? LZ4FLEX_BLOCK_FRAME_FIRST : LZ4FLEX_BLOCK_DEFAULT;
    uint32_t olen = 0;
    int32_t st = 0;
    static const uint8_t empty = 0;
    rc = run_host_batch(t->ctx, true, in ? in : &empty, &off0, &len, &flags, 1, out, &off0, &cap, &olen, &st, nullptr, nullptr);
    if (rc) return rc;
    if (st) return -(int64_t)st;
    return (int64_t)olen;
}

// compress_prepend_size_with_dict, src/block/compress.rs:692-694: LE u32 length, then compress_into_with_dict's block
int64_t lz4flex_compress_prepend_size_with_dict(const uint8_t* in, size_t in_len, uint8_t* out, size_t out_cap, const uint8_t* dict,
                                                size_t dict_len) {
    if (in_len > 0x7FFFFFFFull) return -LZ4FLEX_E_INVALID_ARG;
    if (out_cap < 4 || out_cap - 4 < lz4flex_get_maximum_output_size(in_len)) return -LZ4FLEX_E_OUTPUT_TOO_SMALL;
    const uint32_t n = (uint32_t)in_len;
    out[0] = (uint8_t)n; out[1] = (uint8_t)(n >> 8); out[2] = (uint8_t)(n >> 16); out[3] = (uint8_t)(n >> 24);
    // compress_into_vec_with_dict: dictionaries of <= 3 bytes are ignored (compress.rs:626-628)
    const int64_t r = dict_len <= 3 ? lz4flex_compress_into(in, in_len, out + 4, out_cap - 4)
                                    : lz4flex_compress_into_with_dict(in, in_len, out + 4, out_cap - 4, dict, dict_len);
    return r < 0 ? r : r + 4;
}

int64_t lz4flex_uncompressed_size(const uint8_t* in, size_t in_len) {
    if (in_len < 4) return -LZ4FLEX_E_EXPECTED_ANOTHER_BYTE;   // mod.rs:152
    return (int64_t)((uint32_t)in[0] | ((uint32_t)in[1] << 8) | ((uint32_t)in[2] << 16) | ((uint32_t)in[3] << 24));
}

int64_t lz4flex_decompress_size_prepended(const uint8_t* in, size_t in_len, uint8_t* out, size_t out_cap,
                                          lz4flex_err_detail* detail) {
    const int64_t sz = lz4flex_uncompressed_size(in, in_len);
    if (sz < 0) return sz;
    if ((uint64_t)sz > out_cap) return -LZ4FLEX_E_INVALID_ARG;   // the Vec variant allocates `sz`; here the caller must
    // decompress.rs:493-496 -> decompress(input, uncompressed_size): capacity is exactly the prefix
    return decompress_common(in + 4, in_len - 4, out, (size_t)sz, nullptr, 0, false, detail);
}

// decompress_size_prepended_with_dict, src/block/decompress.rs:521-527
int64_t lz4flex_decompress_size_prepended_with_dict(const uint8_t* in, size_t in_len, uint8_t* out, size_t out_cap, const uint8_t* dict,
                                                    size_t dict_len, lz4flex_err_detail* detail) {
    const int64_t sz = lz4flex_uncompressed_size(in, in_len);
    if (sz < 0) return sz;
    if ((uint64_t)sz > out_cap) return -LZ4FLEX_E_INVALID_ARG;   // the Vec variant allocates `sz`; here the caller must
    return decompress_common(in + 4, in_len - 4, out, (size_t)sz, dict, dict_len, true, detail);
}

}  // extern "C"
