// lz4_decompress_split.hip -- batched LZ4 block decoder, PARSER / COPIER split ("v5").
//
// Same contract as lz4_decompress.hip (reference src/block/decompress.rs:201-449: result bytes, byte count,
// error variant and OutputTooSmall{expected,actual}, unsafe-flavour check order), blocks without dictionary /
// prefix.  What the pipelined decoder (lz4_decompress_lds.hip) spends its issue slots on is the token parse:
// 8 lanes own a block and all 8 run the same ~150-instruction parse to produce ONE sequence.  Here the two
// halves of the reference loop run in different wavefronts of a workgroup:
//
//   * PARSER wavefront: ONE LANE PER BLOCK (up to 64 blocks).  A lane walks its block's token chain
//     (token, literal length, offset, match length: decompress.rs:244-332,377-391) with every bounds check of
//     the reference, and pushes {literal source, literal length, match length, offset} records into the
//     block's LDS queue.  It never touches the output.  The compressed stream is kept in a 48-byte REGISTER
//     window per lane (three 16-byte chunks + one chunk in flight), so no memory round trip sits on the chain;
//     the last 48 bytes of a block are staged in LDS (zero padded) so that nothing is read behind the block.
//     Everything that is not a plain sequence (255-chains, errors, the block's last sequence) goes through an
//     exact byte-wise path that follows decompress.rs line by line.
//   * COPIER wavefronts: G = 8 lanes per block as before.  A group pops records and executes them as 32-byte
//     pieces on the LDS output buffer of lz4_decompress_lds.hip (512 B of history, 16 B/lane coalesced
//     write-back): literal pieces and far match pieces are global loads issued three steps before their bytes
//     are needed, near matches are LDS -> LDS.  Steps without a load touch the compressed stream ahead of the
//     parser (one 128-byte line per step), which keeps the parser's chunk loads out of HBM latency.
//
// LDS per block: 2 080 B output buffer + 16 x 16 B queue + 16 B head/tail + 80 B tail copy = 2 432 B;
// 64 blocks = 152 KiB of the CU's 160 KiB.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lz4_device.h"
#include "lz4_split_parser.h"

namespace lz4flex_dev {
namespace v5 {

// =====================================================================================================
// COPIER: G lanes = one block
// =====================================================================================================
struct Copier {
    const uint8_t* gin;
    uint8_t* gout;
    lds_u8* lout;         // the block's LDS output buffer
    Queue q;
    uint32_t g;
    uint32_t ilen;
    uint32_t op, L0, F;   // LDS output holds positions [L0, op); [0, F) is written back
    uint32_t lit_src, lit_rem, ml_rem, moff;
    uint32_t head;
    uint32_t pf_next;     // next compressed position whose line has not been touched yet
    uint32_t blocked, done;
    enum : uint32_t { K_NONE = 0, K_MAINT = 1, K_RARE = 2, K_CAREFUL = 3, K_FINISH = 4 };
    struct Slot { uint32_t n, dst, msrc, glob, v; };

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

    // Front end: pop a record when the current one is finished, cut one piece, issue its global load.
    __device__ __forceinline__ void fe_step(Slot& s) {
        const bool active = (done | blocked) == 0u;
        const uint32_t qt = q.tail();
        const u32x4 e = q.get(head);
        const bool boundary = (lit_rem | ml_rem) == 0u;
        const bool pop = active && boundary && qt != head;
        lit_src = pop ? e.x : lit_src;
        lit_rem = pop ? e.y : lit_rem;
        ml_rem = pop ? e.z : ml_rem;
        moff = pop ? (e.w & 0xFFFFu) : moff;
        head = pop ? head + 1u : head;
        if (pop && g == 0u) q.set_head(head);
        const bool special = pop && (e.w & (F_FIN | F_CAREFUL)) != 0u;
        const uint32_t special_kind = (e.w & F_FIN) ? (uint32_t)K_FINISH : (uint32_t)K_CAREFUL;
        const bool space_ok = out_space() >= 64u;
        const bool go = active && !special;
        const bool want_l = go && lit_rem != 0u;
        const bool want_m = go && lit_rem == 0u && ml_rem != 0u;
        const bool do_l = want_l && space_ok;
        const bool rare = want_m && space_ok && moff < 4u;
        const bool do_m = want_m && space_ok && moff >= 4u;
        const bool maint = (want_l || want_m) && !space_ok;
        const uint32_t ln = lit_rem < PIECE ? lit_rem : PIECE;
        const uint32_t pm = moff >= PIECE ? PIECE : (moff & ~3u);
        const uint32_t mn = ml_rem < pm ? ml_rem : pm;
        const uint32_t n = do_l ? ln : (do_m ? mn : 0u);
        const uint32_t msrc = op - moff;
        const bool far = do_m && msrc < L0;
        const bool glob = do_l || far;
        // a step without a load touches the next line of the compressed stream ahead of the parser
        const uint32_t pf_pos = pf_next + 16u * g;
        const bool pf = !glob && pf_next < lit_src + PF_AHEAD && pf_pos + 4u <= ilen;
        const bool lane_in = 4u * g < n;   // a literal piece never reads behind lit_end + 3 (<= ilen for plain records)
        const uint8_t* addr = (do_l && lane_in) ? gin + lit_src + 4u * g : (far ? gout + msrc + 4u * g : (pf ? gin + pf_pos : g_pad));
        s.v = ld32(addr);   // exactly one load per step and lane: exact vmcnt bookkeeping
        pf_next = (!glob && pf_next < lit_src + PF_AHEAD) ? pf_next + 16u * G : pf_next;
        s.n = n;
        s.dst = op - L0;
        s.msrc = msrc - L0;
        s.glob = glob ? 1u : 0u;
        lit_src = do_l ? lit_src + ln : lit_src;
        lit_rem = do_l ? lit_rem - ln : lit_rem;
        ml_rem = do_m ? ml_rem - mn : ml_rem;
        op += n;
        blocked = special ? special_kind : (rare ? (uint32_t)K_RARE : (maint ? (uint32_t)K_MAINT : blocked));
    }
    __device__ __forceinline__ void be_step(const Slot& s) {
        if (4u * g < s.n) {
            uint32_t x = s.v;
            if (!s.glob) x = ld32o(s.msrc + 4u * g);
            st32l(s.dst + 4u * g, x);
        }
    }
    __device__ __forceinline__ void service() {
        if (done) return;
        if (out_space() < FLUSH_AT) flush_slide();
        if (blocked == K_RARE) {
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
        Slot s0, s1, s2, s3;
        for (;;) {
            s1.n = 0u; s1.dst = 0u; s1.msrc = 0u; s1.glob = 0u; s1.v = 0u;
            s2 = s1; s3 = s1;
            do {
                const uint32_t before = head + op;
                fe_step(s0); be_step(s1);
                fe_step(s1); be_step(s2);
                fe_step(s2); be_step(s3);
                fe_step(s3); be_step(s0);
                if (!__any(head + op != before)) __builtin_amdgcn_s_sleep(2);   // nothing to do in the whole wave: yield issue slots
            } while (!__any(blocked != K_NONE));
            be_step(s1); be_step(s2); be_step(s3);
            service();
            if (__all(done != 0u)) break;
        }
    }
};

// NB blocks per workgroup: NB/8 copier wavefronts followed by the parser wavefront (NB lanes in use)
template <uint32_t NB>
__global__ void __launch_bounds__(64 * (NB / 8 + 1)) lz4_decompress_split_kernel(DecompressArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t dyn_lds[];
    lds_u8* lds = (lds_u8*)dyn_lds;
    constexpr uint32_t CW = NB * G / 64u;
    const uint32_t wave = threadIdx.x / 64u;
    const uint32_t lane = threadIdx.x % 64u;
    const uint32_t first = blockIdx.x * NB;
    if (wave < CW) {
        // ---- copier group: set up the block's queue and tail copy, then run
        const uint32_t j = wave * (64u / G) + lane / G;
        const uint32_t b = first + j;
        const bool valid = b < a.n;
        Copier c;
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
        c.lit_src = 0u; c.lit_rem = 0u; c.ml_rem = 0u; c.moff = 0u; c.head = 0u; c.pf_next = 0u;
        c.blocked = Copier::K_NONE;
        c.done = valid ? 0u : 1u;
        __syncthreads();
        c.run();
    } else {
        // ---- parser: lane j owns block first + j
        const uint32_t j = lane;
        const uint32_t b = first + j;
        const bool valid = j < NB && b < a.n;
        Parser p;
        p.q.blk = lds + (j < NB ? j : 0u) * BLK_LDS;
        p.gin = valid ? a.in_base + a.in_off[b] : g_pad;
        p.A = (uint32_t)(reinterpret_cast<uintptr_t>(p.gin) & 3u);
        p.gal = p.gin - p.A;
        p.ilen = valid ? a.in_len[b] : 0u;
        p.cap = valid ? a.out_cap[b] : 0u;
        p.tstart = p.ilen > TAILB ? p.ilen - TAILB : 0u;
        p.ip = 0u; p.op = 0u; p.need_off = 0u; p.mlc_saved = 0u; p.qtail = 0u;
        p.status = 0; p.expected = 0u;
        p.done = valid ? 0u : 1u;
        p.base = 0u;
#ifdef LZ4FLEX_SPLIT_DEBUG
        p.dbg_ip = 0xFFFFFFFFu; p.dbg_w0 = 0u; p.dbg_w1 = 0u; p.dbg_kb = 0u;
#endif
        p.C0 = *reinterpret_cast<const u32x4*>(p.chunk_addr(0u));
        p.C1 = *reinterpret_cast<const u32x4*>(p.chunk_addr(16u));
        p.C2 = *reinterpret_cast<const u32x4*>(p.chunk_addr(32u));
        p.N = *reinterpret_cast<const u32x4*>(p.chunk_addr(48u));
        __syncthreads();
        __builtin_amdgcn_s_setprio(3);
        if (valid && p.ilen == 0u) p.fail(LZ4FLEX_DEV_E_EXPECTED_ANOTHER_BYTE);   // :207-209
        while (!__all(p.done != 0u)) p.step();
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

template <uint32_t NB>
static hipError_t launch_nb(const DecompressArgs& a, hipStream_t s) {
    const uint32_t grid = (a.n + NB - 1u) / NB;
    const size_t lds = (size_t)NB * BLK_LDS;
    auto kern = lz4_decompress_split_kernel<NB>;
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
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64u * (NB / 8u + 1u)), lds, s, a);
    return hipGetLastError();
}

}  // namespace v5

// blocks_per_wg: 8, 16, 32 or 64; 0 = the largest that still gives every CU a workgroup
hipError_t launch_decompress_split(const DecompressArgs& a, hipStream_t s, int blocks_per_wg) {
    if (a.n == 0u) return hipSuccess;
    if (a.dict_base != nullptr || a.out_pos != nullptr) return hipErrorInvalidValue;   // dictionary / prefix: v1 kernel
    if (blocks_per_wg == 0) blocks_per_wg = a.n >= 64u * 256u ? 64 : (a.n >= 32u * 256u ? 32 : (a.n >= 16u * 256u ? 16 : 8));
    switch (blocks_per_wg) {
        case 64: return v5::launch_nb<64>(a, s);
        case 32: return v5::launch_nb<32>(a, s);
        case 16: return v5::launch_nb<16>(a, s);
        case 8: return v5::launch_nb<8>(a, s);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace lz4flex_dev
