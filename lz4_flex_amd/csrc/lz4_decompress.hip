// lz4_decompress.hip -- batched LZ4 block decoder for MI355X (gfx950, wave64).
//
// Replaces the per-block call lz4_flex::block::decompress_into /
// decompress_internal (reference src/block/decompress.rs:201-449) for MANY independent
// blocks at once.  Semantics (result bytes, byte count, error variant and OutputTooSmall
// {expected, actual}) follow the unsafe flavour's check order; see DESIGN.md "Decoder".
//
// Work decomposition (MI355X-first, not a translation of the CPU loop):
//   * one GROUP of G lanes (G = 8/16/32/64, a slice of a 64-wide wavefront) owns one block;
//     a 256-thread workgroup therefore decodes 256/G blocks, 64/G per wave.
//   * every lane of a group parses the token stream redundantly from the same addresses
//     (the loads coalesce to one request per group), so no cross-lane traffic is needed for
//     control; the byte copies (literals, matches) are split across the group's lanes in
//     dword units.
//   * overlapping matches (offset < match length) use the periodic form
//     out[op+i] = out[op-offset + (i mod offset)], which only reads bytes written by EARLIER
//     sequences, so the copy stays lane-parallel.
//   * a match reads bytes other lanes of the SAME wave stored earlier; CDNA issues a wave's
//     vector-memory instructions in order through one TA/TCP path, so a later load observes
//     the earlier store; the wavefront-scope fences below pin that order for the compiler.
//
// HBM traffic per block is the algorithmic minimum: compressed bytes read once (token windows
// and literal sources hit the same L1/L2 lines), uncompressed bytes written once; match
// sources are re-read from L2 (recently written lines).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lz4_device.h"

namespace lz4flex_dev {

__device__ __forceinline__ uint32_t ld32(const uint8_t* p) {
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}
__device__ __forceinline__ uint64_t ld64(const uint8_t* p) {
    uint64_t v;
    __builtin_memcpy(&v, p, 8);
    return v;
}
__device__ __forceinline__ void st32(uint8_t* p, uint32_t v) { __builtin_memcpy(p, &v, 4); }
__device__ __forceinline__ void st64(uint8_t* p, uint64_t v) { __builtin_memcpy(p, &v, 8); }

// order this wave's earlier stores before its later loads (no ISA cost at wavefront scope)
__device__ __forceinline__ void wave_fence() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); }

// Copy `len` bytes src->dst with the group's lanes, 4 bytes per lane per step.
// Requires: the first 4*G-byte step never reads bytes written by the same step, i.e. either
// src is a different buffer or dst - src >= 4*G.  `wild` allows whole-dword reads and writes
// up to 3 bytes past `len` (caller guarantees both buffers have that slack).
template <int G>
__device__ __forceinline__ void group_copy(uint8_t* dst, const uint8_t* src, uint32_t len, uint32_t g, bool wild) {
    for (uint32_t i = 4u * g; i < len; i += 4u * G) {
        if (wild || i + 4u <= len) {
            st32(dst + i, ld32(src + i));
        } else {
            for (uint32_t k = i; k < len; ++k) dst[k] = src[k];
        }
    }
}

// Overlapping (offset < 4*G) match: periodic read from the `offset` bytes before dst.
template <int G>
__device__ __forceinline__ void group_copy_periodic(uint8_t* dst, uint32_t offset, uint32_t len, uint32_t g) {
    const uint8_t* pat = dst - offset;
    if (offset == 1u) {
        // run of one byte: build the dword once
        const uint32_t b = pat[0];
        const uint32_t v = b * 0x01010101u;
        for (uint32_t i = 4u * g; i < len; i += 4u * G) {
            if (i + 4u <= len) st32(dst + i, v);
            else for (uint32_t k = i; k < len; ++k) dst[k] = (uint8_t)b;
        }
        return;
    }
    uint32_t idx = (4u * g) % offset;         // pattern index of this lane's first byte
    const uint32_t step = (4u * G) % offset;  // advance per iteration
    for (uint32_t i = 4u * g; i < len; i += 4u * G) {
        uint32_t j = idx;
        uint32_t v = 0;
        const uint32_t nb = (len - i < 4u) ? (len - i) : 4u;
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) {
            if (k < nb) {
                v |= (uint32_t)pat[j] << (8u * k);
                j = (j + 1u == offset) ? 0u : j + 1u;
            }
        }
        if (nb == 4u) st32(dst + i, v);
        else for (uint32_t k = 0; k < nb; ++k) dst[i + k] = (uint8_t)(v >> (8u * k));
        idx += step;
        if (idx >= offset) idx -= offset;
    }
}

// Decode one block with a group of G lanes. Returns the status code; *produced = bytes written.
template <int G, bool USE_DICT>
__device__ __forceinline__ int32_t decode_block(const uint8_t* __restrict__ in, uint32_t ilen, uint8_t* out,
                                                uint32_t out_pos, uint32_t cap, const uint8_t* __restrict__ dict,
                                                uint32_t dict_len, uint32_t g, uint32_t* produced,
                                                uint64_t* det_expected) {
    if (ilen == 0u) return LZ4FLEX_DEV_E_EXPECTED_ANOTHER_BYTE;   // decompress.rs:207-209
    if (!USE_DICT) dict_len = 0u;
    uint32_t ip = 0u, op = out_pos;
    for (;;) {
        // ---- token + (when far from the input end) an 8-byte window of what follows --------
        const bool win = ip + 8u <= ilen;
        uint64_t w;
        if (win) w = ld64(in + ip);
        else w = in[ip];
        const uint32_t token = (uint32_t)w & 0xFFu;
        ip += 1u;
        uint32_t lit = token >> 4;
        // ---- literals (decompress.rs:334-362) ----------------------------------------------
        if (lit != 0u) {
            if (lit == 15u) {
                uint64_t acc = lit;   // usize in the reference: > 16 MiB of 0xFF length bytes must not wrap a 32-bit sum
                for (;;) {   // read_integer_ptr, decompress.rs:126-157
                    if (ip >= ilen) return LZ4FLEX_DEV_E_EXPECTED_ANOTHER_BYTE;
                    const uint32_t e = in[ip++];
                    acc += e;
                    if (e != 0xFFu) break;
                }
                if (acc > (uint64_t)(ilen - ip)) return LZ4FLEX_DEV_E_LITERAL_OUT_OF_BOUNDS;
                lit = (uint32_t)acc;
            }
            if (lit > ilen - ip) return LZ4FLEX_DEV_E_LITERAL_OUT_OF_BOUNDS;
            if (lit > cap - op) {
                *det_expected = (uint64_t)op + lit;
                return LZ4FLEX_DEV_E_OUTPUT_TOO_SMALL;
            }
            if (win && lit <= 7u && cap - op >= 8u) {
                // the literals are bytes 1..lit of the window: one 8-byte store by lane 0; the
                // bytes past `lit` are rewritten by what follows (same-wave stores stay ordered)
                if (g == 0u) st64(out + op, w >> 8);
            } else {
                const bool wild = (lit + 3u <= ilen - ip) && (lit + 3u <= cap - op);
                group_copy<G>(out + op, in + ip, lit, g, wild);
            }
            op += lit;
            ip += lit;
        }
        if (ip >= ilen) break;   // decompress.rs:366-368: normal end, last sequence is literal-only
        // ---- offset + match length (decompress.rs:373-391) ---------------------------------
        if (ilen - ip < 2u) return LZ4FLEX_DEV_E_EXPECTED_ANOTHER_BYTE;
        uint32_t offset;
        if (win && lit <= 5u) offset = (uint32_t)(w >> (8u * (1u + lit))) & 0xFFFFu;
        else offset = (uint32_t)in[ip] | ((uint32_t)in[ip + 1u] << 8);
        ip += 2u;
        if (offset == 0u) return LZ4FLEX_DEV_E_OFFSET_ZERO;
        uint32_t ml = 4u + (token & 15u);
        if (ml == 19u) {
            uint64_t acc = ml;
            for (;;) {
                if (ip >= ilen) return LZ4FLEX_DEV_E_EXPECTED_ANOTHER_BYTE;
                const uint32_t e = in[ip++];
                acc += e;
                if (e != 0xFFu) break;
            }
            if (acc > 0xFFFFFFFFull) {   // (same order as below: the offset check comes first)
                if (offset > op + dict_len) return LZ4FLEX_DEV_E_OFFSET_OUT_OF_BOUNDS;
                *det_expected = (uint64_t)op + acc;
                return LZ4FLEX_DEV_E_OUTPUT_TOO_SMALL;
            }
            ml = (uint32_t)acc;
        }
        // ---- bounds (decompress.rs:398-408; unsafe-flavour order) --------------------------
        if (offset > op + dict_len) return LZ4FLEX_DEV_E_OFFSET_OUT_OF_BOUNDS;
        if (ml > cap - op) {
            *det_expected = (uint64_t)op + ml;
            return LZ4FLEX_DEV_E_OUTPUT_TOO_SMALL;
        }
        // ---- external dictionary part (decompress.rs:410-426, copy_from_dict :85-109) ------
        if (USE_DICT && offset > op) {
            const uint32_t dict_offset = dict_len + op - offset;
            const uint32_t n = (ml < dict_len - dict_offset) ? ml : (dict_len - dict_offset);
            group_copy<G>(out + op, dict + dict_offset, n, g, false);
            op += n;
            if (n == ml) {
                if (ip >= ilen) return LZ4FLEX_DEV_E_EXPECTED_ANOTHER_BYTE;
                continue;
            }
            ml -= n;
        }
        // ---- match copy (duplicate(), decompress.rs:11-82: byte-serial forward semantics) --
        wave_fence();
        if (offset >= 4u * G) {
            const bool wild = (ml + 3u <= cap - op);
            group_copy<G>(out + op, out + op - offset, ml, g, wild);
        } else {
            group_copy_periodic<G>(out + op, offset, ml, g);
        }
        wave_fence();
        op += ml;
        if (ip >= ilen) return LZ4FLEX_DEV_E_EXPECTED_ANOTHER_BYTE;   // decompress.rs:439-443
    }
    *produced = op - out_pos;
    return 0;
}

template <int G, bool USE_DICT>
__global__ void __launch_bounds__(256) lz4_decompress_blocks_kernel(DecompressArgs a) {
    const uint32_t tid = blockIdx.x * 256u + threadIdx.x;
    const uint32_t b = tid / G;
    const uint32_t g = tid % G;
    if (b >= a.n) return;
    if (a.only_status != 0 && a.status[b] != a.only_status) return;   // second pass behind the wave decoder: marked blocks only
    const uint8_t* in = a.in_base + a.in_off[b];
    uint8_t* out = a.out_base + a.out_off[b];
    const uint32_t ilen = a.in_len[b];
    const uint32_t cap = a.out_cap[b];
    const uint8_t* dict = nullptr;
    uint32_t dict_len = 0u, out_pos = 0u;
    if (USE_DICT) {
        dict = a.dict_base + a.dict_off[b];
        dict_len = a.dict_len[b];
    }
    if (a.out_pos) out_pos = a.out_pos[b];
    uint32_t produced = 0u;
    uint64_t expected = 0u;
    const int32_t st = decode_block<G, USE_DICT>(in, ilen, out, out_pos, cap, dict, dict_len, g, &produced, &expected);
    if (g == 0u) {
        a.status[b] = st;
        a.out_len[b] = st == 0 ? produced : 0u;
        if (a.detail) {
            a.detail[2u * b] = st == LZ4FLEX_DEV_E_OUTPUT_TOO_SMALL ? expected : 0u;
            a.detail[2u * b + 1u] = st == LZ4FLEX_DEV_E_OUTPUT_TOO_SMALL ? (uint64_t)cap : 0u;
        }
    }
}

// Second pass behind a CHAINED batch (Linked frames: block i's prefix is what blocks 0..i-1 of the batch write).  The marked blocks
// of such a batch cannot be decoded side by side -- block k + 1 would read block k's last 64 KiB while block k is being written
// again (status 0, wrong bytes) -- so ONE wavefront decodes them one after the other, in chain order.  Slow, and only
// ever busy when a first-pass block gave up for a non-error reason (a time-sliced or oversubscribed GPU) or the frame is corrupt.
// Several chains in one batch (DecompressArgs::chain_prev: N Linked frames side by side) are independent of each other: the grid then
// holds several wavefronts, and the chain whose first block is r belongs to wavefront r mod gridDim.x -- every chain is still decoded by
// ONE wavefront in index order, different chains side by side (the serial second pass).
// The first block of every marked block's chain, once (a thread per block follows the predecessors: indices fall), into the batch's
// "done" words -- free behind the first pass.  Round 5 had every wavefront of the second pass walk every marked block's chain
// (O(wavefronts x marked x depth) dependent loads).
__global__ void lz4_decompress_chain_roots_kernel(DecompressArgs a) {
    const uint32_t bi = blockIdx.x * blockDim.x + threadIdx.x;
    if (bi >= a.n || a.status[bi] != a.only_status) return;
    uint32_t r = bi;
    for (uint32_t p = a.chain_prev[r]; p < r; p = a.chain_prev[r]) r = p;
    a.chain_done[bi] = r;
}
__global__ void __launch_bounds__(64) lz4_decompress_chain_redo_kernel(DecompressArgs a) {
    constexpr int G = 16;
    const uint32_t lane = threadIdx.x;
    for (uint32_t b0 = 0u; b0 < a.n; b0 += 64u) {
        const uint32_t bi = b0 + lane;
        bool marked = bi < a.n && a.status[bi] == a.only_status;
        if (marked && a.chain_prev != nullptr && gridDim.x > 1u) marked = a.chain_done[bi] % gridDim.x == blockIdx.x;    // (its chain's first block: lz4_decompress_chain_roots_kernel)
        uint64_t m = __builtin_amdgcn_ballot_w64(marked);
        while (m != 0ull) {
            const uint32_t b = b0 + (uint32_t)__builtin_ctzll(m);
            m &= m - 1ull;
            if (lane < (uint32_t)G) {
                uint32_t produced = 0u;
                uint64_t expected = 0u;
                const uint32_t cap = a.out_cap[b];
                const int32_t st = decode_block<G, false>(a.in_base + a.in_off[b], a.in_len[b], a.out_base + a.out_off[b], a.out_pos ? a.out_pos[b] : 0u,
                                                          cap, nullptr, 0u, lane, &produced, &expected);
                if (lane == 0u) {
                    a.status[b] = st;
                    a.out_len[b] = st == 0 ? produced : 0u;
                    if (a.detail) {
                        a.detail[2u * b] = st == LZ4FLEX_DEV_E_OUTPUT_TOO_SMALL ? expected : 0u;
                        a.detail[2u * b + 1u] = st == LZ4FLEX_DEV_E_OUTPUT_TOO_SMALL ? (uint64_t)cap : 0u;
                    }
                }
            }
            __threadfence();      // the next block of the chain reads what this one wrote
        }
    }
}

hipError_t launch_decompress_chain_redo(const DecompressArgs& a, hipStream_t s) {
    if (a.n == 0u) return hipSuccess;
    if (a.dict_base != nullptr || a.only_status == 0) return hipErrorInvalidValue;
    const uint32_t grid = a.chain_prev != nullptr && a.chain_done != nullptr ? (a.n_chains > 1u ? (a.n_chains < 1024u ? a.n_chains : 1024u) : 64u) : 1u;
    if (grid > 1u) hipLaunchKernelGGL(lz4_decompress_chain_roots_kernel, dim3((a.n + 255u) / 256u), dim3(256), 0, s, a);
    hipLaunchKernelGGL(lz4_decompress_chain_redo_kernel, dim3(grid), dim3(64), 0, s, a);
    return hipGetLastError();
}

template <int G, bool USE_DICT>
static hipError_t launch_g(const DecompressArgs& a, hipStream_t s) {
    const uint32_t blocks_per_wg = 256u / G;
    const uint32_t grid = (a.n + blocks_per_wg - 1u) / blocks_per_wg;
    hipLaunchKernelGGL((lz4_decompress_blocks_kernel<G, USE_DICT>), dim3(grid), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_decompress(const DecompressArgs& a, int lanes_per_block, hipStream_t s) {
    if (a.n == 0u) return hipSuccess;
    const bool d = a.dict_base != nullptr;
    switch (lanes_per_block) {
        case 16: return d ? launch_g<16, true>(a, s) : launch_g<16, false>(a, s);
#ifdef LZ4FLEX_ALL_VARIANTS   // the other group widths measured slower at every batch size (round 1); variant builds only
        case 8: return d ? launch_g<8, true>(a, s) : launch_g<8, false>(a, s);
        case 32: return d ? launch_g<32, true>(a, s) : launch_g<32, false>(a, s);
        case 64: return d ? launch_g<64, true>(a, s) : launch_g<64, false>(a, s);
#endif
        default: return hipErrorInvalidValue;
    }
}

}  // namespace lz4flex_dev
