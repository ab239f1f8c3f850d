// frame_kernels.hip -- device-side assembly of LZ4 frame segments for the sharded (multi-GPU) frame path.
//
// What it replaces: the per-block bookkeeping of FrameEncoder::write_block (src/frame/compress.rs:261-371) for a rank's
// contiguous block range whose blocks were compressed by the batched block encoder: the 4-byte block header (length,
// high bit = stored raw), the store-raw rule (:301-306: a block that did not shrink is stored uncompressed), the payload
// and the optional block checksum (:313-316), laid out back to back.  Round 1 did this in a Python loop with three tiny
// tensor copies per block; here it is three launches for any number of blocks:
//   1. sizes + exclusive prefix sum (one workgroup),
//   2. header + payload copy, one workgroup per block, 16 B per lane where source and destination allow it,
//   3. (block checksums) XXH32 of every payload (xxh32_kernel.hip), then 4 bytes per block.
// lz4flex_copy_batch_device is the decode-side companion: n independent byte ranges copied in one launch (blocks stored
// raw go straight to their place in the output; compressed ones are decoded there by the batched decoder).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lz4_device.h"

namespace lz4flex_dev {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// one workgroup of 1024 threads: per[i] = 4 + min(comp_len, in_len stored raw) (+ 4), seg_off = exclusive sum, seg_off[n] = total
__global__ void __launch_bounds__(1024) frame_sizes_scan_kernel(const uint32_t* __restrict__ in_len, const uint32_t* __restrict__ comp_len,
                                                                uint32_t n, uint32_t tail, uint64_t* __restrict__ seg_off) {
    __shared__ uint64_t part[1024];
    const uint32_t t = threadIdx.x;
    const uint32_t per_thread = (n + 1023u) / 1024u;
    const uint32_t lo = t * per_thread, hi = lo + per_thread < n ? lo + per_thread : n;
    uint64_t s = 0;
    for (uint32_t i = lo; i < hi; ++i) {
        const uint32_t c = comp_len[i], u = in_len[i];
        s += 4ull + (c >= u ? u : c) + tail;
    }
    part[t] = s;
    __syncthreads();
    for (uint32_t d = 1; d < 1024u; d <<= 1) {          // Hillis-Steele inclusive scan of the 1024 partial sums
        const uint64_t v = t >= d ? part[t - d] : 0ull;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    uint64_t o = t == 0u ? 0ull : part[t - 1u];
    for (uint32_t i = lo; i < hi; ++i) {
        seg_off[i] = o;
        const uint32_t c = comp_len[i], u = in_len[i];
        o += 4ull + (c >= u ? u : c) + tail;
    }
    if (t == 1023u) seg_off[n] = part[1023];
}

__device__ __forceinline__ void copy_range(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src, uint64_t n, uint32_t t, uint32_t nt) {
    if ((((uintptr_t)dst | (uintptr_t)src) & 15u) == 0u) {
        const uint64_t nv = n / 16u;
        for (uint64_t i = t; i < nv; i += nt) reinterpret_cast<u32x4*>(dst)[i] = reinterpret_cast<const u32x4*>(src)[i];
        for (uint64_t i = nv * 16u + t; i < n; i += nt) dst[i] = src[i];
    } else if (((uintptr_t)dst & 15u) == ((uintptr_t)src & 15u)) {
        const uint64_t head = (16u - ((uintptr_t)dst & 15u)) & 15u;
        const uint64_t h = head < n ? head : n;
        for (uint64_t i = t; i < h; i += nt) dst[i] = src[i];
        const uint64_t nv = (n - h) / 16u;
        for (uint64_t i = t; i < nv; i += nt) reinterpret_cast<u32x4*>(dst + h)[i] = reinterpret_cast<const u32x4*>(src + h)[i];
        for (uint64_t i = h + nv * 16u + t; i < n; i += nt) dst[i] = src[i];
    } else {
        // different phase: aligned 16-byte stores, unaligned loads (global memory takes any alignment)
        const uint64_t head = (16u - ((uintptr_t)dst & 15u)) & 15u;
        const uint64_t h = head < n ? head : n;
        for (uint64_t i = t; i < h; i += nt) dst[i] = src[i];
        const uint64_t nv = (n - h) / 16u;
        for (uint64_t i = t; i < nv; i += nt) {
            u32x4 v;
            __builtin_memcpy(&v, src + h + 16u * i, 16);
            reinterpret_cast<u32x4*>(dst + h)[i] = v;
        }
        for (uint64_t i = h + nv * 16u + t; i < n; i += nt) dst[i] = src[i];
    }
}

// block b: [u32 header][payload]; the checksum slot behind it is filled by frame_checksum_kernel
__global__ void __launch_bounds__(256) frame_assemble_kernel(const uint8_t* __restrict__ src_base, const uint64_t* __restrict__ src_off,
                                                             const uint32_t* __restrict__ in_len, const uint8_t* __restrict__ comp_base,
                                                             const uint64_t* __restrict__ comp_off, const uint32_t* __restrict__ comp_len,
                                                             uint32_t n, const uint64_t* __restrict__ seg_off, uint8_t* __restrict__ seg,
                                                             uint64_t* __restrict__ pay_off, uint32_t* __restrict__ pay_len) {
    const uint32_t b = blockIdx.x;
    if (b >= n || seg_off[b] == ~0ull) return;                      // (~0: a block of a frame that does not fit its buffer, frame_many_layout_kernel)
    const uint32_t c = comp_len[b], u = in_len[b];
    const bool raw = c >= u;                                         // frame/compress.rs:301-306
    const uint32_t size = raw ? u : c;
    uint8_t* d = seg + seg_off[b];
    if (threadIdx.x == 0u) {
        const uint32_t w = raw ? (u | 0x80000000u) : c;              // BlockInfo::write, frame/header.rs:108-124
        d[0] = (uint8_t)w; d[1] = (uint8_t)(w >> 8); d[2] = (uint8_t)(w >> 16); d[3] = (uint8_t)(w >> 24);
        if (pay_off) { pay_off[b] = seg_off[b] + 4ull; pay_len[b] = size; }
    }
    copy_range(d + 4, raw ? src_base + src_off[b] : comp_base + comp_off[b], size, threadIdx.x, 256u);
}

__global__ void frame_checksum_kernel(const uint64_t* __restrict__ pay_off, const uint32_t* __restrict__ pay_len, const uint32_t* __restrict__ sums,
                                      uint32_t n, uint8_t* __restrict__ seg) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n) return;
    uint8_t* d = seg + pay_off[b] + pay_len[b];
    const uint32_t w = sums[b];
    d[0] = (uint8_t)w; d[1] = (uint8_t)(w >> 8); d[2] = (uint8_t)(w >> 16); d[3] = (uint8_t)(w >> 24);
}

// (many frames: a block the layout kernel dropped -- dst_off ~0 -- has no checksum slot)
__global__ void frame_checksum_many_kernel(const uint64_t* __restrict__ dst_off, const uint64_t* __restrict__ pay_off, const uint32_t* __restrict__ pay_len,
                                           const uint32_t* __restrict__ sums, uint32_t n, uint8_t* __restrict__ seg) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n || dst_off[b] == ~0ull) return;
    uint8_t* d = seg + pay_off[b] + pay_len[b];
    const uint32_t w = sums[b];
    d[0] = (uint8_t)w; d[1] = (uint8_t)(w >> 8); d[2] = (uint8_t)(w >> 16); d[3] = (uint8_t)(w >> 24);
}

__global__ void __launch_bounds__(256) copy_batch_kernel(const uint8_t* __restrict__ src_base, const uint64_t* __restrict__ src_off,
                                                         const uint32_t* __restrict__ len, uint8_t* __restrict__ dst_base,
                                                         const uint64_t* __restrict__ dst_off, uint32_t n) {
    const uint32_t b = blockIdx.x;
    if (b >= n) return;
    copy_range(dst_base + dst_off[b], src_base + src_off[b], len[b], threadIdx.x, 256u);
}

hipError_t launch_frame_assemble(const uint8_t* src_base, const uint64_t* src_off, const uint32_t* in_len, const uint8_t* comp_base,
                                 const uint64_t* comp_off, const uint32_t* comp_len, uint32_t n, int block_checksums, uint8_t* seg,
                                 uint64_t* seg_off, uint64_t* pay_off, uint32_t* pay_len, uint32_t* sums, hipStream_t s) {
    if (n == 0u) return hipMemsetAsync(seg_off, 0, 8, s);
    hipLaunchKernelGGL(frame_sizes_scan_kernel, dim3(1), dim3(1024), 0, s, in_len, comp_len, n, block_checksums ? 4u : 0u, seg_off);
    hipLaunchKernelGGL(frame_assemble_kernel, dim3(n), dim3(256), 0, s, src_base, src_off, in_len, comp_base, comp_off, comp_len, n,
                       (const uint64_t*)seg_off, seg, block_checksums ? pay_off : nullptr, pay_len);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || !block_checksums) return e;
    e = launch_xxh32_batch(seg, pay_off, pay_len, n, 0u, sums, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(frame_checksum_kernel, dim3((n + 255u) / 256u), dim3(256), 0, s, (const uint64_t*)pay_off, (const uint32_t*)pay_len,
                       (const uint32_t*)sums, n, seg);
    return hipGetLastError();
}

// The block-header walk of FrameDecoder::read_block (src/frame/decompress.rs:231-247) over a frame that lives in device
// memory: one thread follows the chain of 4-byte BlockInfo words (each position depends on the previous length) and writes
// payload offset / length-with-the-uncompressed-bit per block.  info[0] = blocks, info[1] = status (0 ok, 1 truncated,
// 2 BlockTooBig, 3 more than max_blocks), info[2..3] = offset behind the EndMark.
__global__ void frame_walk_kernel(const uint8_t* __restrict__ f, uint64_t n, uint32_t hdr, uint32_t tail, uint32_t block_size, uint32_t max_blocks,
                                  uint64_t* __restrict__ off, uint32_t* __restrict__ len, uint32_t* __restrict__ info) {
    if (threadIdx.x != 0u || blockIdx.x != 0u) return;
    uint64_t p = hdr;
    uint32_t k = 0u, st = 0u;
    for (;;) {
        if (p + 4u > n) { st = 1u; break; }
        const uint32_t w = (uint32_t)f[p] | ((uint32_t)f[p + 1] << 8) | ((uint32_t)f[p + 2] << 16) | ((uint32_t)f[p + 3] << 24);
        p += 4u;
        if (w == 0u) break;                                   // EndMark
        const uint32_t ln = w & 0x7FFFFFFFu;
        if (ln > block_size) { st = 2u; break; }
        if (p + ln + tail > n) { st = 1u; break; }
        if (k >= max_blocks) { st = 3u; break; }
        off[k] = p; len[k] = w; k += 1u;
        p += (uint64_t)ln + tail;
    }
    info[0] = k; info[1] = st; info[2] = (uint32_t)p; info[3] = (uint32_t)(p >> 32);
}

hipError_t launch_frame_walk(const uint8_t* f, uint64_t n, uint32_t hdr, uint32_t tail, uint32_t block_size, uint32_t max_blocks, uint64_t* off,
                             uint32_t* len, uint32_t* info, hipStream_t s) {
    hipLaunchKernelGGL(frame_walk_kernel, dim3(1), dim3(64), 0, s, f, n, hdr, tail, block_size, max_blocks, off, len, info);
    return hipGetLastError();
}

// ---- many frames at once (frame_many.cpp: N streams, one frame each; BASELINE configs[4] in the shape with parallelism in it) -----
// Encode side, one thread per stream: where each block of the stream goes ([BlockInfo | payload | (checksum)]* behind the header,
// frame/compress.rs:282-316), the header bytes (prepared by the host: FrameInfo::write), the EndMark and the content checksum
// (frame/compress.rs:209-230).  A frame that does not fit out_cap, or holds a block the encoder failed on, is not written at all.
__global__ void frame_many_layout_kernel(const ManyStream* __restrict__ st, uint32_t n, const uint32_t* __restrict__ in_len,
                                         const uint32_t* __restrict__ comp_len, const int32_t* __restrict__ comp_st,
                                         const uint32_t* __restrict__ content_sum, uint8_t* __restrict__ out_base, uint64_t* __restrict__ dst_off,
                                         uint64_t* __restrict__ pay_off, uint32_t* __restrict__ pay_len, uint64_t* __restrict__ frame_len,
                                         int32_t* __restrict__ verdict) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const ManyStream m = st[s];
    const uint32_t tail = (m.flags & 1u) ? 4u : 0u;
    uint64_t total = m.hdr_len;
    bool failed = false;
    for (uint32_t k = 0; k < m.count; ++k) {
        const uint32_t b = m.first + k, c = comp_len[b], u = in_len[b];
        failed |= comp_st[b] != 0;
        total += 4ull + (c >= u ? u : c) + tail;
    }
    total += 4ull + ((m.flags & 2u) ? 4ull : 0ull);
    const int32_t v = failed ? 1 : (total > m.out_cap ? 2 : 0);
    verdict[s] = v;
    frame_len[s] = v == 0 ? total : 0ull;
    if (v != 0) {
        for (uint32_t k = 0; k < m.count; ++k) { dst_off[m.first + k] = ~0ull; if (pay_len) pay_len[m.first + k] = 0u; if (pay_off) pay_off[m.first + k] = 0ull; }
        return;
    }
    uint8_t* d = out_base + m.out_off;
    for (uint32_t i = 0; i < m.hdr_len; ++i) d[i] = m.hdr[i];
    uint64_t o = m.hdr_len;
    for (uint32_t k = 0; k < m.count; ++k) {
        const uint32_t b = m.first + k, c = comp_len[b], u = in_len[b];
        dst_off[b] = m.out_off + o;
        o += 4ull + (c >= u ? u : c) + tail;
    }
    for (uint32_t i = 0; i < 4u; ++i) d[o + i] = 0u;                 // EndMark
    if (m.flags & 2u) {
        const uint32_t w = content_sum[s];
        d[o + 4] = (uint8_t)w; d[o + 5] = (uint8_t)(w >> 8); d[o + 6] = (uint8_t)(w >> 16); d[o + 7] = (uint8_t)(w >> 24);
    }
}

hipError_t launch_frame_many_assemble(const ManyStream* st, uint32_t n_streams, const uint8_t* src_base, const uint64_t* src_off, const uint32_t* in_len,
                                      const uint8_t* comp_base, const uint64_t* comp_off, const uint32_t* comp_len, const int32_t* comp_st, uint32_t n_blocks,
                                      int block_checksums, const uint32_t* content_sum, uint8_t* out_base, uint64_t* dst_off, uint64_t* pay_off,
                                      uint32_t* pay_len, uint32_t* sums, uint64_t* frame_len, int32_t* verdict, hipStream_t s) {
    if (n_streams == 0u) return hipSuccess;
    hipLaunchKernelGGL(frame_many_layout_kernel, dim3((n_streams + 63u) / 64u), dim3(64), 0, s, st, n_streams, in_len, comp_len, comp_st, content_sum,
                       out_base, dst_off, block_checksums ? pay_off : nullptr, block_checksums ? pay_len : nullptr, frame_len, verdict);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || n_blocks == 0u) return e;
    hipLaunchKernelGGL(frame_assemble_kernel, dim3(n_blocks), dim3(256), 0, s, src_base, src_off, in_len, comp_base, comp_off, comp_len, n_blocks,
                       (const uint64_t*)dst_off, out_base, block_checksums ? pay_off : nullptr, pay_len);
    e = hipGetLastError();
    if (e != hipSuccess || !block_checksums) return e;
    e = launch_xxh32_batch(out_base, pay_off, pay_len, n_blocks, 0u, sums, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(frame_checksum_many_kernel, dim3((n_blocks + 255u) / 256u), dim3(256), 0, s, (const uint64_t*)dst_off, (const uint64_t*)pay_off,
                       (const uint32_t*)pay_len, (const uint32_t*)sums, n_blocks, out_base);
    return hipGetLastError();
}

// Decode side.  The first bytes of every frame, 32 per frame (zero-filled behind a short one): the host parses the headers
// (FrameInfo::read, frame/header.rs:277-373: lz4flex_frame_info_read) without moving the frames.
__global__ void frame_many_heads_kernel(const uint8_t* __restrict__ base, const uint64_t* __restrict__ off, const uint64_t* __restrict__ len, uint32_t n,
                                        uint8_t* __restrict__ heads) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t s = t >> 5, i = t & 31u;
    if (s >= n) return;
    heads[t] = i < len[s] ? base[off[s] + i] : (uint8_t)0;
}

// The block-header walk of FrameDecoder::read_block (frame/decompress.rs:231-247) for n frames, one thread per frame (a frame's
// BlockInfo words are a chain: each position depends on the previous length).  Per frame: table slots [slot, slot + slot_cap) receive
// payload offset (in `base`) and length word; info[8 s ..]: blocks, status (0 ok, 1 truncated, 2 BlockTooBig, 3 more blocks than
// slots), offset behind the EndMark and the content checksum (lo, hi), the stored content checksum.
__global__ void frame_many_walk_kernel(const uint8_t* __restrict__ base, const ManyFrame* __restrict__ fr, uint32_t n, uint64_t* __restrict__ pay_off,
                                       uint32_t* __restrict__ word, uint32_t* __restrict__ info) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const ManyFrame m = fr[s];
    if (m.skip) return;
    const uint8_t* f = base + m.off;
    const uint32_t tail = (m.flags & 1u) ? 4u : 0u;
    uint64_t p = m.hdr_len;
    uint32_t k = 0u, st = 0u, sum = 0u;
    for (;;) {
        if (p + 4u > m.len) { st = 1u; break; }
        const uint32_t w = (uint32_t)f[p] | ((uint32_t)f[p + 1] << 8) | ((uint32_t)f[p + 2] << 16) | ((uint32_t)f[p + 3] << 24);
        p += 4u;
        if (w == 0u) break;                                   // EndMark
        const uint32_t ln = w & 0x7FFFFFFFu;
        if (ln > m.block_size) { st = 2u; break; }
        if (p + ln + tail > m.len) { st = 1u; break; }
        if (k >= m.slot_cap) { st = 3u; break; }
        pay_off[m.slot + k] = m.off + p; word[m.slot + k] = w; k += 1u;
        p += (uint64_t)ln + tail;
    }
    if (st == 0u && (m.flags & 2u)) {
        if (p + 4u > m.len) st = 1u;
        else { sum = (uint32_t)f[p] | ((uint32_t)f[p + 1] << 8) | ((uint32_t)f[p + 2] << 16) | ((uint32_t)f[p + 3] << 24); p += 4u; }
    }
    uint32_t* o = info + 8u * s;
    o[0] = k; o[1] = st; o[2] = (uint32_t)p; o[3] = (uint32_t)(p >> 32); o[4] = sum;
}

// block checksums of n payloads against the 4 bytes stored behind each (frame/decompress.rs:255-261,275-278): bad[i] = 1 on a mismatch
__global__ void frame_sums_check_kernel(const uint8_t* __restrict__ base, const uint64_t* __restrict__ pay_off, const uint32_t* __restrict__ pay_len,
                                        const uint32_t* __restrict__ sums, uint32_t n, uint32_t* __restrict__ bad) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n) return;
    const uint8_t* q = base + pay_off[b] + pay_len[b];
    const uint32_t w = (uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16) | ((uint32_t)q[3] << 24);
    bad[b] = w != sums[b] ? 1u : 0u;
}

hipError_t launch_frame_many_heads(const uint8_t* base, const uint64_t* off, const uint64_t* len, uint32_t n, uint8_t* heads, hipStream_t s) {
    if (n == 0u) return hipSuccess;
    hipLaunchKernelGGL(frame_many_heads_kernel, dim3((n * 32u + 255u) / 256u), dim3(256), 0, s, base, off, len, n, heads);
    return hipGetLastError();
}
hipError_t launch_frame_many_walk(const uint8_t* base, const ManyFrame* fr, uint32_t n, uint64_t* pay_off, uint32_t* word, uint32_t* info, hipStream_t s) {
    if (n == 0u) return hipSuccess;
    hipLaunchKernelGGL(frame_many_walk_kernel, dim3((n + 63u) / 64u), dim3(64), 0, s, base, fr, n, pay_off, word, info);
    return hipGetLastError();
}
hipError_t launch_frame_sums_check(const uint8_t* base, const uint64_t* pay_off, const uint32_t* pay_len, const uint32_t* sums, uint32_t n, uint32_t* bad,
                                   hipStream_t s) {
    if (n == 0u) return hipSuccess;
    hipLaunchKernelGGL(frame_sums_check_kernel, dim3((n + 255u) / 256u), dim3(256), 0, s, base, pay_off, pay_len, sums, n, bad);
    return hipGetLastError();
}

hipError_t launch_copy_batch(const uint8_t* src_base, const uint64_t* src_off, const uint32_t* len, uint8_t* dst_base, const uint64_t* dst_off,
                             uint32_t n, hipStream_t s) {
    if (n == 0u) return hipSuccess;
    hipLaunchKernelGGL(copy_batch_kernel, dim3(n), dim3(256), 0, s, src_base, src_off, len, dst_base, dst_off, n);
    return hipGetLastError();
}

}  // namespace lz4flex_dev
