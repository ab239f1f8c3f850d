// xxh32_kernel.hip -- batched XXH32 (LZ4 frame block checksums, reference src/frame/compress.rs:313-316,
// src/frame/decompress.rs:178-187; algorithm: public XXH32 spec, third-party twox-hash in the reference).
// One GROUP of 4 lanes per buffer: lane j owns accumulator v(j+1) and reads dword j of every 16-byte stripe,
// so a group's loads are one contiguous 16 bytes and a wavefront hashes 16 buffers at once.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lz4_device.h"

namespace lz4flex_dev {

namespace {
constexpr uint32_t P1 = 2654435761u, P2 = 2246822519u, P3 = 3266489917u, P4 = 668265263u, P5 = 374761393u;
__device__ __forceinline__ uint32_t rotl(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
__device__ __forceinline__ uint32_t xld32(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
}  // namespace

__global__ void __launch_bounds__(256) xxh32_batch_kernel(const uint8_t* base, const uint64_t* off, const uint32_t* len,
                                                         uint32_t n, uint32_t seed, uint32_t* out) {
    const uint32_t tid = blockIdx.x * 256u + threadIdx.x;
    const uint32_t b = tid >> 2, j = tid & 3u;
    if (b >= n) return;
    const uint8_t* p = base + off[b];
    const uint32_t L = len[b];
    uint32_t v = j == 0u ? seed + P1 + P2 : (j == 1u ? seed + P2 : (j == 2u ? seed : seed - P1));
    const uint32_t stripes = L >> 4;
    for (uint32_t s = 0; s < stripes; ++s) v = rotl(v + xld32(p + 16u * s + 4u * j) * P2, 13) * P1;
    // merge: lane 0 of the group gathers the four accumulators
    const int lane = (int)(threadIdx.x & 63u);
    const int g0 = lane & ~3;
    const uint32_t v1 = __shfl(v, g0), v2 = __shfl(v, g0 + 1), v3 = __shfl(v, g0 + 2), v4 = __shfl(v, g0 + 3);
    if (j != 0u) return;
    uint32_t h = L >= 16u ? rotl(v1, 1) + rotl(v2, 7) + rotl(v3, 12) + rotl(v4, 18) : seed + P5;
    h += L;
    uint32_t q = stripes << 4;
    for (; q + 4u <= L; q += 4u) h = rotl(h + xld32(p + q) * P3, 17) * P4;
    for (; q < L; ++q) h = rotl(h + (uint32_t)p[q] * P5, 11) * P1;
    h ^= h >> 15; h *= P2; h ^= h >> 13; h *= P3; h ^= h >> 16;
    out[b] = h;
}

hipError_t launch_xxh32_batch(const uint8_t* base, const uint64_t* off, const uint32_t* len, uint32_t n, uint32_t seed,
                              uint32_t* out, hipStream_t s) {
    if (n == 0u) return hipSuccess;
    const uint32_t grid = (n * 4u + 255u) / 256u;
    hipLaunchKernelGGL(xxh32_batch_kernel, dim3(grid), dim3(256), 0, s, base, off, len, n, seed, out);
    return hipGetLastError();
}

}  // namespace lz4flex_dev
