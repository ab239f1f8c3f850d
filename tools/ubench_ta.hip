// ubench_ta.hip -- issue cost of batches of independent VMEM instructions (diagnostic only):
// K loads are issued back to back, then waited for together; reports cycles per batch for K = 1, 2, 4, 8.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int W>
__device__ __forceinline__ uint32_t ldw(const uint8_t* p) {
    if (W == 1) return *p;
    if (W == 4) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
    if (W == 8) { uint64_t v; __builtin_memcpy(&v, p, 8); return (uint32_t)v ^ (uint32_t)(v >> 32); }
    uint4 v; __builtin_memcpy(&v, p, 16); return v.x ^ v.y ^ v.z ^ v.w;
}

// pattern: 0 = every lane its own random 64 B slot; 1 = 8-lane groups: consecutive W bytes inside a random slot;
//          2 = only lane 0 of each 8-lane group active (random slot); 3 = whole wave consecutive
template <int W, int K, int pattern>
__global__ void batch(const uint8_t* __restrict__ buf, uint32_t region, uint32_t, uint32_t iters, uint64_t* cycles,
                      uint32_t* sink, uint32_t mis) {
    const uint32_t wave = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    const uint32_t lane = threadIdx.x & 63;
    const uint8_t* base = buf + (uint64_t)wave * region;
    uint32_t r = wave * 2654435761u + lane * 40503u + 12345u;
    uint32_t acc = 0;
    const uint32_t mask = region - 1u;
    for (uint32_t i = lane * 16u; i < region; i += 64u * 16u) acc += *(const uint32_t*)(base + i);
    __syncthreads();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (uint32_t it = 0; it < iters; ++it) {
        uint32_t v[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            uint32_t rr = r * (2u * k + 3u) + 977u * k;
            uint32_t a;
            if (pattern == 0) a = (rr & mask & ~63u) + mis;
            else if (pattern == 1) a = ((__shfl(rr, lane & ~7u) & mask & ~127u) + (lane & 7u) * W + mis);
            else if (pattern == 2) a = (rr & mask & ~63u) + mis;
            else a = ((__shfl(rr, 0) & mask & ~2047u) + lane * W + mis);
            a &= mask;
            v[k] = 0;
            if (pattern != 2 || (lane & 7u) == 0u) v[k] = ldw<W>(base + a);
        }
#pragma unroll
        for (int k = 0; k < K; ++k) acc += v[k];
        r = r * 1664525u + 1013904223u + (acc & 1u);   // next batch depends on this one
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    if (lane == 0) cycles[wave] = t1 - t0;
    if (acc == 0x12345678u) sink[0] = acc;
}

static uint32_t g_mis = 1;
template <int W, int K, int P>
static double run1(const uint8_t* buf, uint32_t region, uint32_t pattern, int blocks, uint64_t* dcyc, uint32_t* dsink) {
    const uint32_t iters = 1000;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((batch<W, K, P>), dim3(blocks), dim3(64), 0, 0, buf, region, pattern, iters, dcyc, dsink, g_mis);
    CHECK(hipDeviceSynchronize());
    std::vector<uint64_t> c(blocks);
    CHECK(hipMemcpy(c.data(), dcyc, blocks * 8, hipMemcpyDeviceToHost));
    double s = 0; for (auto v : c) s += (double)v;
    return s / blocks / iters;
}

template <int W, int P>
static void runw(const uint8_t* buf, uint32_t region, uint32_t pattern, int blocks, uint64_t* dcyc, uint32_t* dsink) {
    printf("mis=%u W=%2d region=%7u pattern=%u waves=%4d : K=1 %7.0f  K=2 %7.0f  K=4 %7.0f  K=8 %7.0f cycles/batch\n", g_mis, W, region, pattern, blocks,
           run1<W, 1, P>(buf, region, pattern, blocks, dcyc, dsink), run1<W, 2, P>(buf, region, pattern, blocks, dcyc, dsink),
           run1<W, 4, P>(buf, region, pattern, blocks, dcyc, dsink), run1<W, 8, P>(buf, region, pattern, blocks, dcyc, dsink));
}

int main() {
    uint8_t* buf; uint64_t* dcyc; uint32_t* dsink;
    CHECK(hipMalloc(&buf, (1ull << 30) + 4096));
    CHECK(hipMemset(buf, 1, (1ull << 30) + 4096));
    CHECK(hipMalloc(&dcyc, 8 * 65536)); CHECK(hipMalloc(&dsink, 64));
    for (uint32_t mis : {0u, 1u}) {
        g_mis = mis;
        const int waves = 512; const uint32_t region = 65536u;
        runw<4, 0>(buf, region, 0, waves, dcyc, dsink);
        runw<4, 1>(buf, region, 1, waves, dcyc, dsink);
        runw<4, 2>(buf, region, 2, waves, dcyc, dsink);
        runw<4, 3>(buf, region, 3, waves, dcyc, dsink);
        runw<1, 1>(buf, region, 1, waves, dcyc, dsink);
        runw<8, 1>(buf, region, 1, waves, dcyc, dsink);
        runw<8, 0>(buf, region, 0, waves, dcyc, dsink);
        runw<16, 1>(buf, region, 1, waves, dcyc, dsink);
        runw<16, 0>(buf, region, 0, waves, dcyc, dsink);
        runw<16, 3>(buf, region, 3, waves, dcyc, dsink);
    }
    return 0;
}
