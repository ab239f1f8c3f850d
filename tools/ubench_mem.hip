// ubench_mem.hip -- dependent-chain latency of the memory paths the LZ4 kernels use (diagnostic only).
// hipcc --offload-arch=gfx950 -O3 tools/ubench_mem.hip -o /tmp/ubench_mem && /tmp/ubench_mem
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// mode: 0 divergent (random per lane), 1 group-coalesced (8 lanes consecutive), 2 wave-coalesced
// W: bytes per lane load (4, 8, 16); mis: byte misalignment added to every address
template <int W>
__device__ __forceinline__ uint32_t ldw(const uint8_t* p) {
    if (W == 4) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
    if (W == 8) { uint64_t v; __builtin_memcpy(&v, p, 8); return (uint32_t)v ^ (uint32_t)(v >> 32); }
    uint4 v; __builtin_memcpy(&v, p, 16); return v.x ^ v.y ^ v.z ^ v.w;
}

template <int W>
__global__ void chase(const uint8_t* __restrict__ buf, uint64_t region, uint32_t mode, uint32_t mis, uint32_t iters,
                      uint64_t* cycles, uint32_t* sink) {
    const uint32_t wave = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    const uint32_t lane = threadIdx.x & 63;
    const uint8_t* base = buf + (uint64_t)wave * region;
    uint32_t r = wave * 2654435761u + lane * 40503u + 12345u;
    uint32_t acc = 0;
    const uint32_t mask = (uint32_t)region - 1u;
    // warm the region into whatever cache can hold it
    for (uint32_t i = lane * 16u; i < region; i += 64u * 16u) acc += *(const uint32_t*)(base + i);
    __syncthreads();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (uint32_t it = 0; it < iters; ++it) {
        uint32_t a;
        if (mode == 0) a = (r & mask & ~63u);
        else if (mode == 1) a = ((__shfl(r, lane & ~7u) & mask & ~63u) + (lane & 7u) * W) & mask;
        else a = ((__shfl(r, 0) & mask & ~1023u) + lane * W) & mask;
        a = (a + mis) & (mask & ~31u | 31u);
        if (a + 32u > region) a = 0;
        const uint32_t v = ldw<W>(base + a);
        r = r * 1664525u + 1013904223u + (v & 1u);   // next address depends on the loaded value
        acc += v;
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    if (lane == 0) cycles[wave] = t1 - t0;
    if (acc == 0x12345678u) sink[0] = acc;
}

__global__ void lds_chase(uint32_t iters, uint32_t mis, uint64_t* cycles, uint32_t* sink) {
    __shared__ uint8_t lds[16384 + 64];
    const uint32_t lane = threadIdx.x & 63;
    for (uint32_t i = threadIdx.x; i < 16384 / 4; i += blockDim.x) ((uint32_t*)lds)[i] = i * 2654435761u;
    __syncthreads();
    uint32_t r = lane * 40503u + 7u, acc = 0;
    const uint64_t t0 = __builtin_readcyclecounter();
    for (uint32_t it = 0; it < iters; ++it) {
        const uint32_t a = ((r >> 4) & 16380u & ~3u) + mis;
        uint32_t v;
        __builtin_memcpy(&v, lds + a, 4);
        r = r * 1664525u + 1013904223u + (v & 1u);
        acc += v;
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    if (lane == 0) cycles[blockIdx.x] = t1 - t0;
    if (acc == 0x12345678u) sink[0] = acc;
}

// independent (non-dependent) loads: issue throughput of divergent loads, K loads in flight per wave
template <int W>
__global__ void stream(const uint8_t* __restrict__ buf, uint64_t region, uint32_t mode, uint32_t mis, uint32_t iters,
                       uint64_t* cycles, uint32_t* sink) {
    const uint32_t wave = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    const uint32_t lane = threadIdx.x & 63;
    const uint8_t* base = buf + (uint64_t)wave * region;
    uint32_t r = wave * 2654435761u + lane * 40503u + 12345u;
    uint32_t acc = 0;
    const uint32_t mask = (uint32_t)region - 1u;
    for (uint32_t i = lane * 16u; i < region; i += 64u * 16u) acc += *(const uint32_t*)(base + i);
    __syncthreads();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (uint32_t it = 0; it < iters; ++it) {
        uint32_t a;
        if (mode == 0) a = (r & mask & ~63u);
        else if (mode == 1) a = ((__shfl(r, lane & ~7u) & mask & ~63u) + (lane & 7u) * W) & mask;
        else a = ((__shfl(r, 0) & mask & ~1023u) + lane * W) & mask;
        a = a + mis;
        if (a + 32u > region) a = 0;
        acc += ldw<W>(base + a);
        r = r * 1664525u + 1013904223u;
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    if (lane == 0) cycles[wave] = t1 - t0;
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int W>
static void run(const char* what, bool dep, const uint8_t* buf, uint64_t region, uint32_t mode, uint32_t mis, int blocks, int threads,
                uint64_t* dcyc, uint32_t* dsink) {
    const uint32_t iters = 2000;
    const int waves = blocks * threads / 64;
    for (int rep = 0; rep < 2; ++rep) {
        if (dep) hipLaunchKernelGGL(chase<W>, dim3(blocks), dim3(threads), 0, 0, buf, region, mode, mis, iters, dcyc, dsink);
        else hipLaunchKernelGGL(stream<W>, dim3(blocks), dim3(threads), 0, 0, buf, region, mode, mis, iters, dcyc, dsink);
    }
    CHECK(hipDeviceSynchronize());
    std::vector<uint64_t> c(waves);
    CHECK(hipMemcpy(c.data(), dcyc, waves * 8, hipMemcpyDeviceToHost));
    double s = 0; for (auto v : c) s += (double)v;
    printf("%-10s W=%2d region=%8llu mode=%u mis=%u blocks=%d thr=%d : %8.1f cycles/load\n", what, W, (unsigned long long)region, mode, mis,
           blocks, threads, s / waves / iters);
}

int main() {
    const uint64_t total = 1ull << 30;
    uint8_t* buf; uint64_t* dcyc; uint32_t* dsink;
    CHECK(hipMalloc(&buf, total + 4096));
    CHECK(hipMemset(buf, 1, total + 4096));
    CHECK(hipMalloc(&dcyc, 8 * 65536)); CHECK(hipMalloc(&dsink, 64));
    // LDS
    for (uint32_t mis = 0; mis < 2; ++mis) {
        hipLaunchKernelGGL(lds_chase, dim3(256), dim3(64), 0, 0, 2000u, mis, dcyc, dsink);
        CHECK(hipDeviceSynchronize());
        std::vector<uint64_t> c(256); CHECK(hipMemcpy(c.data(), dcyc, 256 * 8, hipMemcpyDeviceToHost));
        double s = 0; for (auto v : c) s += (double)v;
        printf("lds dependent chain mis=%u: %.1f cycles/load (includes ~6 VALU)\n", mis, s / 256 / 2000);
    }
    // dependent chains, one wave per CU x 2 (like the encoder: 512 waves)
    const uint64_t regions[] = {4096, 65536, 1 << 20};
    for (uint64_t region : regions) {
        for (uint32_t mode = 0; mode < 3; ++mode) {
            for (uint32_t mis = 0; mis < 2; ++mis) {
                run<4>("dep", true, buf, region, mode, mis, 512, 64, dcyc, dsink);
            }
        }
        run<8>("dep", true, buf, region, 0, 1, 512, 64, dcyc, dsink);
        run<16>("dep", true, buf, region, 0, 0, 512, 64, dcyc, dsink);
        run<16>("dep", true, buf, region, 0, 1, 512, 64, dcyc, dsink);
        run<8>("dep", true, buf, region, 1, 1, 512, 64, dcyc, dsink);
    }
    // throughput: independent loads, 512 waves and 4096 waves
    for (uint64_t region : regions) {
        for (uint32_t mode = 0; mode < 3; ++mode) {
            run<4>("indep", false, buf, region, mode, 1, 512, 64, dcyc, dsink);
        }
        run<16>("indep", false, buf, region, 0, 1, 512, 64, dcyc, dsink);
        run<16>("indep", false, buf, region, 0, 0, 512, 64, dcyc, dsink);
        run<8>("indep", false, buf, region, 1, 1, 512, 64, dcyc, dsink);
    }
    for (uint32_t mode = 0; mode < 3; ++mode) run<4>("indep", false, buf, 65536, mode, 1, 1024, 256, dcyc, dsink);
    run<16>("indep", false, buf, 65536, 0, 1, 1024, 256, dcyc, dsink);
    return 0;
}
