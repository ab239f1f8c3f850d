// LDS access cost on gfx950: aligned vs byte-unaligned reads of 4/8/16 bytes per lane, random and consecutive
// addresses (the access shapes of lz4_compress_wave.hip).  hipcc --offload-arch=gfx950 -O3 tools/ubench_lds.hip -o /tmp/ubench_lds
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
template <int BYTES, bool WRITE = false>
__global__ void k(const uint32_t* addr, uint64_t* out, int iters, int waves, int nact) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    for (int i = threadIdx.x; i < 65536 / 4; i += blockDim.x) ((uint32_t*)lds)[i] = i * 2654435761u;
    __syncthreads();
    uint32_t a[8];
    for (int j = 0; j < 8; j++) a[j] = addr[(j * 64 + (threadIdx.x & 63)) ];
    uint32_t acc = 0;
    __syncthreads();
    uint64_t t0 = __builtin_readcyclecounter();
    if ((int)(threadIdx.x & 63) < nact)
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            a[j] = (a[j] + 4112u) & 0xFFFFu;             // same alignment class, new bank every time; defeats hoisting
            asm volatile("" : "+v"(a[j]));
            if (WRITE) {     // (round 4: the replay decoder's ring writes)
                if (BYTES == 16) { u32x4 v = {acc, a[j], acc, a[j]}; __builtin_memcpy(lds + a[j], &v, 16); }
                if (BYTES == 8) { u32x2 v = {acc, a[j]}; __builtin_memcpy(lds + a[j], &v, 8); }
                if (BYTES == 4) { uint32_t v = acc + a[j]; __builtin_memcpy(lds + a[j], &v, 4); }
                if (BYTES == 2) { uint16_t v = (uint16_t)(acc + a[j]); __builtin_memcpy(lds + a[j], &v, 2); }
                if (BYTES == 1) { lds[a[j]] = (uint8_t)(acc + a[j]); }
                acc += a[j];
                continue;
            }
            if (BYTES == 16) { u32x4 v; __builtin_memcpy(&v, lds + a[j], 16); acc += v.x ^ v.y ^ v.z ^ v.w; }
            if (BYTES == 8) { u32x2 v; __builtin_memcpy(&v, lds + a[j], 8); acc += v.x ^ v.y; }
            if (BYTES == 4) { uint32_t v; __builtin_memcpy(&v, lds + a[j], 4); acc += v; }
            if (BYTES == 2) { uint16_t v; __builtin_memcpy(&v, lds + a[j], 2); acc += v; }
            if (BYTES == 1) { acc += lds[a[j]]; }
        }
    }
    uint64_t t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * waves + (threadIdx.x >> 6)] = t1 - t0;
    if (acc == 0x12345678) out[0] = acc;
}
int main() {
    uint32_t* d_addr; uint64_t* d_out; hipMalloc(&d_addr, 512 * 4); hipMalloc(&d_out, 8 * 64 * 8);
    const int iters = 200;
    struct { const char* name; int mode; } pats[] = {{"random aligned", 0}, {"random unaligned", 1}, {"random, dword aligned only", 5}};
    for (int wr : {0, 1}) for (int waves : {8, 4}) for (int nact : {64, 32, 16, 8, 4, 1}) for (auto& p : pats) for (int bytes : {1, 2, 4, 8, 16}) {
        if (p.mode == 5 && bytes < 8) continue;
        if (waves == 4 && (bytes != 16 || nact < 16)) continue;
        uint32_t h[512]; srand(7);
        for (int j = 0; j < 512; j++) {
            uint32_t r = rand() % 60000;
            if (p.mode == 0) r &= ~(bytes - 1u);
            if (p.mode == 1) r |= 1u;
            if (p.mode == 5) r = (r & ~15u) | 4u;
            if (p.mode == 2) r = 1000 * (j / 64) + (j % 64) + 1;
            if (p.mode == 3) r = 2048 * (j / 64) + (j % 64) * 16;
            if (p.mode == 4) r = ((j % 64) % 5 == 0) ? (r | 1u) : 4096u * (j / 64) + 3u;
            h[j] = r;
        }
        hipMemcpy(d_addr, h, sizeof h, hipMemcpyHostToDevice);
        void (*kern)(const uint32_t*, uint64_t*, int, int, int) = wr ? (bytes == 16 ? k<16, true> : bytes == 8 ? k<8, true> : bytes == 1 ? k<1, true> : bytes == 4 ? k<4, true> : k<2, true>)
                                                                     : (bytes == 16 ? k<16> : bytes == 8 ? k<8> : bytes == 1 ? k<1> : bytes == 4 ? k<4> : k<2>);
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 65536 + 64);
        hipLaunchKernelGGL(kern, dim3(1), dim3(64 * waves), 65536 + 64, 0, d_addr, d_out, iters, waves, nact);
        hipDeviceSynchronize();
        uint64_t o[8]; hipMemcpy(o, d_out, sizeof o, hipMemcpyDeviceToHost);
        double mx = 0; for (int w = 0; w < waves; w++) mx = o[w] > mx ? o[w] : mx;
        printf("%s waves/CU %d active lanes %2d  %-40s %2d B/lane: %.1f cycles per wave-instruction (%.1f per CU)\n", wr ? "WRITE" : "READ ", waves, nact, p.name, bytes, mx / (iters * 8.0), mx / (iters * 8.0) / waves);
    }
    return 0;
}
