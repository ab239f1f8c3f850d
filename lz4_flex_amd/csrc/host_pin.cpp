// host_pin.cpp -- the cache behind host_pin.h
#include "host_pin.h"

#include <hip/hip_runtime_api.h>

#include <cstdlib>
#include <mutex>
#include <vector>

namespace lz4flex {
namespace {

constexpr size_t MIN_CAP = 64u << 10;
constexpr size_t CACHE_BYTES = 512u << 20;      // released buffers kept for reuse; anything beyond is freed at once
struct Free { uint8_t* p; size_t cap; bool pinned; };
std::mutex g_mu;
std::vector<Free> g_free;
size_t g_cached = 0;

size_t round_cap(size_t n) {
    size_t c = MIN_CAP;
    while (c < n) c <<= 1;
    return c;
}

}  // namespace

uint8_t* pin_acquire(size_t min_cap, size_t* cap, bool* pinned) {
    const size_t want = round_cap(min_cap);
    {
        std::lock_guard<std::mutex> lk(g_mu);
        for (size_t i = 0; i < g_free.size(); i++) {
            if (g_free[i].cap == want) {
                const Free f = g_free[i];
                g_free[i] = g_free.back();
                g_free.pop_back();
                g_cached -= f.cap;
                *cap = f.cap; *pinned = f.pinned;
                return f.p;
            }
        }
    }
    void* p = nullptr;
    // (portable: every device of the process may transfer from it -- the frame layer runs on the default context's device,
    // and that is whatever device is current when the context is created)
    if (hipHostMalloc(&p, want, hipHostMallocPortable) == hipSuccess && p) {
        *cap = want; *pinned = true;
        return (uint8_t*)p;
    }
    (void)hipGetLastError();                     // no device / no pinned memory left: pageable memory does the same job, slower
    p = malloc(want);
    *cap = want; *pinned = false;
    return (uint8_t*)p;
}

void pin_release(uint8_t* p, size_t cap, bool pinned) {
    if (!p) return;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (g_cached + cap <= CACHE_BYTES) {
            g_free.push_back(Free{p, cap, pinned});
            g_cached += cap;
            return;
        }
    }
    if (pinned) (void)hipHostFree(p); else free(p);
}

}  // namespace lz4flex
