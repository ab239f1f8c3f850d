// mock_rccl.cpp -- TEST INFRASTRUCTURE: the six RCCL entry points csrc/sharded.cpp uses, for ranks that are THREADS of one
// process on ONE device (RCCL refuses two ranks on a device, and this build had no multi-GPU node): every operation waits for
// the caller's stream, meets its peers at a host-side barrier and moves the bytes with hipMemcpy device-to-device.  It checks
// what a real communicator would also insist on: every rank takes part in every collective with the same count and root,
// every send meets a receive of the same size.  Loaded through LZ4FLEX_RCCL_LIB (tests/test_gpu_sharded_native.py).
#include <hip/hip_runtime.h>

#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <deque>
#include <map>
#include <mutex>
#include <vector>

namespace {

struct Msg { const void* p; size_t bytes; bool taken; };
struct World {
    int n = 0;
    std::mutex m;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t gen = 0;
    std::vector<const void*> ptr;
    std::vector<size_t> cnt;
    std::vector<int> roots;
    std::map<std::pair<int, int>, std::deque<Msg*>> box;      // (src, dst) -> messages in order
    int errors = 0;
};
struct Op { bool send; const void* sp; void* rp; size_t bytes; int peer; };
struct Comm { World* w; int rank; int group = 0; std::vector<Op> ops; };

size_t dsize(int dt) { return dt == 5 || dt == 4 ? 8 : (dt == 2 || dt == 3 ? 4 : 1); }

void barrier(World* w, std::unique_lock<std::mutex>& lk) {
    const uint64_t g = w->gen;
    if (++w->arrived == w->n) { w->arrived = 0; w->gen++; w->cv.notify_all(); }
    else w->cv.wait(lk, [&] { return w->gen != g; });
}

int run_p2p(Comm* c) {
    World* w = c->w;
    std::vector<Msg*> mine;
    {
        std::unique_lock<std::mutex> lk(w->m);
        for (const Op& o : c->ops)
            if (o.send) { Msg* m = new Msg{o.sp, o.bytes, false}; w->box[{c->rank, o.peer}].push_back(m); mine.push_back(m); }
        w->cv.notify_all();
    }
    for (const Op& o : c->ops) {
        if (o.send) continue;
        Msg* m = nullptr;
        {
            std::unique_lock<std::mutex> lk(w->m);
            auto& q = w->box[{o.peer, c->rank}];
            w->cv.wait(lk, [&] { return !q.empty(); });
            m = q.front();
            q.pop_front();
            if (m->bytes != o.bytes) { w->errors++; fprintf(stderr, "mock_rccl: rank %d receives %zu bytes from %d, which sends %zu\n", c->rank, o.bytes, o.peer, m->bytes); }
        }
        if (hipMemcpy(o.rp, m->p, m->bytes < o.bytes ? m->bytes : o.bytes, hipMemcpyDeviceToDevice) != hipSuccess) return 1;
        std::unique_lock<std::mutex> lk(w->m);
        m->taken = true;
        w->cv.notify_all();
    }
    std::unique_lock<std::mutex> lk(w->m);
    for (Msg* m : mine) { w->cv.wait(lk, [&] { return m->taken; }); delete m; }
    c->ops.clear();
    return 0;
}

}  // namespace

extern "C" {

void* mock_world_create(int n) { World* w = new World; w->n = n; w->ptr.resize(n); w->cnt.resize(n); w->roots.resize(n); return w; }
int mock_world_errors(void* w) { return ((World*)w)->errors; }
void* mock_comm_create(void* w, int rank) { Comm* c = new Comm; c->w = (World*)w; c->rank = rank; return c; }

// ncclGroupStart / End have no communicator argument: the operations queued between them are kept per thread (a rank is a thread)
thread_local int t_group = 0;
thread_local std::vector<Comm*> t_comms;
int ncclGroupStart() { t_group++; return 0; }
int ncclGroupEnd() {
    if (--t_group > 0) return 0;
    int rc = 0;
    for (Comm* c : t_comms) rc |= run_p2p(c);       // sends are posted, receives done, then the sends' completion awaited
    t_comms.clear();
    return rc;
}

int ncclAllGather(const void* send, void* recv, size_t count, int dt, void* comm, hipStream_t s) {
    Comm* c = (Comm*)comm; World* w = c->w;
    if (hipStreamSynchronize(s) != hipSuccess) return 1;
    const size_t bytes = count * dsize(dt);
    std::unique_lock<std::mutex> lk(w->m);
    w->ptr[c->rank] = send; w->cnt[c->rank] = bytes;
    barrier(w, lk);
    for (int r = 0; r < w->n; r++) {
        if (w->cnt[r] != bytes) { w->errors++; fprintf(stderr, "mock_rccl: all-gather counts differ\n"); }
        if (hipMemcpy((uint8_t*)recv + (size_t)r * bytes, w->ptr[r], bytes, hipMemcpyDeviceToDevice) != hipSuccess) return 1;
    }
    barrier(w, lk);
    return 0;
}

int ncclBroadcast(const void* send, void* recv, size_t count, int dt, int root, void* comm, hipStream_t s) {
    Comm* c = (Comm*)comm; World* w = c->w;
    if (hipStreamSynchronize(s) != hipSuccess) return 1;
    const size_t bytes = count * dsize(dt);
    std::unique_lock<std::mutex> lk(w->m);
    w->ptr[c->rank] = send; w->cnt[c->rank] = bytes; w->roots[c->rank] = root;
    barrier(w, lk);
    for (int r = 0; r < w->n; r++)
        if (w->cnt[r] != bytes || w->roots[r] != root) { w->errors++; fprintf(stderr, "mock_rccl: broadcast arguments differ between ranks\n"); }
    if (recv != w->ptr[root] && hipMemcpy(recv, w->ptr[root], bytes, hipMemcpyDeviceToDevice) != hipSuccess) return 1;
    barrier(w, lk);
    return 0;
}

static int queue_op(Comm* c, const Op& o) {
    c->ops.push_back(o);
    if (t_group > 0) {
        bool known = false;
        for (Comm* k : t_comms) known |= k == c;
        if (!known) t_comms.push_back(c);
        return 0;
    }
    return run_p2p(c);
}
int ncclSend(const void* p, size_t count, int dt, int peer, void* comm, hipStream_t s) {
    if (hipStreamSynchronize(s) != hipSuccess) return 1;
    return queue_op((Comm*)comm, Op{true, p, nullptr, count * dsize(dt), peer});
}
int ncclRecv(void* p, size_t count, int dt, int peer, void* comm, hipStream_t s) {
    if (hipStreamSynchronize(s) != hipSuccess) return 1;
    return queue_op((Comm*)comm, Op{false, nullptr, p, count * dsize(dt), peer});
}

}  // extern "C"
