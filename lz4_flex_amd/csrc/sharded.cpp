// sharded.cpp -- the frame layer across the GPUs of one node, behind the C ABI (BASELINE.json configs[3]; SURVEY 8(b)
// "one-shot lz4flex_frame_compress ... for the multi-GPU path", 8(e)).  One process per GPU calls these entry points with its
// RCCL communicator; a Rust host binds them like the rest of include/lz4flex_amd.h.  lz4_flex_amd/sharded.py is the same
// algorithm over torch.distributed (what bench.py --config 4 and the gloo tests run).
//
// BlockMode::Independent frames shard naturally (reference src/frame/compress.rs:261-371: one codec call per block): every
// rank owns a contiguous range of blocks, compresses them with the batched kernels and assembles its SEGMENT
// ([BlockInfo | payload | (XXH32)]* , csrc/frame_kernels.hip) on its own device.  ONE exchange step reassembles the frame:
// ncclAllGather of the segment sizes, an exclusive prefix sum, grouped ncclSend / ncclRecv of the segments to the root
// (ncclGather needs equal counts), which puts the header in front and the EndMark behind.  Decoding runs the other way:
// the root walks the block headers on its device, broadcasts the table, sends every rank the contiguous byte range of its
// blocks, and every rank decodes straight into its slice.  Linked frames and content checksums do not shard (every block /
// the running XXH32 depends on everything before it): -LZ4FLEX_E_UNSUPPORTED.
//
// RCCL is bound at the first call with more than one rank (dlopen: a single-GPU user of this library needs no RCCL).
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/lz4flex_amd.h"

namespace {

constexpr uint32_t UNCOMPRESSED_BIT = 0x80000000u;
constexpr uint64_t WINDOW_SIZE = 65536;

// ---- the few RCCL entry points used (rccl.h: ncclResult_t f(...), 0 = ncclSuccess; ncclUint8 = 1, ncclUint64 = 5)
struct Rccl {
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    bool ok = false;
};
constexpr int NCCL_U8 = 1, NCCL_U64 = 5;

const Rccl& rccl() {
    static Rccl r = [] {
        Rccl x;
        // LZ4FLEX_RCCL_LIB: another library with the six ncclXxx entry points (a site's own build of RCCL; tests/sim/mock_rccl.cpp --
        // ranks as threads of one process on one device, which RCCL itself refuses)
        void* h = nullptr;
        if (const char* own = getenv("LZ4FLEX_RCCL_LIB")) h = dlopen(own, RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return x;
        x.AllGather = (decltype(x.AllGather))dlsym(h, "ncclAllGather");
        x.Broadcast = (decltype(x.Broadcast))dlsym(h, "ncclBroadcast");
        x.Send = (decltype(x.Send))dlsym(h, "ncclSend");
        x.Recv = (decltype(x.Recv))dlsym(h, "ncclRecv");
        x.GroupStart = (decltype(x.GroupStart))dlsym(h, "ncclGroupStart");
        x.GroupEnd = (decltype(x.GroupEnd))dlsym(h, "ncclGroupEnd");
        x.ok = x.AllGather && x.Broadcast && x.Send && x.Recv && x.GroupStart && x.GroupEnd;
        return x;
    }();
    return r;
}

// LZ4FLEX_FORCE_COLLECTIVES=1: a communicator of ONE rank still goes through every collective (size all-gather, broadcasts, the
// segment gather / range scatter as a grouped send + receive to itself) instead of taking the shortcuts a single rank allows.  This is
// how tests/test_gpu_sharded_native.py runs the REAL librccl on a one-GPU box: the only RCCL evidence obtainable without a multi-GPU node.
bool force_collectives() {
    const char* e = getenv("LZ4FLEX_FORCE_COLLECTIVES");
    return e && !strcmp(e, "1");
}

size_t block_bytes(int code) {
    switch (code) {
        case 4: return 64u << 10;
        case 5: return 256u << 10;
        case 6: return 1u << 20;
        case 7: return 4u << 20;
        default: return 0;
    }
}

struct DevBuf {                       // a device allocation that frees itself
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t n) { return hipMalloc(&p, n ? n : 1); }
    template <class T> T* as() const { return (T*)p; }
};

// (LZ4FLEX_TRACE=1: which call failed, on stderr -- these paths have no other way to say so)
static int failed(const char* what, int line, int code) {
    if (getenv("LZ4FLEX_TRACE")) fprintf(stderr, "lz4flex sharded.cpp:%d: %s failed (%d)\n", line, what, code);
    return -LZ4FLEX_E_HIP;
}
#define TRY_HIP(e) do { const hipError_t e_ = (e); if (e_ != hipSuccess) return failed("HIP call", __LINE__, (int)e_); } while (0)
#define TRY_NCCL(e) do { const int e_ = (e); if (e_ != 0) return failed("RCCL call", __LINE__, e_); } while (0)
#define TRY_RC(e) do { const int rc_ = (e); if (rc_) return rc_; } while (0)

// contiguous block ranges [lo, hi) per rank, sizes differ by at most one (lz4_flex_amd/sharded.py partition)
void partition(uint64_t n_blocks, int world, int r, uint64_t* lo, uint64_t* hi) {
    const uint64_t base = n_blocks / (uint64_t)world, extra = n_blocks % (uint64_t)world;
    *lo = base * (uint64_t)r + std::min<uint64_t>((uint64_t)r, extra);
    *hi = *lo + base + ((uint64_t)r < extra ? 1 : 0);
}

}  // namespace

extern "C" {

// Worst-case bytes of the segment a rank produces for local_len bytes (what the root must be able to receive per rank is
// bounded by the same figure): the blocks stored raw + 8 bytes of BlockInfo / checksum each.
uint64_t lz4flex_frame_segment_bound(uint64_t local_len, const lz4flex_frame_info* info) {
    const size_t bs = info ? block_bytes(info->block_size) : 0;
    if (!bs) return 0;
    const uint64_t n = (local_len + bs - 1) / bs;
    return local_len + 8 * n + 16;
}

int lz4flex_frame_compress_sharded(lz4flex_ctx* ctx, void* nccl_comm, int rank, int world, int root, const void* local,
                                   uint64_t local_len, uint64_t first_block, const lz4flex_frame_info* info, void* frame,
                                   uint64_t frame_cap, uint64_t* frame_len, void* hip_stream) {
    if (!info || world < 1 || rank < 0 || rank >= world || root < 0 || root >= world || (local_len && !local)) return -LZ4FLEX_E_INVALID_ARG;
    if (info->block_mode != 0 || info->content_checksum || info->has_content_size || info->legacy_frame) return -LZ4FLEX_E_UNSUPPORTED;
    const size_t bs = block_bytes(info->block_size);
    if (!bs) return -LZ4FLEX_E_INVALID_ARG;                        // an explicit block size (Auto is a property of a stream, not of a shard)
    const bool coll = world > 1 || (nccl_comm != nullptr && force_collectives());     // the exchange goes through RCCL
    const bool self = coll && world == 1;                                             // ... the root's own segment too (send + receive to itself)
    if (coll && (!nccl_comm || !rccl().ok)) return -LZ4FLEX_E_UNSUPPORTED;
    hipStream_t s = (hipStream_t)hip_stream;
    const uint32_t n = (uint32_t)((local_len + bs - 1) / bs);
    const uint64_t stride = (lz4flex_get_maximum_output_size(bs) + 63) / 64 * 64;
    // ---- this rank's blocks -> its segment, on the device.  No rank may leave before a collective its peers will enter: whatever
    // fails in this phase (an allocation, a launch, a block) is SAID in the size all-gather below (seg_bytes = ~0) and every rank
    // returns there, this one with its own code.  The one thing that has to work first is the buffer the exchange itself uses.
    DevBuf comp, desc, seg, sizes;
    uint64_t seg_bytes = 0;
    TRY_HIP(sizes.alloc(8ull * (size_t)world + 8));
    const int local_rc = [&]() -> int {
    TRY_HIP(seg.alloc(lz4flex_frame_segment_bound(local_len, info)));
    if (n) {
        TRY_HIP(comp.alloc(stride * n));
        // descriptor arrays: in_off, comp_off, seg_off (n + 1) [u64]; in_len, cap, comp_len, status, flags [u32]; scratch 16 n
        const size_t o_in_off = 0, o_comp_off = 8ull * n, o_seg_off = 16ull * n, o_in_len = 24ull * n + 8, o_cap = o_in_len + 4ull * n,
                     o_clen = o_cap + 4ull * n, o_st = o_clen + 4ull * n, o_fl = o_st + 4ull * n, o_scr = (o_fl + 4ull * n + 15) & ~15ull,
                     total = o_scr + 16ull * n;
        TRY_HIP(desc.alloc(total));
        std::vector<uint8_t> h(o_scr, 0);
        // per-block table mode of a FrameEncoder that has already written first_block blocks (frame/compress.rs:266-271,:357-367)
        uint64_t so = 0;
        const uint64_t limit = 0xFFFFFFFFull / 2;
        std::vector<uint32_t> fl(n);
        for (uint64_t k = 0; k < first_block + n; k++) {
            if (so + bs + WINDOW_SIZE >= limit) so = 0;
            if (k >= first_block) fl[k - first_block] = so == 0 ? LZ4FLEX_BLOCK_FRAME_FIRST : LZ4FLEX_BLOCK_FRAME_CONTINUATION;
            so += bs;
        }
        for (uint32_t i = 0; i < n; i++) {
            ((uint64_t*)(h.data() + o_in_off))[i] = (uint64_t)i * bs;
            ((uint64_t*)(h.data() + o_comp_off))[i] = (uint64_t)i * stride;
            ((uint32_t*)(h.data() + o_in_len))[i] = (uint32_t)std::min<uint64_t>(bs, local_len - (uint64_t)i * bs);
            ((uint32_t*)(h.data() + o_cap))[i] = (uint32_t)stride;
            ((uint32_t*)(h.data() + o_fl))[i] = fl[i];
        }
        uint8_t* d = desc.as<uint8_t>();
        TRY_HIP(hipMemcpyAsync(d, h.data(), o_scr, hipMemcpyHostToDevice, s));
        TRY_RC(lz4flex_compress_batch(ctx, local, (const uint64_t*)(d + o_in_off), (const uint32_t*)(d + o_in_len), (const uint32_t*)(d + o_fl), n,
                                      comp.p, (const uint64_t*)(d + o_comp_off), (const uint32_t*)(d + o_cap), (uint32_t*)(d + o_clen),
                                      (int32_t*)(d + o_st), LZ4FLEX_MEM_DEVICE | (bs > 65536 ? LZ4FLEX_MEM_BIG_BLOCKS : 0), s));
        TRY_RC(lz4flex_frame_assemble_device(local, (const uint64_t*)(d + o_in_off), (const uint32_t*)(d + o_in_len), comp.p,
                                             (const uint64_t*)(d + o_comp_off), (const uint32_t*)(d + o_clen), n, info->block_checksums, seg.p,
                                             (uint64_t*)(d + o_seg_off), d + o_scr, s));
        std::vector<int32_t> st(n);
        TRY_HIP(hipMemcpyAsync(st.data(), d + o_st, 4ull * n, hipMemcpyDeviceToHost, s));
        TRY_HIP(hipMemcpyAsync(&seg_bytes, d + o_seg_off + 8ull * n, 8, hipMemcpyDeviceToHost, s));
        TRY_HIP(hipStreamSynchronize(s));
        for (uint32_t i = 0; i < n; i++) if (st[i] != 0) return -LZ4FLEX_FE_COMPRESSION;
    }
    return 0;
    }();
    if (local_rc) seg_bytes = ~0ull - (uint64_t)(uint32_t)(-local_rc);    // the code travels with the verdict: every rank returns the same one
    // ---- 1) all-gather of the segment sizes, 2) exclusive prefix sum
    std::vector<uint64_t> all((size_t)world, 0);
    if (coll) {
        uint64_t* dsz = sizes.as<uint64_t>();
        TRY_HIP(hipMemcpyAsync(dsz + world, &seg_bytes, 8, hipMemcpyHostToDevice, s));
        TRY_NCCL(rccl().AllGather(dsz + world, dsz, 1, NCCL_U64, nccl_comm, s));
        TRY_HIP(hipMemcpyAsync(all.data(), dsz, 8ull * (size_t)world, hipMemcpyDeviceToHost, s));
        TRY_HIP(hipStreamSynchronize(s));
    } else {
        all[0] = seg_bytes;
    }
    for (int r = 0; r < world; r++)
        if (all[(size_t)r] > ~0ull - 0x10000ull) return -(int)(uint32_t)(~0ull - all[(size_t)r]);       // every rank sees the same sizes: all leave here, with the first failing rank's code
    uint8_t hdr[19];
    const int64_t hl = lz4flex_frame_info_write(info, hdr, sizeof hdr);
    if (hl < 0) return (int)hl;
    std::vector<uint64_t> off((size_t)world);
    uint64_t at = (uint64_t)hl;
    for (int r = 0; r < world; r++) { off[(size_t)r] = at; at += all[(size_t)r]; }
    const uint64_t total = at + 4;                                   // + EndMark (frame/compress.rs:222-224)
    if (frame_len) *frame_len = rank == root ? total : 0;
    // ---- the root's verdict on its buffer reaches every rank before anybody sends (a root that left alone would leave the
    // senders waiting)
    uint64_t go = (rank != root || (frame && frame_cap >= total)) ? 1 : 0;
    if (coll) {
        uint64_t* dsz = sizes.as<uint64_t>();
        TRY_HIP(hipMemcpyAsync(dsz, &go, 8, hipMemcpyHostToDevice, s));
        TRY_NCCL(rccl().Broadcast(dsz, dsz, 1, NCCL_U64, root, nccl_comm, s));
        TRY_HIP(hipMemcpyAsync(&go, dsz, 8, hipMemcpyDeviceToHost, s));
        TRY_HIP(hipStreamSynchronize(s));
    }
    if (!go) return -LZ4FLEX_FE_OUTPUT_FULL;
    // ---- 3) variable-size gather to the root
    if (rank == root) {
        uint8_t* f = (uint8_t*)frame;
        TRY_HIP(hipMemcpyAsync(f, hdr, (size_t)hl, hipMemcpyHostToDevice, s));
        TRY_HIP(hipMemsetAsync(f + total - 4, 0, 4, s));
        if (seg_bytes && !self) TRY_HIP(hipMemcpyAsync(f + off[(size_t)rank], seg.p, seg_bytes, hipMemcpyDeviceToDevice, s));
        if (coll) {
            TRY_NCCL(rccl().GroupStart());
            for (int r = 0; r < world; r++)
                if ((r != rank || self) && all[(size_t)r]) TRY_NCCL(rccl().Recv(f + off[(size_t)r], all[(size_t)r], NCCL_U8, r, nccl_comm, s));
            if (self && seg_bytes) TRY_NCCL(rccl().Send(seg.p, seg_bytes, NCCL_U8, root, nccl_comm, s));
            TRY_NCCL(rccl().GroupEnd());
        }
    } else if (seg_bytes) {
        TRY_NCCL(rccl().GroupStart());
        TRY_NCCL(rccl().Send(seg.p, seg_bytes, NCCL_U8, root, nccl_comm, s));
        TRY_NCCL(rccl().GroupEnd());
    }
    TRY_HIP(hipStreamSynchronize(s));                                // the temporaries are freed on return
    return 0;
}

int lz4flex_frame_decompress_sharded(lz4flex_ctx* ctx, void* nccl_comm, int rank, int world, int root, const void* frame,
                                     uint64_t frame_bytes, void* out, uint64_t out_cap, uint64_t* out_len, uint64_t* first_block,
                                     uint64_t* n_blocks, lz4flex_frame_info* info_out, lz4flex_err_detail* detail, void* hip_stream) {
    if (world < 1 || rank < 0 || rank >= world || root < 0 || root >= world) return -LZ4FLEX_E_INVALID_ARG;
    const bool coll = world > 1 || (nccl_comm != nullptr && force_collectives());     // (see lz4flex_frame_compress_sharded)
    const bool self = coll && world == 1;
    if (coll && (!nccl_comm || !rccl().ok)) return -LZ4FLEX_E_UNSUPPORTED;
    hipStream_t s = (hipStream_t)hip_stream;
    // ---- the root validates the header (host: 19 bytes) and walks the block headers on its device
    uint64_t meta[4] = {0, 0, 0, 0};                                 // blocks, BlockSize code, block checksums, error code
    DevBuf d_off, d_word, d_info, d_meta;
    TRY_HIP(d_meta.alloc(32));
    uint32_t hdr_len = 0;
    // (as in the encoder: a call that fails on the root before the broadcast below is said IN that broadcast -- meta[3] --, the
    // root does not leave its peers waiting)
    const int root_rc = rank != root ? 0 : [&]() -> int {
        if (!frame || frame_bytes < 7) meta[3] = LZ4FLEX_FE_IO;
        else {
            uint8_t h[19] = {0};
            TRY_HIP(hipMemcpyAsync(h, frame, (size_t)std::min<uint64_t>(19, frame_bytes), hipMemcpyDeviceToHost, s));
            TRY_HIP(hipStreamSynchronize(s));
            lz4flex_frame_info fi{};
            const int64_t hl = lz4flex_frame_info_read(h, (size_t)std::min<uint64_t>(19, frame_bytes), &fi, detail);
            if (hl < 0) meta[3] = (uint64_t)(-hl);
            else if (fi.legacy_frame || fi.block_mode != 0 || fi.content_checksum) meta[3] = LZ4FLEX_E_UNSUPPORTED;
            else {
                hdr_len = (uint32_t)hl;
                const uint32_t bs = (uint32_t)block_bytes(fi.block_size);
                uint32_t max_blocks = (uint32_t)std::min<uint64_t>(1u << 26, std::max<uint64_t>(1024, 4 * (frame_bytes / bs) + 16));
                for (;;) {
                    if (d_off.p) { (void)hipFree(d_off.p); d_off.p = nullptr; }
                    if (d_word.p) { (void)hipFree(d_word.p); d_word.p = nullptr; }
                    TRY_HIP(d_off.alloc(8ull * max_blocks));
                    TRY_HIP(d_word.alloc(4ull * max_blocks));
                    if (!d_info.p) TRY_HIP(d_info.alloc(16));
                    TRY_RC(lz4flex_frame_walk_device(frame, frame_bytes, hdr_len, fi.block_checksums, bs, max_blocks, d_off.as<uint64_t>(),
                                                     d_word.as<uint32_t>(), d_info.as<uint32_t>(), s));
                    uint32_t wi[4];
                    TRY_HIP(hipMemcpyAsync(wi, d_info.p, 16, hipMemcpyDeviceToHost, s));
                    TRY_HIP(hipStreamSynchronize(s));
                    if (wi[1] == 3 && max_blocks < (1u << 26)) { max_blocks *= 8; continue; }
                    if (wi[1] == 2) meta[3] = LZ4FLEX_FE_BLOCK_TOO_BIG;       // frame/decompress.rs:242-247
                    else if (wi[1] != 0) meta[3] = LZ4FLEX_FE_IO;             // truncated frame
                    meta[0] = wi[0];
                    break;
                }
                meta[1] = (uint64_t)fi.block_size;
                meta[2] = fi.block_checksums ? 1 : 0;
            }
        }
        return 0;
    }();
    if (root_rc && !meta[3]) meta[3] = (uint64_t)(-root_rc);
    if (coll) {
        TRY_HIP(hipMemcpyAsync(d_meta.p, meta, 32, hipMemcpyHostToDevice, s));
        TRY_NCCL(rccl().Broadcast(d_meta.p, d_meta.p, 4, NCCL_U64, root, nccl_comm, s));
        TRY_HIP(hipMemcpyAsync(meta, d_meta.p, 32, hipMemcpyDeviceToHost, s));
        TRY_HIP(hipStreamSynchronize(s));
    }
    if (meta[3]) return root_rc ? root_rc : -(int)meta[3];
    const uint64_t nb = meta[0];
    const size_t bs = block_bytes((int)meta[1]);
    const bool has_bc = meta[2] != 0;
    const uint32_t tail = has_bc ? 4u : 0u;
    if (info_out) { memset(info_out, 0, sizeof *info_out); info_out->block_size = (int)meta[1]; info_out->block_checksums = has_bc; }
    // ---- the block table (payload offset, length word) on every rank
    std::vector<uint64_t> h_off((size_t)nb);
    std::vector<uint32_t> h_word((size_t)nb);
    if (nb) {
        if (rank != root) { TRY_HIP(d_off.alloc(8ull * nb)); TRY_HIP(d_word.alloc(4ull * nb)); }
        if (coll) {
            TRY_NCCL(rccl().Broadcast(d_off.p, d_off.p, nb, NCCL_U64, root, nccl_comm, s));
            TRY_NCCL(rccl().Broadcast(d_word.p, d_word.p, 4 * nb, NCCL_U8, root, nccl_comm, s));
        }
        TRY_HIP(hipMemcpyAsync(h_off.data(), d_off.p, 8ull * nb, hipMemcpyDeviceToHost, s));
        TRY_HIP(hipMemcpyAsync(h_word.data(), d_word.p, 4ull * nb, hipMemcpyDeviceToHost, s));
        TRY_HIP(hipStreamSynchronize(s));
    }
    uint64_t lo, hi;
    partition(nb, world, rank, &lo, &hi);
    const uint32_t n = (uint32_t)(hi - lo);
    if (first_block) *first_block = lo;
    if (n_blocks) *n_blocks = n;
    if (out_len) *out_len = 0;
    auto range_of = [&](int r, uint64_t* a, uint64_t* b) {
        uint64_t l2, h2;
        partition(nb, world, r, &l2, &h2);
        if (l2 == h2) { *a = *b = 0; return; }
        *a = h_off[(size_t)l2];
        *b = h_off[(size_t)h2 - 1] + (h_word[(size_t)h2 - 1] & ~UNCOMPRESSED_BIT) + tail;
    };
    // ---- every rank's blocks are contiguous in the frame: one transfer per rank
    uint64_t a = 0, b = 0;
    range_of(rank, &a, &b);
    DevBuf recv;
    const uint8_t* local = nullptr;
    if (rank == root) {
        local = (const uint8_t*)frame + a;
        if (self) { TRY_HIP(recv.alloc(b - a)); if (b > a) local = recv.as<uint8_t>(); }    // (forced: the root's own range travels too)
        if (coll) {
            TRY_NCCL(rccl().GroupStart());
            for (int r = 0; r < world; r++) {
                uint64_t ra, rb;
                range_of(r, &ra, &rb);
                if ((r != rank || self) && rb > ra) TRY_NCCL(rccl().Send((const uint8_t*)frame + ra, rb - ra, NCCL_U8, r, nccl_comm, s));
            }
            if (self && b > a) TRY_NCCL(rccl().Recv(recv.p, b - a, NCCL_U8, root, nccl_comm, s));
            TRY_NCCL(rccl().GroupEnd());
        }
    } else {
        TRY_HIP(recv.alloc(b - a));
        if (b > a) {
            TRY_NCCL(rccl().GroupStart());
            TRY_NCCL(rccl().Recv(recv.p, b - a, NCCL_U8, root, nccl_comm, s));
            TRY_NCCL(rccl().GroupEnd());
        }
        local = recv.as<uint8_t>();
    }
    if (n == 0) { TRY_HIP(hipStreamSynchronize(s)); return 0; }
    if ((uint64_t)n * bs > out_cap + (bs - 1) || !out) return -LZ4FLEX_FE_OUTPUT_FULL;   // (the frame's last block may be short)
    // ---- decode straight into place: compressed blocks through the batched decoder, stored ones through one batched copy
    std::vector<uint64_t> c_in, c_out, r_in, r_out, p_off;
    std::vector<uint32_t> c_len, c_cap, r_len, p_len;
    for (uint32_t i = 0; i < n; i++) {
        const uint64_t po = h_off[(size_t)(lo + i)] - a;
        const uint32_t w = h_word[(size_t)(lo + i)], len = w & ~UNCOMPRESSED_BIT;
        const uint64_t dst = (uint64_t)i * bs;
        const uint32_t cap = (uint32_t)std::min<uint64_t>(bs, out_cap > dst ? out_cap - dst : 0);
        p_off.push_back(po); p_len.push_back(len);
        if (w & UNCOMPRESSED_BIT) {
            if (len > cap) return -LZ4FLEX_FE_OUTPUT_FULL;
            r_in.push_back(po); r_out.push_back(dst); r_len.push_back(len);
        } else {
            c_in.push_back(po); c_out.push_back(dst); c_len.push_back(len); c_cap.push_back(cap);
        }
    }
    const uint32_t nc = (uint32_t)c_in.size(), nr = (uint32_t)r_in.size();
    // device arrays: [c_in c_out | r_in r_out | p_off] u64, then [c_len c_cap c_olen c_st | r_len | p_len p_sum] u32
    const size_t o_cin = 0, o_cout = 8ull * nc, o_rin = 16ull * nc, o_rout = o_rin + 8ull * nr, o_poff = o_rout + 8ull * nr,
                 o_clen = o_poff + 8ull * n, o_ccap = o_clen + 4ull * nc, o_colen = o_ccap + 4ull * nc, o_cst = o_colen + 4ull * nc,
                 o_rlen = o_cst + 4ull * nc, o_plen = o_rlen + 4ull * nr, o_psum = o_plen + 4ull * n, tot = o_psum + 4ull * n;
    DevBuf dd;
    TRY_HIP(dd.alloc(tot));
    std::vector<uint8_t> h(tot, 0);
    auto put = [&](size_t at, const void* src, size_t bytes) { if (bytes) memcpy(h.data() + at, src, bytes); };
    put(o_cin, c_in.data(), 8ull * nc); put(o_cout, c_out.data(), 8ull * nc); put(o_rin, r_in.data(), 8ull * nr); put(o_rout, r_out.data(), 8ull * nr);
    put(o_poff, p_off.data(), 8ull * n); put(o_clen, c_len.data(), 4ull * nc); put(o_ccap, c_cap.data(), 4ull * nc);
    put(o_rlen, r_len.data(), 4ull * nr); put(o_plen, p_len.data(), 4ull * n);
    uint8_t* d = dd.as<uint8_t>();
    TRY_HIP(hipMemcpyAsync(d, h.data(), tot, hipMemcpyHostToDevice, s));
    if (has_bc) {                                                    // frame/decompress.rs:255-261,275-278: verify before decoding
        TRY_RC(lz4flex_xxh32_batch_device(local, (const uint64_t*)(d + o_poff), (const uint32_t*)(d + o_plen), n, 0, (uint32_t*)(d + o_psum), s));
        std::vector<uint32_t> sums(n), stored(n);
        TRY_HIP(hipMemcpyAsync(sums.data(), d + o_psum, 4ull * n, hipMemcpyDeviceToHost, s));
        for (uint32_t i = 0; i < n; i++) TRY_HIP(hipMemcpyAsync(&stored[i], local + p_off[i] + p_len[i], 4, hipMemcpyDeviceToHost, s));
        TRY_HIP(hipStreamSynchronize(s));
        for (uint32_t i = 0; i < n; i++) if (sums[i] != stored[i]) return -LZ4FLEX_FE_BLOCK_CHECKSUM;
    }
    if (nc) TRY_RC(lz4flex_decompress_batch(ctx, local, (const uint64_t*)(d + o_cin), (const uint32_t*)(d + o_clen), nc, out, (const uint64_t*)(d + o_cout),
                                            (const uint32_t*)(d + o_ccap), (uint32_t*)(d + o_colen), (int32_t*)(d + o_cst), nullptr,
                                            LZ4FLEX_MEM_DEVICE | (bs > 65536 ? LZ4FLEX_MEM_BIG_BLOCKS : 0), s));
    if (nr) TRY_RC(lz4flex_copy_batch_device(local, (const uint64_t*)(d + o_rin), (const uint32_t*)(d + o_rlen), out, (const uint64_t*)(d + o_rout), nr, s));
    std::vector<uint32_t> olen(nc);
    std::vector<int32_t> st(nc);
    if (nc) {
        TRY_HIP(hipMemcpyAsync(olen.data(), d + o_colen, 4ull * nc, hipMemcpyDeviceToHost, s));
        TRY_HIP(hipMemcpyAsync(st.data(), d + o_cst, 4ull * nc, hipMemcpyDeviceToHost, s));
    }
    TRY_HIP(hipStreamSynchronize(s));
    uint64_t produced = 0;
    for (uint32_t i = 0, ci = 0, ri = 0; i < n; i++) {
        uint32_t got;
        if (h_word[(size_t)(lo + i)] & UNCOMPRESSED_BIT) got = r_len[ri++];
        else {
            if (st[ci] != 0) { if (detail) detail->inner = st[ci]; return -LZ4FLEX_FE_DECOMPRESSION; }
            got = olen[ci++];
        }
        if (got != bs && i + 1 != n) return -LZ4FLEX_E_UNSUPPORTED;   // a short block inside a shard (flush() boundary): not a sharded frame's shape
        produced += got;
    }
    if (out_len) *out_len = produced;
    return 0;
}

}  // extern "C"
