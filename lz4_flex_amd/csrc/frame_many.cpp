// frame_many.cpp -- N frames at once: lz4flex_frame_compress_many / lz4flex_frame_decompress_many (include/lz4flex_amd.h).
//
// What it replaces: N FrameEncoders / FrameDecoders (src/frame/compress.rs:261-371, src/frame/decompress.rs:189-342), one per
// stream, each running its own sequence of block codec calls -- the reference's answer to many streams is one thread per stream.
// A Linked frame is ONE dependency chain (every block's matches reach into the block before it, decompress.rs:195-222): a single
// stream gives a GPU nothing to do side by side, N streams give it N chains.  Here the blocks of ALL streams form one batch:
//   encode: one launch of the block encoder over every block of every stream (throughput encoder: a Linked block's history is input,
//           LZ4FLEX_BLOCK_HISTORY; reference-exact mode: lz4flex_compress_chains, N chains side by side), then one thread per stream
//           lays the frame out (header, BlockInfo words, EndMark, content checksum) and one workgroup per block moves its payload;
//   decode: the headers come to the host (32 bytes per frame), one thread per frame walks its BlockInfo words, the host turns the
//           tables into ONE batch -- Independent frames: plain blocks; Linked frames: a chained batch of N chains ordered level by
//           level (every stream's block k before any stream's block k + 1, lz4flex_decompress_ext::chain_prev), so that all chains
//           advance together and a block waits for its own predecessor only.
// A decoded size is not in the frame: every block but a frame's last is taken to fill the block size.  Whatever does not fit that
// picture -- a frame that does not parse, a flush() boundary inside a frame, an empty block, a block or content
// checksum that does not match, a decode error, a buffer too small -- is decoded AGAIN by lz4flex_frame_decompress (the streaming
// decoder over this library's kernels), which has the reference's order of checks and names the error: the fast path never reports
// an error of its own.  The same on the encode side for streams of 2 GiB and more (lz4flex_frame_compress).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/lz4flex_amd.h"
#include "lz4_device.h"

namespace {

using namespace lz4flex_dev;

constexpr uint32_t UNCOMPRESSED_BIT = 0x80000000u;
constexpr uint64_t WINDOW_SIZE = 65536, FAST_HISTORY = 32768;
constexpr uint32_t CHAIN_MAX = 65536u;                   // blocks per chained decode batch (LZ4FLEX_MEM_CHAINED)
// ("decompress_level_chains", default 1 024: from this many Linked streams in one call on their blocks are decoded a level per launch -- the groups loop)
constexpr uint64_t MAX_SLOTS = 1ull << 26;               // block-table entries per decompress_many call (12 bytes each on the host and on the device)
constexpr uint64_t STREAM_MAX = 0x7FFF0000ull - (8u << 20);   // longer streams take the one-shot path (table reposition near 2 GiB, frame/compress.rs:266-271)

size_t block_bytes(int code) {                           // BlockSize::get_size, frame/header.rs:68-77
    switch (code) {
        case 4: return 64u << 10;
        case 5: return 256u << 10;
        case 6: return 1u << 20;
        case 7: return 4u << 20;
        default: return 0;
    }
}
int block_size_from_buf_length(uint64_t n) { return n > 256u * 1024 ? 7 : (n > 64u * 1024 ? 5 : 4); }   // frame/header.rs:57-67

#define TRY_HIP(expr)                                                        \
    do {                                                                     \
        const hipError_t e_ = (expr);                                        \
        if (e_ != hipSuccess) return e_ == hipErrorOutOfMemory ? -LZ4FLEX_E_NOMEM : -LZ4FLEX_E_HIP; \
    } while (0)
#define TRY_RC(expr)                   \
    do {                               \
        const int rc_ = (expr);        \
        if (rc_) return rc_;           \
    } while (0)

// descriptor arrays: one host image, one device copy (scratch slot 0)
struct Desc {
    std::vector<uint8_t> h;
    uint8_t* d = nullptr;
    size_t take(size_t bytes) {
        const size_t at = (h.size() + 63) / 64 * 64;
        h.resize(at + bytes, 0);
        return at;
    }
    template <class T> T* host(size_t at) { return reinterpret_cast<T*>(h.data() + at); }
    template <class T> T* dev(size_t at) const { return reinterpret_cast<T*>(d + at); }
    int upload(lz4flex_ctx* c, hipStream_t s) {
        void* p = nullptr;
        TRY_RC(ctx_scratch(c, 0, h.size() + 64, &p));
        d = (uint8_t*)p;
        if (!h.empty()) TRY_HIP(hipMemcpyAsync(d, h.data(), h.size(), hipMemcpyHostToDevice, s));
        return 0;
    }
    template <class T> int fetch(size_t at, size_t count, hipStream_t s) {
        if (count) TRY_HIP(hipMemcpyAsync(h.data() + at, d + at, count * sizeof(T), hipMemcpyDeviceToHost, s));
        return 0;
    }
};

struct DeviceGuard {
    int prev = 0;
    explicit DeviceGuard(int dev) { (void)hipGetDevice(&prev); (void)hipSetDevice(dev); }
    ~DeviceGuard() { (void)hipSetDevice(prev); }
};

// ---- the streams the batch path leaves out: FrameEncoder / FrameDecoder over flat buffers, one stream at a time
int compress_one(const uint8_t* d_in, uint64_t len, const lz4flex_frame_info* info, uint8_t* d_out, uint64_t cap, uint64_t* out_len,
                 int32_t* status, hipStream_t s) {
    std::vector<uint8_t> in((size_t)len), out((size_t)std::min<uint64_t>(cap, lz4flex_frame_compress_bound((size_t)len, info)));
    if (len) TRY_HIP(hipMemcpyAsync(in.data(), d_in, (size_t)len, hipMemcpyDeviceToHost, s));
    TRY_HIP(hipStreamSynchronize(s));
    const int64_t rc = lz4flex_frame_compress(in.data(), in.size(), info, out.data(), out.size(), nullptr);
    *status = rc < 0 ? (int32_t)rc : 0;
    *out_len = rc < 0 ? 0 : (uint64_t)rc;
    if (rc > 0) { TRY_HIP(hipMemcpyAsync(d_out, out.data(), (size_t)rc, hipMemcpyHostToDevice, s)); TRY_HIP(hipStreamSynchronize(s)); }
    return 0;
}
int decompress_one(const uint8_t* d_in, uint64_t len, uint8_t* d_out, uint64_t cap, uint64_t* out_len, int32_t* status,
                   lz4flex_err_detail* detail, hipStream_t s) {
    // (an "unbounded" out_cap must not become a host allocation of that size: LZ4 expands at most 255 x)
    const uint64_t most = len > (1ull << 40) ? cap : std::min<uint64_t>(cap, 255ull * len + 64u);
    std::vector<uint8_t> in((size_t)len), out((size_t)most);
    if (len) TRY_HIP(hipMemcpyAsync(in.data(), d_in, (size_t)len, hipMemcpyDeviceToHost, s));
    TRY_HIP(hipStreamSynchronize(s));
    static uint8_t none = 0;
    const int64_t rc = lz4flex_frame_decompress(len ? in.data() : &none, in.size(), cap ? out.data() : &none, out.size(), nullptr, detail);
    *status = rc < 0 ? (int32_t)rc : 0;
    *out_len = rc < 0 ? 0 : (uint64_t)rc;
    if (rc > 0) { TRY_HIP(hipMemcpyAsync(d_out, out.data(), (size_t)rc, hipMemcpyHostToDevice, s)); TRY_HIP(hipStreamSynchronize(s)); }
    return 0;
}

// =============================================================================================================== encode
int compress_many_device(lz4flex_ctx* c, const uint8_t* in, const uint64_t* in_off, const uint64_t* in_len, uint32_t n,
                         const lz4flex_frame_info* info, uint8_t* out, const uint64_t* out_off, const uint64_t* out_cap, uint64_t* out_len,
                         int32_t* status, hipStream_t s) {
    const bool linked = info->block_mode == 1, exact = ctx_comp_mode(c) == 1;
    const bool bsum = info->block_checksums != 0, csum = info->content_checksum != 0;
    // ---- which streams go through the batch, and their blocks
    std::vector<uint32_t> fast;                        // stream indices
    std::vector<uint32_t> slow;
    std::vector<int> code(n, 0);
    uint64_t nb64 = 0;
    for (uint32_t i = 0; i < n; i++) {
        code[i] = info->block_size ? info->block_size : block_size_from_buf_length(in_len[i]);
        const size_t mbs = block_bytes(code[i]);
        if (mbs == 0) return -LZ4FLEX_E_INVALID_ARG;                                   // (Max8MB is legacy-decode only, frame/header.rs:287)
        if (in_len[i] > STREAM_MAX) { slow.push_back(i); continue; }
        fast.push_back(i);
        nb64 += (in_len[i] + mbs - 1) / mbs;
    }
    if (nb64 > 0x7FFFFFFFull) return -LZ4FLEX_E_INVALID_ARG;
    const uint32_t nb = (uint32_t)nb64, nf = (uint32_t)fast.size();
    Desc D;
    const size_t a_ms = D.take(sizeof(ManyStream) * (size_t)nf), a_soff = D.take(8ull * nf), a_slen = D.take(4ull * nf),
                 a_in_off = D.take(8ull * nb), a_comp_off = D.take(8ull * nb), a_in_len = D.take(4ull * nb), a_flags = D.take(4ull * nb),
                 a_comp_cap = D.take(4ull * nb), a_cb = (exact && linked) ? D.take(sizeof(lz4flex_chain_block) * (size_t)nb) : 0,
                 a_first = (exact && linked) ? D.take(4ull * nf) : 0, a_count = (exact && linked) ? D.take(4ull * nf) : 0;
    const size_t up_bytes = D.h.size();
    // (results: written by the device)
    const size_t a_comp_len = D.take(4ull * nb), a_comp_st = D.take(4ull * nb), a_dst_off = D.take(8ull * nb), a_pay_off = D.take(8ull * nb),
                 a_pay_len = D.take(4ull * nb), a_sums = D.take(4ull * nb), a_csum = D.take(4ull * nf), a_flen = D.take(8ull * nf),
                 a_verdict = D.take(4ull * nf);
    uint64_t comp_bytes = 0;
    bool big = false;
    {
        ManyStream* ms = D.host<ManyStream>(a_ms);
        uint64_t* soff = D.host<uint64_t>(a_soff);
        uint32_t* slen = D.host<uint32_t>(a_slen);
        uint64_t* b_in = D.host<uint64_t>(a_in_off);
        uint64_t* b_co = D.host<uint64_t>(a_comp_off);
        uint32_t* b_len = D.host<uint32_t>(a_in_len);
        uint32_t* b_fl = D.host<uint32_t>(a_flags);
        uint32_t* b_cap = D.host<uint32_t>(a_comp_cap);
        lz4flex_chain_block* cb = (exact && linked) ? D.host<lz4flex_chain_block>(a_cb) : nullptr;
        uint32_t b = 0;
        for (uint32_t f = 0; f < nf; f++) {
            const uint32_t i = fast[f];
            const size_t mbs = block_bytes(code[i]);
            big |= mbs > 65536;
            const uint32_t cnt = (uint32_t)((in_len[i] + mbs - 1) / mbs);
            ManyStream& m = ms[f];
            m.out_off = out_off[i]; m.out_cap = out_cap[i]; m.first = b; m.count = cnt;
            m.flags = (bsum ? 1u : 0u) | (csum ? 2u : 0u);
            lz4flex_frame_info fi = *info;
            fi.block_size = code[i];
            if (fi.has_content_size) fi.content_size = in_len[i];                      // every frame carries ITS stream's length
            const int64_t hl = lz4flex_frame_info_write(&fi, m.hdr, sizeof m.hdr);
            if (hl < 0) return (int)hl;
            m.hdr_len = (uint32_t)hl;
            soff[f] = in_off[i]; slen[f] = (uint32_t)in_len[i];
            if (cb) { D.host<uint32_t>(a_first)[f] = b; D.host<uint32_t>(a_count)[f] = cnt; }
            // the reference's window bookkeeping for a Linked frame (frame/compress.rs:324-356), in coordinates of the stream: the
            // block's prefix starts at vbase, its external dictionary (the 64 KiB in front of the prefix) at dict_stream
            uint64_t vbase = 0, v_src_start = 0, dict_stream = 0, sso = 0;
            uint32_t ext = 0;
            for (uint32_t k = 0; k < cnt; k++, b++) {
                const uint64_t at = (uint64_t)k * mbs;
                const uint32_t len = (uint32_t)std::min<uint64_t>(mbs, in_len[i] - at);
                b_in[b] = in_off[i] + at; b_len[b] = len;
                b_cap[b] = (uint32_t)((lz4flex_get_maximum_output_size(len) + 63) / 64 * 64);
                b_co[b] = comp_bytes; comp_bytes += b_cap[b];
                if (!exact) b_fl[b] = (linked && k > 0) ? LZ4FLEX_BLOCK_HISTORY(std::min<uint64_t>(at, FAST_HISTORY)) : 0u;
                else b_fl[b] = k == 0 ? LZ4FLEX_BLOCK_FRAME_FIRST : LZ4FLEX_BLOCK_FRAME_CONTINUATION;   // (Independent; frame/compress.rs:357-367)
                if (cb) {
                    const uint64_t v_src_end = v_src_start + len;
                    lz4flex_chain_block& q = cb[b];
                    q.in_off = in_off[i] + vbase; q.in_len = (uint32_t)v_src_end; q.in_pos = (uint32_t)v_src_start;
                    q.dict_off = ext ? in_off[i] + dict_stream : 0; q.dict_len = ext; q.so = (uint32_t)sso; q.repos = 0; q.flags = 0;
                    v_src_start += len;
                    if (v_src_start >= mbs + WINDOW_SIZE) {
                        dict_stream = vbase + v_src_end - WINDOW_SIZE; ext = (uint32_t)WINDOW_SIZE;
                        sso += v_src_end; vbase += v_src_end; v_src_start = 0;
                    } else if (v_src_start + ext > WINDOW_SIZE) {
                        const uint64_t delta = std::min<uint64_t>(ext, v_src_start + ext - WINDOW_SIZE);
                        dict_stream += delta; ext -= (uint32_t)delta;
                    }
                }
            }
        }
    }
    (void)up_bytes;
    TRY_RC(D.upload(c, s));
    void* comp = nullptr;
    TRY_RC(ctx_scratch(c, 1, (size_t)comp_bytes + 64, &comp));
    if (nb) {
        if (exact && linked)
            TRY_RC(lz4flex_compress_chains(c, in, D.dev<lz4flex_chain_block>(a_cb), nb, D.dev<uint32_t>(a_first), D.dev<uint32_t>(a_count), nf, comp,
                                           D.dev<uint64_t>(a_comp_off), D.dev<uint32_t>(a_comp_cap), D.dev<uint32_t>(a_comp_len), D.dev<int32_t>(a_comp_st),
                                           nullptr, LZ4FLEX_MEM_DEVICE, s));
        else
            TRY_RC(lz4flex_compress_batch(c, in, D.dev<uint64_t>(a_in_off), D.dev<uint32_t>(a_in_len), D.dev<uint32_t>(a_flags), nb, comp,
                                          D.dev<uint64_t>(a_comp_off), D.dev<uint32_t>(a_comp_cap), D.dev<uint32_t>(a_comp_len), D.dev<int32_t>(a_comp_st),
                                          LZ4FLEX_MEM_DEVICE | (big ? LZ4FLEX_MEM_BIG_BLOCKS : 0), s));
    }
    if (csum && nf) TRY_RC(lz4flex_xxh32_batch_device(in, D.dev<uint64_t>(a_soff), D.dev<uint32_t>(a_slen), nf, 0, D.dev<uint32_t>(a_csum), s));   // frame/compress.rs:319-321
    TRY_HIP(launch_frame_many_assemble(D.dev<ManyStream>(a_ms), nf, in, D.dev<uint64_t>(a_in_off), D.dev<uint32_t>(a_in_len), (const uint8_t*)comp,
                                       D.dev<uint64_t>(a_comp_off), D.dev<uint32_t>(a_comp_len), D.dev<int32_t>(a_comp_st), nb, bsum ? 1 : 0,
                                       D.dev<uint32_t>(a_csum), out, D.dev<uint64_t>(a_dst_off), D.dev<uint64_t>(a_pay_off), D.dev<uint32_t>(a_pay_len),
                                       D.dev<uint32_t>(a_sums), D.dev<uint64_t>(a_flen), D.dev<int32_t>(a_verdict), s));
    TRY_RC(D.fetch<uint64_t>(a_flen, nf, s));
    TRY_RC(D.fetch<int32_t>(a_verdict, nf, s));
    TRY_HIP(hipStreamSynchronize(s));
    for (uint32_t f = 0; f < nf; f++) {
        const uint32_t i = fast[f];
        const int32_t v = D.host<int32_t>(a_verdict)[f];
        status[i] = v == 0 ? 0 : (v == 1 ? -LZ4FLEX_FE_COMPRESSION : -LZ4FLEX_FE_OUTPUT_FULL);
        out_len[i] = v == 0 ? D.host<uint64_t>(a_flen)[f] : 0;
    }
    for (uint32_t i : slow) {
        lz4flex_frame_info fi = *info;
        if (fi.has_content_size) fi.content_size = in_len[i];
        TRY_RC(compress_one(in + in_off[i], in_len[i], &fi, out + out_off[i], out_cap[i], &out_len[i], &status[i], s));
    }
    return 0;
}

// =============================================================================================================== decode
struct Entry {            // one compressed block of the batch
    uint32_t stream, k;   // its frame; its index among the frame's blocks
    uint64_t in_off, out_off;
    uint32_t in_len, cap, pos;
};

int decompress_many_device(lz4flex_ctx* c, const uint8_t* in, const uint64_t* in_off, const uint64_t* in_len, uint32_t n, uint8_t* out,
                           const uint64_t* out_off, const uint64_t* out_cap, uint64_t* out_len, int32_t* status, lz4flex_err_detail* detail,
                           hipStream_t s) {
    std::vector<uint8_t> again(n, 0);                  // 1: the stream goes through lz4flex_frame_decompress
    std::vector<lz4flex_frame_info> fi(n);
    // ---- 1. headers
    std::vector<uint8_t> heads(32ull * n);
    {
        Desc H;
        const size_t a_off = H.take(8ull * n), a_len = H.take(8ull * n), a_heads = H.take(32ull * n);
        memcpy(H.host<uint64_t>(a_off), in_off, 8ull * n);
        memcpy(H.host<uint64_t>(a_len), in_len, 8ull * n);
        TRY_RC(H.upload(c, s));
        TRY_HIP(launch_frame_many_heads(in, H.dev<uint64_t>(a_off), H.dev<uint64_t>(a_len), n, H.dev<uint8_t>(a_heads), s));
        TRY_HIP(hipMemcpyAsync(heads.data(), H.dev<uint8_t>(a_heads), 32ull * n, hipMemcpyDeviceToHost, s));
        TRY_HIP(hipStreamSynchronize(s));
    }
    // ---- 2. the block tables
    Desc W;
    const size_t a_fr = W.take(sizeof(ManyFrame) * (size_t)n);
    uint64_t slots = 0;
    for (uint32_t i = 0; i < n; i++) {
        ManyFrame& m = W.host<ManyFrame>(a_fr)[i];
        m.off = in_off[i]; m.len = in_len[i]; m.skip = 1;
        const int64_t hl = lz4flex_frame_info_read(heads.data() + 32ull * i, (size_t)std::min<uint64_t>(32, in_len[i]), &fi[i], nullptr);
        const size_t bs = hl < 0 ? 0 : block_bytes(fi[i].block_size);
        // (a header that does not parse, a legacy frame, an output a chained batch cannot address: the streaming decoder's business)
        if (hl < 0 || fi[i].legacy_frame || bs == 0 || (fi[i].block_mode == 1 && out_cap[i] > 0xFFFFFFFFull - 2 * bs)) { again[i] = 1; continue; }
        const uint64_t cap_blocks = out_cap[i] / bs + 2;
        if (slots + cap_blocks > MAX_SLOTS) { again[i] = 1; continue; }             // (a table of that size is not worth building: out_cap far beyond the data)
        m.hdr_len = (uint32_t)hl; m.block_size = (uint32_t)bs;
        m.flags = (fi[i].block_checksums ? 1u : 0u) | (fi[i].content_checksum ? 2u : 0u);
        m.slot = (uint32_t)slots; m.slot_cap = (uint32_t)cap_blocks; m.skip = 0;
        slots += cap_blocks;
    }
    const size_t a_pay = W.take(8ull * slots), a_word = W.take(4ull * slots), a_info = W.take(32ull * n);
    TRY_RC(W.upload(c, s));
    TRY_HIP(launch_frame_many_walk(in, W.dev<ManyFrame>(a_fr), n, W.dev<uint64_t>(a_pay), W.dev<uint32_t>(a_word), W.dev<uint32_t>(a_info), s));
    TRY_RC(W.fetch<uint64_t>(a_pay, slots, s));
    TRY_RC(W.fetch<uint32_t>(a_word, slots, s));
    TRY_RC(W.fetch<uint32_t>(a_info, 8ull * n, s));
    TRY_HIP(hipStreamSynchronize(s));
    // ---- 3. the batch: compressed blocks (Independent: plain; Linked: chained, level by level), stored blocks (one copy launch)
    std::vector<Entry> plain, chained;
    std::vector<uint64_t> r_in, r_out;
    std::vector<uint32_t> r_len;
    std::vector<uint64_t> p_off;                       // every payload, for the block checksums
    std::vector<uint32_t> p_len, p_stream;
    std::vector<uint64_t> raw_bytes(n, 0);             // bytes of a stream's stored blocks
    std::vector<uint32_t> n_comp(n, 0);
    for (uint32_t i = 0; i < n; i++) {
        if (again[i]) continue;
        const ManyFrame& m = W.host<ManyFrame>(a_fr)[i];
        const uint32_t* nf = W.host<uint32_t>(a_info) + 8ull * i;
        if (nf[1] != 0) { again[i] = 1; continue; }                                  // does not parse (bytes behind the frame are not read: read_to_end ends with the frame)
        const uint32_t cnt = nf[0], bs = m.block_size;
        const size_t plain0 = plain.size(), chained0 = chained.size(), r0 = r_in.size(), p0 = p_off.size();
        bool ok = true;
        for (uint32_t k = 0; k < cnt && ok; k++) {
            const uint32_t w = W.host<uint32_t>(a_word)[m.slot + k], len = w & ~UNCOMPRESSED_BIT;
            const uint64_t po = W.host<uint64_t>(a_pay)[m.slot + k], pos = (uint64_t)k * bs;
            if (len == 0 || pos >= out_cap[i]) { ok = false; break; }
            const uint32_t cap = (uint32_t)std::min<uint64_t>(bs, out_cap[i] - pos);
            if (m.flags & 1u) { p_off.push_back(po); p_len.push_back(len); p_stream.push_back(i); }
            if (w & UNCOMPRESSED_BIT) {
                if (len > cap || (k + 1 != cnt && len != bs)) { ok = false; break; }
                r_in.push_back(po); r_out.push_back(out_off[i] + pos); r_len.push_back(len);
                raw_bytes[i] += len;
            } else {
                Entry e;
                e.stream = i; e.k = k; e.in_off = po; e.in_len = len; e.pos = (uint32_t)pos;
                if (fi[i].block_mode == 1) { e.out_off = out_off[i]; e.cap = (uint32_t)pos + cap; chained.push_back(e); }
                else { e.out_off = out_off[i] + pos; e.cap = cap; e.pos = 0; plain.push_back(e); }
                n_comp[i] += 1;
            }
        }
        if (ok && fi[i].block_mode == 1 && n_comp[i] > CHAIN_MAX) ok = false;
        if (!ok) {
            again[i] = 1; raw_bytes[i] = 0; n_comp[i] = 0;
            plain.resize(plain0); chained.resize(chained0); r_in.resize(r0); r_out.resize(r0); r_len.resize(r0);
            p_off.resize(p0); p_len.resize(p0); p_stream.resize(p0);
        }
    }
    // chained groups: whole streams, at most CHAIN_MAX blocks per call, each ordered by (block index, stream)
    struct Group { size_t lo, hi; };
    std::vector<Group> groups;
    {
        size_t lo = 0, at = 0;
        while (at < chained.size()) {
            size_t hi = at;
            const uint32_t st = chained[at].stream;
            while (hi < chained.size() && chained[hi].stream == st) hi++;
            if (hi - lo > CHAIN_MAX) { groups.push_back({lo, at}); lo = at; }
            at = hi;
        }
        if (chained.size() > lo) groups.push_back({lo, chained.size()});
        for (const Group& g : groups)
            std::stable_sort(chained.begin() + g.lo, chained.begin() + g.hi, [](const Entry& a, const Entry& b) { return a.k < b.k; });
    }
    const uint32_t np = (uint32_t)plain.size(), nc = (uint32_t)chained.size(), nr = (uint32_t)r_in.size(), nk = (uint32_t)p_off.size(), ne = np + nc;
    Desc B;
    const size_t b_in = B.take(8ull * ne), b_out = B.take(8ull * ne), b_len = B.take(4ull * ne), b_cap = B.take(4ull * ne), b_pos = B.take(4ull * ne),
                 b_prev = B.take(4ull * ne), b_rin = B.take(8ull * nr), b_rout = B.take(8ull * nr), b_rlen = B.take(4ull * nr),
                 b_poff = B.take(8ull * nk), b_plen = B.take(4ull * nk);
    const size_t b_olen = B.take(4ull * ne), b_st = B.take(4ull * ne), b_det = B.take(16ull * ne), b_psum = B.take(4ull * nk), b_bad = B.take(4ull * nk),
                 b_coff = B.take(8ull * n), b_clen = B.take(4ull * n), b_csum = B.take(4ull * n);
    {
        std::vector<uint32_t> last(n, 0xFFFFFFFFu);    // a stream's latest compressed block in the current group
        auto put = [&](uint32_t at, const Entry& e, uint32_t prev) {
            B.host<uint64_t>(b_in)[at] = e.in_off; B.host<uint64_t>(b_out)[at] = e.out_off; B.host<uint32_t>(b_len)[at] = e.in_len;
            B.host<uint32_t>(b_cap)[at] = e.cap; B.host<uint32_t>(b_pos)[at] = e.pos; B.host<uint32_t>(b_prev)[at] = prev;
        };
        for (uint32_t j = 0; j < np; j++) put(j, plain[j], 0xFFFFFFFFu);
        for (const Group& g : groups)
            for (size_t j = g.lo; j < g.hi; j++) {
                const Entry& e = chained[j];
                put(np + (uint32_t)j, e, last[e.stream]);
                last[e.stream] = (uint32_t)(j - g.lo);                                 // (indices count from the group's first block)
            }
        if (nr) { memcpy(B.host<uint64_t>(b_rin), r_in.data(), 8ull * nr); memcpy(B.host<uint64_t>(b_rout), r_out.data(), 8ull * nr); memcpy(B.host<uint32_t>(b_rlen), r_len.data(), 4ull * nr); }
        if (nk) { memcpy(B.host<uint64_t>(b_poff), p_off.data(), 8ull * nk); memcpy(B.host<uint32_t>(b_plen), p_len.data(), 4ull * nk); }
    }
    TRY_RC(B.upload(c, s));
    if (nk) {                                          // frame/decompress.rs:255-261,275-278: block checksums, before anything is decoded
        TRY_RC(lz4flex_xxh32_batch_device(in, B.dev<uint64_t>(b_poff), B.dev<uint32_t>(b_plen), nk, 0, B.dev<uint32_t>(b_psum), s));
        TRY_HIP(launch_frame_sums_check(in, B.dev<uint64_t>(b_poff), B.dev<uint32_t>(b_plen), B.dev<uint32_t>(b_psum), nk, B.dev<uint32_t>(b_bad), s));
    }
    if (nr) TRY_RC(lz4flex_copy_batch_device(in, B.dev<uint64_t>(b_rin), B.dev<uint32_t>(b_rlen), out, B.dev<uint64_t>(b_rout), nr, s));   // :262-271
    bool big = false;
    for (const Entry& e : plain) big |= e.in_len > 131072u;
    if (np)
        TRY_RC(lz4flex_decompress_batch(c, in, B.dev<uint64_t>(b_in), B.dev<uint32_t>(b_len), np, out, B.dev<uint64_t>(b_out), B.dev<uint32_t>(b_cap),
                                        B.dev<uint32_t>(b_olen), B.dev<int32_t>(b_st), B.dev<uint64_t>(b_det), LZ4FLEX_MEM_DEVICE | (big ? LZ4FLEX_MEM_BIG_BLOCKS : 0), s));
    for (const Group& g : groups) {
        const size_t o = np + g.lo;
        lz4flex_decompress_ext ext{};
        ext.out_pos = B.dev<uint32_t>(b_pos) + o;
        ext.chain_prev = B.dev<uint32_t>(b_prev) + o;
        {
            uint32_t chains = 0;                       // (level-by-level order: the blocks without a predecessor are the chains)
            for (size_t j = g.lo; j < g.hi; j++) chains += B.host<uint32_t>(b_prev)[np + j] == 0xFFFFFFFFu ? 1u : 0u;
            ext.n_chains = chains;
        }
        const int level_min = lz4flex_get_tuning(c, "decompress_level_chains");
        if (level_min > 0 && ext.n_chains >= (uint32_t)level_min) {
            // MANY short chains (round 6): a LEVEL per launch -- block k of every stream, a plain batch with prefixes (out_pos) whose bytes the
            // launches before it have written: the batch decoders by batch shape (the sequence decoder from 641 blocks on: 4 096 chains of
            // 4 blocks 6.7 -> 2.3 ms per GiB) instead of one workgroup per block polling its predecessor.  A level costs ~0.3 ms whatever it
            // holds, so this is for thousands of chains; hundreds of long ones (256 x 64 blocks) stay with the chained launch
            for (size_t j = g.lo; j < g.hi;) {
                size_t e = j;
                while (e < g.hi && chained[e].k == chained[j].k) e++;
                const size_t oj = np + j;
                lz4flex_decompress_ext lv{};
                lv.out_pos = B.dev<uint32_t>(b_pos) + oj;
                TRY_RC(lz4flex_decompress_batch_ex(c, in, B.dev<uint64_t>(b_in) + oj, B.dev<uint32_t>(b_len) + oj, (uint32_t)(e - j), out, B.dev<uint64_t>(b_out) + oj,
                                                   B.dev<uint32_t>(b_cap) + oj, B.dev<uint32_t>(b_olen) + oj, B.dev<int32_t>(b_st) + oj, B.dev<uint64_t>(b_det) + 2 * oj, &lv,
                                                   LZ4FLEX_MEM_DEVICE, s));
                j = e;
            }
            continue;
        }
        TRY_RC(lz4flex_decompress_batch_ex(c, in, B.dev<uint64_t>(b_in) + o, B.dev<uint32_t>(b_len) + o, (uint32_t)(g.hi - g.lo), out, B.dev<uint64_t>(b_out) + o,
                                           B.dev<uint32_t>(b_cap) + o, B.dev<uint32_t>(b_olen) + o, B.dev<int32_t>(b_st) + o, B.dev<uint64_t>(b_det) + 2 * o, &ext,
                                           LZ4FLEX_MEM_DEVICE | LZ4FLEX_MEM_CHAINED, s));
    }
    TRY_RC(B.fetch<uint32_t>(b_olen, ne, s));
    TRY_RC(B.fetch<int32_t>(b_st, ne, s));
    TRY_RC(B.fetch<uint32_t>(b_bad, nk, s));
    TRY_HIP(hipStreamSynchronize(s));
    // ---- 4. verdicts
    std::vector<uint64_t> produced(raw_bytes);
    for (uint32_t j = 0; j < nk; j++) if (B.host<uint32_t>(b_bad)[j]) again[p_stream[j]] = 1;
    auto judge = [&](const Entry& e, uint32_t at) {
        const uint32_t i = e.stream;
        const ManyFrame& m = W.host<ManyFrame>(a_fr)[i];
        const uint32_t cnt = W.host<uint32_t>(a_info)[8ull * i], got = B.host<uint32_t>(b_olen)[at];
        if (B.host<int32_t>(b_st)[at] != 0 || (e.k + 1 != cnt && got != m.block_size)) again[i] = 1;   // an error; a short block inside the frame (flush())
        produced[i] += got;
    };
    for (uint32_t j = 0; j < np; j++) judge(plain[j], j);
    for (uint32_t j = 0; j < nc; j++) judge(chained[j], np + j);
    // content size and checksum (frame/decompress.rs:205-229)
    std::vector<uint32_t> cs;                          // streams whose content checksum is being computed
    for (uint32_t i = 0; i < n; i++) {
        if (again[i]) continue;
        if (fi[i].has_content_size && fi[i].content_size != produced[i]) { again[i] = 1; continue; }
        if (fi[i].content_checksum) {
            if (produced[i] > 0xFFFFFFFFull) { again[i] = 1; continue; }
            B.host<uint64_t>(b_coff)[cs.size()] = out_off[i]; B.host<uint32_t>(b_clen)[cs.size()] = (uint32_t)produced[i];
            cs.push_back(i);
        }
    }
    if (!cs.empty()) {
        const uint32_t m = (uint32_t)cs.size();
        TRY_HIP(hipMemcpyAsync(B.dev<uint64_t>(b_coff), B.host<uint64_t>(b_coff), 8ull * m, hipMemcpyHostToDevice, s));
        TRY_HIP(hipMemcpyAsync(B.dev<uint32_t>(b_clen), B.host<uint32_t>(b_clen), 4ull * m, hipMemcpyHostToDevice, s));
        TRY_RC(lz4flex_xxh32_batch_device(out, B.dev<uint64_t>(b_coff), B.dev<uint32_t>(b_clen), m, 0, B.dev<uint32_t>(b_csum), s));
        TRY_RC(B.fetch<uint32_t>(b_csum, m, s));
        TRY_HIP(hipStreamSynchronize(s));
        for (uint32_t j = 0; j < m; j++)
            if (B.host<uint32_t>(b_csum)[j] != W.host<uint32_t>(a_info)[8ull * cs[j] + 4]) again[cs[j]] = 1;
    }
    for (uint32_t i = 0; i < n; i++) {
        if (detail) memset(&detail[i], 0, sizeof detail[i]);
        if (!again[i]) { status[i] = 0; out_len[i] = produced[i]; continue; }
        TRY_RC(decompress_one(in + in_off[i], in_len[i], out + out_off[i], out_cap[i], &out_len[i], &status[i], detail ? &detail[i] : nullptr, s));
    }
    return 0;
}

// ---- host buffers: staged through device scratch (slots 2 and 3), then the device path
uint64_t staged(const uint64_t* len, uint32_t n, std::vector<uint64_t>& off) {
    uint64_t at = 0;
    off.resize(n);
    for (uint32_t i = 0; i < n; i++) { off[i] = at; at += (len[i] + 255) / 256 * 256; }
    return at;
}

}  // namespace

extern "C" {

int lz4flex_frame_compress_many(lz4flex_ctx* ctx, const void* in_base, const uint64_t* in_off, const uint64_t* in_len, uint32_t n,
                                const lz4flex_frame_info* info, void* out_base, const uint64_t* out_off, const uint64_t* out_cap,
                                uint64_t* out_len, int32_t* status, int mem_kind, void* hip_stream) {
    if (n == 0) return 0;
    if (!in_off || !in_len || !out_off || !out_cap || !out_len || !status || !out_base) return -LZ4FLEX_E_INVALID_ARG;
    TRY_RC(ctx_resolve(&ctx));
    lz4flex_frame_info def{};
    if (!info) info = &def;                                                             // FrameInfo::default(), frame/header.rs:151-163
    if (info->legacy_frame) return -LZ4FLEX_E_INVALID_ARG;
    try {
    if (mem_kind != LZ4FLEX_MEM_HOST && mem_kind != LZ4FLEX_MEM_DEVICE) return -LZ4FLEX_E_INVALID_ARG;
    // (the DEVICE path too: the call's scratch is allocated on -- and cached with -- the context's device, whatever the thread's current one is)
    DeviceGuard guard(ctx_device(ctx));
    if (mem_kind == LZ4FLEX_MEM_DEVICE)
        return compress_many_device(ctx, (const uint8_t*)in_base, in_off, in_len, n, info, (uint8_t*)out_base, out_off, out_cap, out_len, status,
                                    (hipStream_t)hip_stream);
    hipStream_t s = ctx_stream(ctx);
    std::vector<uint64_t> s_in, s_out, cap(n);
    for (uint32_t i = 0; i < n; i++) cap[i] = std::min<uint64_t>(out_cap[i], lz4flex_frame_compress_bound((size_t)in_len[i], info));
    const uint64_t in_bytes = staged(in_len, n, s_in), out_bytes = staged(cap.data(), n, s_out);
    void *d_in = nullptr, *d_out = nullptr;
    TRY_RC(ctx_scratch(ctx, 2, (size_t)in_bytes + 64, &d_in));
    TRY_RC(ctx_scratch(ctx, 3, (size_t)out_bytes + 64, &d_out));
    for (uint32_t i = 0; i < n; i++)
        if (in_len[i]) TRY_HIP(hipMemcpyAsync((uint8_t*)d_in + s_in[i], (const uint8_t*)in_base + in_off[i], (size_t)in_len[i], hipMemcpyHostToDevice, s));
    TRY_RC(compress_many_device(ctx, (const uint8_t*)d_in, s_in.data(), in_len, n, info, (uint8_t*)d_out, s_out.data(), cap.data(), out_len, status, s));
    for (uint32_t i = 0; i < n; i++)
        if (status[i] == 0 && out_len[i]) TRY_HIP(hipMemcpyAsync((uint8_t*)out_base + out_off[i], (const uint8_t*)d_out + s_out[i], (size_t)out_len[i], hipMemcpyDeviceToHost, s));
    TRY_HIP(hipStreamSynchronize(s));
    return 0;
    } catch (const std::bad_alloc&) { return -LZ4FLEX_E_NOMEM; }      // (host tables and staging are std::vectors: nothing escapes the C ABI)
      catch (...) { return -LZ4FLEX_E_NOMEM; }                         // (std::length_error of an absurd size)
}

int lz4flex_frame_decompress_many(lz4flex_ctx* ctx, const void* in_base, const uint64_t* in_off, const uint64_t* in_len, uint32_t n,
                                  void* out_base, const uint64_t* out_off, const uint64_t* out_cap, uint64_t* out_len, int32_t* status,
                                  lz4flex_err_detail* detail, int mem_kind, void* hip_stream) {
    if (n == 0) return 0;
    if (!in_off || !in_len || !out_off || !out_cap || !out_len || !status || !in_base) return -LZ4FLEX_E_INVALID_ARG;
    TRY_RC(ctx_resolve(&ctx));
    try {
    if (mem_kind != LZ4FLEX_MEM_HOST && mem_kind != LZ4FLEX_MEM_DEVICE) return -LZ4FLEX_E_INVALID_ARG;
    DeviceGuard guard(ctx_device(ctx));
    if (mem_kind == LZ4FLEX_MEM_DEVICE)
        return decompress_many_device(ctx, (const uint8_t*)in_base, in_off, in_len, n, (uint8_t*)out_base, out_off, out_cap, out_len, status, detail,
                                      (hipStream_t)hip_stream);
    hipStream_t s = ctx_stream(ctx);
    std::vector<uint64_t> s_in, s_out;
    const uint64_t in_bytes = staged(in_len, n, s_in), out_bytes = staged(out_cap, n, s_out);
    void *d_in = nullptr, *d_out = nullptr;
    TRY_RC(ctx_scratch(ctx, 2, (size_t)in_bytes + 64, &d_in));
    TRY_RC(ctx_scratch(ctx, 3, (size_t)out_bytes + 64, &d_out));
    for (uint32_t i = 0; i < n; i++)
        if (in_len[i]) TRY_HIP(hipMemcpyAsync((uint8_t*)d_in + s_in[i], (const uint8_t*)in_base + in_off[i], (size_t)in_len[i], hipMemcpyHostToDevice, s));
    TRY_RC(decompress_many_device(ctx, (const uint8_t*)d_in, s_in.data(), in_len, n, (uint8_t*)d_out, s_out.data(), out_cap, out_len, status, detail, s));
    for (uint32_t i = 0; i < n; i++)
        if (status[i] == 0 && out_len[i]) TRY_HIP(hipMemcpyAsync((uint8_t*)out_base + out_off[i], (const uint8_t*)d_out + s_out[i], (size_t)out_len[i], hipMemcpyDeviceToHost, s));
    TRY_HIP(hipStreamSynchronize(s));
    return 0;
    } catch (const std::bad_alloc&) { return -LZ4FLEX_E_NOMEM; }
      catch (...) { return -LZ4FLEX_E_NOMEM; }
}

}  // extern "C"
