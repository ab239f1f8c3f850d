// frame.cpp -- host side of the LZ4 frame layer (lz4_flex::frame::{FrameEncoder, FrameDecoder,
// FrameInfo}; reference src/frame/{compress,decompress,header}.rs).  Header (de)serialisation,
// block framing, store-raw rule, checksums and the io::Write / io::Read behaviour live here;
// every block's bytes are produced by the batched HIP kernels through the C ABI.  Where the
// reference calls the block codec once per block (src/frame/compress.rs:282-298,
// src/frame/decompress.rs:288-305) this layer gathers up to `batch_blocks` blocks per launch.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/lz4flex_amd.h"
#include "host_pin.h"
#include "xxh32.h"

namespace {

using lz4flex::PinBuf;
using lz4flex::XxHash32;

// src/frame/header.rs:11-34
constexpr uint8_t FLG_RESERVED_MASK = 0x02, FLG_VERSION_MASK = 0xC0, FLG_SUPPORTED_VERSION_BITS = 0x40,
                  FLG_INDEPENDENT_BLOCKS = 0x20, FLG_BLOCK_CHECKSUMS = 0x10, FLG_CONTENT_SIZE = 0x08,
                  FLG_CONTENT_CHECKSUM = 0x04, FLG_DICTIONARY_ID = 0x01, BD_BLOCK_SIZE_MASK = 0x70,
                  BD_RESERVED_MASK = 0x8F;
constexpr uint32_t BLOCK_UNCOMPRESSED_SIZE_BIT = 0x80000000u, LZ4F_MAGIC_NUMBER = 0x184D2204u,
                   LZ4F_LEGACY_MAGIC_NUMBER = 0x184C2102u;
constexpr size_t MIN_FRAME_INFO_SIZE = 7, MAX_FRAME_INFO_SIZE = 19, WINDOW_SIZE = 64 * 1024;

uint32_t rd32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
uint64_t rd64(const uint8_t* p) { return (uint64_t)rd32(p) | ((uint64_t)rd32(p + 4) << 32); }
void wr32(uint8_t* p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }
void wr64(uint8_t* p, uint64_t v) { wr32(p, (uint32_t)v); wr32(p + 4, (uint32_t)(v >> 32)); }

// BlockSize::get_size, header.rs:68-77
size_t block_size_bytes(int code) {
    switch (code) {
        case 4: return 64u * 1024;
        case 5: return 256u * 1024;
        case 6: return 1024u * 1024;
        case 7: return 4u * 1024 * 1024;
        case 8: return 8u * 1024 * 1024;
        default: return 0;
    }
}
// BlockSize::from_buf_length, header.rs:57-67
int block_size_from_buf_length(size_t n) { return n > 256u * 1024 ? 7 : (n > 64u * 1024 ? 5 : 4); }

}  // namespace

extern "C" {

uint32_t lz4flex_xxh32(const uint8_t* data, size_t len, uint32_t seed) { return XxHash32::oneshot(seed, data, len); }

// FrameInfo::write, header.rs:232-275
int64_t lz4flex_frame_info_write(const lz4flex_frame_info* fi, uint8_t* out, size_t out_cap) {
    if (!fi || !out) return -LZ4FLEX_E_INVALID_ARG;
    const size_t write_size = MIN_FRAME_INFO_SIZE + (fi->has_content_size ? 8 : 0);
    if (out_cap < write_size) return -LZ4FLEX_FE_IO;
    uint8_t b[MAX_FRAME_INFO_SIZE] = {0};
    wr32(b, LZ4F_MAGIC_NUMBER);
    b[4] = FLG_SUPPORTED_VERSION_BITS;
    if (fi->block_checksums) b[4] |= FLG_BLOCK_CHECKSUMS;
    if (fi->content_checksum) b[4] |= FLG_CONTENT_CHECKSUM;
    if (fi->block_mode == 0) b[4] |= FLG_INDEPENDENT_BLOCKS;
    b[5] = (uint8_t)(fi->block_size << 4);
    size_t off = 6;
    if (fi->has_content_size) { b[4] |= FLG_CONTENT_SIZE; wr64(b + off, fi->content_size); off += 8; }
    b[off] = (uint8_t)(XxHash32::oneshot(0, b + 4, off - 4) >> 8);
    memcpy(out, b, write_size);
    return (int64_t)write_size;
}

// FrameInfo::read, header.rs:277-373
int64_t lz4flex_frame_info_read(const uint8_t* in, size_t in_len, lz4flex_frame_info* fi, lz4flex_err_detail* d) {
    if (!in || !fi) return -LZ4FLEX_E_INVALID_ARG;
    memset(fi, 0, sizeof *fi);
    if (in_len < 4) return -LZ4FLEX_FE_IO;
    const uint32_t magic = rd32(in);
    size_t p = 4;
    if (magic == LZ4F_LEGACY_MAGIC_NUMBER) { fi->block_size = 8; fi->legacy_frame = 1; return 4; }
    if (magic >= 0x184D2A50u && magic <= 0x184D2A5Fu) {
        if (in_len < 8) return -LZ4FLEX_FE_IO;
        if (d) d->expected = rd32(in + 4);
        return -LZ4FLEX_FE_SKIPPABLE_FRAME;
    }
    if (magic != LZ4F_MAGIC_NUMBER) return -LZ4FLEX_FE_WRONG_MAGIC;
    if (in_len < p + 2) return -LZ4FLEX_FE_IO;
    const uint8_t flg = in[p], bd = in[p + 1];
    p += 2;
    if ((flg & FLG_VERSION_MASK) != FLG_SUPPORTED_VERSION_BITS) {
        if (d) d->expected = flg & FLG_VERSION_MASK;
        return -LZ4FLEX_FE_UNSUPPORTED_VERSION;
    }
    if ((flg & FLG_RESERVED_MASK) || (bd & BD_RESERVED_MASK)) return -LZ4FLEX_FE_RESERVED_BITS;
    fi->block_mode = (flg & FLG_INDEPENDENT_BLOCKS) ? 0 : 1;
    fi->content_checksum = (flg & FLG_CONTENT_CHECKSUM) != 0;
    fi->block_checksums = (flg & FLG_BLOCK_CHECKSUMS) != 0;
    const int bs = (bd & BD_BLOCK_SIZE_MASK) >> 4;
    if (bs <= 3) { if (d) d->expected = (uint64_t)bs; return -LZ4FLEX_FE_UNSUPPORTED_BLOCKSIZE; }
    fi->block_size = bs;
    if (flg & FLG_CONTENT_SIZE) {
        if (in_len < p + 8) return -LZ4FLEX_FE_IO;
        fi->has_content_size = 1; fi->content_size = rd64(in + p); p += 8;
    }
    bool has_dict = false;
    if (flg & FLG_DICTIONARY_ID) { if (in_len < p + 4) return -LZ4FLEX_FE_IO; has_dict = true; p += 4; }
    if (in_len < p + 1) return -LZ4FLEX_FE_IO;
    if ((uint8_t)(XxHash32::oneshot(0, in + 4, p - 4) >> 8) != in[p]) return -LZ4FLEX_FE_HEADER_CHECKSUM;
    p += 1;
    if (has_dict) return -LZ4FLEX_FE_DICTIONARY_NOT_SUPPORTED;   // frame/decompress.rs:139-142
    return (int64_t)p;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------
// FrameEncoder
// Blocks gathered per kernel launch.  A block is one serial chain on the device (a 4 MiB block takes 64x as long as
// a 64 KiB one), so throughput needs MANY blocks in flight: by default at least 256 blocks per launch (one per
// CU), capped at 1 GiB of staged input; an explicit set_batch_bytes is honoured exactly.
static size_t blocks_per_launch(size_t batch_bytes, size_t mbs, bool automatic) {
    size_t nb = std::max<size_t>(1, batch_bytes / mbs);
    if (automatic) nb = std::max<size_t>(nb, std::min<size_t>(256, ((size_t)1 << 30) / mbs));
    return nb;
}

struct lz4flex_frame_encoder {
    lz4flex_frame_info fi{};
    lz4flex_write_fn w = nullptr;
    void* user = nullptr;
    PinBuf src;                    // staged uncompressed bytes (whole blocks + a partial tail); page-locked: host_pin.h
    size_t src_len = 0;
    PinBuf dst;                    // compressed blocks at a fixed stride
    std::vector<uint64_t> in_off, out_off;
    std::vector<uint32_t> in_len, out_cap, out_len, flags;
    std::vector<int32_t> status;
    size_t batch_blocks = 0;       // blocks per kernel launch
    size_t batch_bytes = 64u << 20;
    bool batch_auto = true;        // no explicit set_batch_bytes: big blocks get enough blocks per launch to fill the GPU
    uint64_t content_len = 0;
    uint64_t src_stream_offset = 0;   // mirrors the reference field that decides the table state (N3)
    XxHash32 content_hasher{0};
    bool is_frame_open = false, data_to_frame_written = false;
    bool final_write = false;       // lz4flex_frame_compress: the write in progress is the stream's last (finish follows)
    int sticky_err = 0;
    // Linked mode (frame/compress.rs:62-93): the reference's src ring (prefix + ext_dict) expressed in
    // stream coordinates; the dependent blocks of a batch run as ONE chain on the GPU.
    PinBuf lstage;                      // stream bytes [lbase, lbase + lstage_len)
    size_t lstage_len = 0;
    uint64_t lbase = 0;                 // stream position of lstage[0]
    uint64_t proc_pos = 0;              // stream position of the first byte not yet compressed
    uint64_t frame_start = 0;           // stream position of the current frame's first byte (history never reaches before it)
    uint64_t vbase = 0;                 // stream position of the reference's src[0]
    uint64_t v_src_start = 0;           // == src_start == src_end between blocks
    uint64_t dict_stream = 0;           // stream position of ext_dict[0]
    uint32_t ext_dict_len = 0;
    std::vector<uint32_t> tbl_state;    // the persistent HashTable4K
    std::vector<lz4flex_chain_block> cblocks;
    bool linked_fast = false;           // Linked frame written with the throughput encoder: blocks parsed independently

    int emit(const uint8_t* p, size_t n) {
        while (n) {
            const int64_t k = w(user, p, n);
            if (k <= 0) return -LZ4FLEX_FE_IO;     // io::ErrorKind::WriteZero / other
            p += (size_t)k; n -= (size_t)k;
        }
        return 0;
    }
    // begin_frame, frame/compress.rs:234-257 (+ init :96-118)
    int begin_frame(size_t buf_len) {
        is_frame_open = true;
        if (fi.block_size == 0) fi.block_size = block_size_from_buf_length(buf_len);
        const size_t mbs = block_size_bytes(fi.block_size);
        if (mbs == 0 || fi.block_size == 8) return -LZ4FLEX_E_INVALID_ARG;   // Max8MB is legacy-decode only (header.rs:287)
        batch_blocks = blocks_per_launch(batch_bytes, mbs, batch_auto);
        uint8_t hdr[MAX_FRAME_INFO_SIZE];
        const int64_t n = lz4flex_frame_info_write(&fi, hdr, sizeof hdr);
        if (n < 0) return (int)n;
        int rc = emit(hdr, (size_t)n);
        if (rc) return rc;
        if (content_len != 0) {   // second or later frame of this encoder: reset compressor state
            content_len = 0; src_stream_offset = 0; src_len = 0;
            content_hasher.reset(0);
            ext_dict_len = 0; v_src_start = 0; vbase = proc_pos;
            std::fill(tbl_state.begin(), tbl_state.end(), 0u);
        }
        if (fi.block_mode == 1 && tbl_state.empty()) tbl_state.assign(4096, 0u);
        linked_fast = fi.block_mode == 1 && lz4flex_get_tuning(nullptr, "compress_mode") == 0;   // decided once per frame
        frame_start = proc_pos;
        return 0;
    }
    // One block's bytes on the wire (frame/compress.rs:301-321): BlockInfo, payload (the source itself where compression did
    // not help), optional block checksum; content hash and length move on
    int emit_block(const uint8_t* s, size_t slen, const uint8_t* comp_bytes, uint32_t comp_len) {
        const uint8_t* block_data; size_t block_len; uint32_t info;
        if (comp_len < slen) { block_data = comp_bytes; block_len = comp_len; info = comp_len; }                      // :301-306
        else { block_data = s; block_len = slen; info = (uint32_t)slen | BLOCK_UNCOMPRESSED_SIZE_BIT; }
        uint8_t bi[4]; wr32(bi, info);
        int rc;
        if ((rc = emit(bi, 4))) return rc;
        if ((rc = emit(block_data, block_len))) return rc;
        if (fi.block_checksums) {                                                                                      // :313-316
            uint8_t c[4]; wr32(c, XxHash32::oneshot(0, block_data, block_len));
            if ((rc = emit(c, 4))) return rc;
        }
        if (fi.content_checksum) content_hasher.write(s, slen);                                                        // :319-321
        content_len += slen;
        return 0;
    }
    // Linked frame, throughput encoder (compress_mode fast): a block's matches reach into the 32 KiB of the stream in front of it
    // (LZ4FLEX_BLOCK_HISTORY: the history is INPUT, so the blocks of a launch are still one batch and not one dependency chain
    // -- 64 blocks of 64 KiB: one launch of a fraction of a millisecond instead of 64 x 8 ms of chain).  Any LZ4 frame decoder
    // returns the input; JSON, 64 KiB blocks: ratio 0.2216 (the reference's Linked frame 0.2226, independent blocks 0.2307).
    // compress_mode exact keeps the reference's bytes (write_blocks_linked below).
    static constexpr size_t FAST_HISTORY = 32768;
    int write_blocks_linked_fast(size_t total) {
        const size_t mbs = block_size_bytes(fi.block_size);
        const size_t nblk = (total + mbs - 1) / mbs;
        if (nblk == 0) return 0;
        const size_t stride = (lz4flex_get_maximum_output_size(mbs) + 63) / 64 * 64;
        if (dst.size() < stride * nblk) dst.resize(stride * nblk);
        in_off.resize(nblk); out_off.resize(nblk); in_len.resize(nblk); out_cap.resize(nblk); out_len.resize(nblk); status.resize(nblk);
        flags.resize(nblk);
        const size_t base = (size_t)(proc_pos - lbase);
        size_t left = total;
        for (size_t i = 0; i < nblk; i++) {
            const size_t len = std::min(mbs, left);
            in_off[i] = base + i * mbs; in_len[i] = (uint32_t)len;
            out_off[i] = i * stride; out_cap[i] = (uint32_t)stride;
            // what the stage holds in front of the block, as far as it belongs to THIS frame
            const uint64_t have = std::min<uint64_t>(in_off[i], proc_pos + i * mbs - frame_start);
            flags[i] = LZ4FLEX_BLOCK_HISTORY(std::min<uint64_t>(have, FAST_HISTORY));
            left -= len;
        }
        int rc = lz4flex_compress_batch(nullptr, lstage.data(), in_off.data(), in_len.data(), flags.data(), (uint32_t)nblk, dst.data(),
                                        out_off.data(), out_cap.data(), out_len.data(), status.data(), LZ4FLEX_MEM_HOST, nullptr);
        if (rc) return rc;
        for (size_t i = 0; i < nblk; i++) {
            if (status[i] != 0) return -LZ4FLEX_FE_COMPRESSION;
            if ((rc = emit_block(lstage.data() + in_off[i], in_len[i], dst.data() + out_off[i], out_len[i]))) return rc;
        }
        proc_pos += total;
        const uint64_t keep_from = std::max<uint64_t>(lbase, proc_pos >= FAST_HISTORY ? proc_pos - FAST_HISTORY : 0);   // the next block's history stays
        const size_t drop = (size_t)(keep_from - lbase);
        memmove(lstage.data(), lstage.data() + drop, lstage_len - drop);
        lstage_len -= drop; lbase = keep_from;
        return 0;
    }
    // Linked mode: compress the stream bytes [proc_pos, proc_pos + total) as blocks of <= block_size, in order,
    // with the reference's window bookkeeping (frame/compress.rs:261-371) done in stream coordinates.
    int write_blocks_linked(size_t total) {
        if (linked_fast) return write_blocks_linked_fast(total);
        const size_t mbs = block_size_bytes(fi.block_size);
        const size_t nblk = (total + mbs - 1) / mbs;
        if (nblk == 0) return 0;
        const size_t stride = (lz4flex_get_maximum_output_size(mbs) + 63) / 64 * 64;
        if (dst.size() < stride * nblk) dst.resize(stride * nblk);
        cblocks.resize(nblk); out_off.resize(nblk); out_cap.resize(nblk); out_len.resize(nblk); status.resize(nblk);
        std::vector<uint64_t> bstart(nblk);
        std::vector<uint32_t> blen(nblk);
        size_t left = total;
        uint64_t pos = proc_pos;
        for (size_t i = 0; i < nblk; i++) {
            const size_t len = std::min(mbs, left);
            // :266-271 reposition near 2 GiB
            uint32_t repos = 0;
            if (src_stream_offset + mbs + WINDOW_SIZE >= (uint64_t)(0xFFFFFFFFu / 2)) {
                repos = (uint32_t)(src_stream_offset - ext_dict_len);
                src_stream_offset = ext_dict_len;
            }
            const uint64_t v_src_end = v_src_start + len;
            lz4flex_chain_block& b = cblocks[i];
            b.in_off = vbase - lbase;
            b.in_len = (uint32_t)v_src_end;
            b.in_pos = (uint32_t)v_src_start;
            b.dict_off = ext_dict_len ? dict_stream - lbase : 0;
            b.dict_len = ext_dict_len;
            b.so = (uint32_t)src_stream_offset;
            b.repos = repos;
            b.flags = 0;
            out_off[i] = i * stride; out_cap[i] = (uint32_t)stride;
            bstart[i] = pos; blen[i] = (uint32_t)len;
            // :324-356 buffer / offset maintenance
            v_src_start += len;
            if (v_src_start >= mbs + WINDOW_SIZE) {
                dict_stream = vbase + v_src_end - WINDOW_SIZE;
                ext_dict_len = (uint32_t)WINDOW_SIZE;
                src_stream_offset += v_src_end;
                vbase += v_src_end;
                v_src_start = 0;
            } else if (v_src_start + ext_dict_len > WINDOW_SIZE) {
                const uint64_t delta = std::min<uint64_t>(ext_dict_len, v_src_start + ext_dict_len - WINDOW_SIZE);
                dict_stream += delta;
                ext_dict_len -= (uint32_t)delta;
            }
            pos += len; left -= len;
        }
        const uint32_t first = 0, count = (uint32_t)nblk;
        int rc = lz4flex_compress_chains(nullptr, lstage.data(), cblocks.data(), (uint32_t)nblk, &first, &count, 1, dst.data(),
                                         out_off.data(), out_cap.data(), out_len.data(), status.data(), tbl_state.data(),
                                         LZ4FLEX_MEM_HOST, nullptr);
        if (rc) return rc;
        for (size_t i = 0; i < nblk; i++) {
            if (status[i] != 0) return -LZ4FLEX_FE_COMPRESSION;
            if ((rc = emit_block(lstage.data() + (bstart[i] - lbase), blen[i], dst.data() + out_off[i], out_len[i]))) return rc;
        }
        proc_pos += total;
        // drop history the window can no longer reach
        const uint64_t keep_from = ext_dict_len ? std::min(vbase, dict_stream) : vbase;
        if (keep_from > lbase) {
            const size_t drop = (size_t)(keep_from - lbase);
            memmove(lstage.data(), lstage.data() + drop, lstage_len - drop);
            lstage_len -= drop; lbase = keep_from;
        }
        return 0;
    }
    int64_t write_linked(const uint8_t* buf, size_t len) {
        const size_t mbs = block_size_bytes(fi.block_size);
        // (exact mode: the blocks of a launch are ONE dependency chain on the device -- 64 of them keep a launch in the tens of
        // milliseconds; throughput encoder: independent blocks, a full batch per launch)
        const size_t per_launch = (linked_fast ? batch_blocks : std::min<size_t>(batch_blocks, 64)) * mbs;
        const size_t total = len;
        while (len) {
            const size_t pending = (size_t)(lbase + lstage_len - proc_pos);
            if (pending == per_launch) { int rc = write_blocks_linked(pending); if (rc) return sticky_err = rc; continue; }
            const size_t n = std::min(per_launch - pending, len);
            if (lstage.size() < lstage_len + n) lstage.resize(std::max(lstage.size() * 2, lstage_len + n));
            memcpy(lstage.data() + lstage_len, buf, n);
            lstage_len += n; buf += n; len -= n;
        }
        return (int64_t)total;
    }

    // write_block (frame/compress.rs:261-371) for `nblk` blocks in one launch; the last may be partial.  The blocks are the staged bytes, or
    // (round 6) `avail` bytes at `direct` in the CALLER's buffer: a write of a whole batch, and the one-shot call's remainder, are compressed
    // where they lie -- staging them cost a host copy of every byte (12 instead of 24 GiB/s through host buffers)
    int write_blocks(size_t nblk, const uint8_t* direct = nullptr, size_t avail = 0) {
        if (nblk == 0) return 0;
        const uint8_t* const base = direct ? direct : src.data();
        const size_t have = direct ? avail : src_len;
        const size_t mbs = block_size_bytes(fi.block_size);
        const size_t stride = (lz4flex_get_maximum_output_size(mbs) + 63) / 64 * 64;
        if (dst.size() < stride * nblk) dst.resize(stride * nblk);
        in_off.resize(nblk); out_off.resize(nblk); in_len.resize(nblk); out_cap.resize(nblk);
        out_len.resize(nblk); flags.resize(nblk); status.resize(nblk);
        uint64_t so = src_stream_offset;
        size_t consumed = 0;
        for (size_t i = 0; i < nblk; i++) {
            const size_t len = std::min(mbs, have - i * mbs);
            // reposition near 2 GiB (frame/compress.rs:266-271): the table collapses to "all zero, offset 0"
            if (so + mbs + WINDOW_SIZE >= (uint64_t)(0xFFFFFFFFu / 2)) so = 0;
            in_off[i] = i * mbs; in_len[i] = (uint32_t)len;
            out_off[i] = i * stride; out_cap[i] = (uint32_t)stride;
            flags[i] = so == 0 ? LZ4FLEX_BLOCK_FRAME_FIRST : LZ4FLEX_BLOCK_FRAME_CONTINUATION;
            so += len;
            consumed += len;
        }
        int rc = lz4flex_compress_batch(nullptr, base, in_off.data(), in_len.data(), flags.data(), (uint32_t)nblk,
                                        dst.data(), out_off.data(), out_cap.data(), out_len.data(), status.data(),
                                        LZ4FLEX_MEM_HOST, nullptr);
        if (rc) return rc;
        for (size_t i = 0; i < nblk; i++) {
            if (status[i] != 0) return -LZ4FLEX_FE_COMPRESSION;
            if ((rc = emit_block(base + in_off[i], in_len[i], dst.data() + out_off[i], out_len[i]))) return rc;
        }
        src_stream_offset = so;
        if (direct) return 0;
        // keep an unconsumed tail (never happens: callers pass every staged byte or whole blocks)
        if (consumed < src_len) memmove(src.data(), src.data() + consumed, src_len - consumed);
        src_len -= consumed;
        return 0;
    }
    // io::Write::write, frame/compress.rs:375-396
    int64_t write(const uint8_t* buf, size_t len) {
        if (sticky_err) return sticky_err;
        int rc;
        if (!is_frame_open && len != 0) { if ((rc = begin_frame(len))) return sticky_err = rc; }
        if (fi.block_mode == 1) return write_linked(buf, len);
        const size_t total = len;
        const size_t mbs = block_size_bytes(fi.block_size);
        while (len) {
            const size_t cap = batch_blocks * mbs;
            if (src_len == cap) {   // staging full: make space by writing the staged blocks
                if ((rc = write_blocks(batch_blocks))) return sticky_err = rc;
                continue;
            }
            if (src_len == 0 && len >= cap) {   // a whole batch lies in the caller's buffer: no staging
                if ((rc = write_blocks(batch_blocks, buf, cap))) return sticky_err = rc;
                buf += cap; len -= cap;
                continue;
            }
            if (final_write && src_len == 0) {  // the one-shot call's remainder: whatever is left is the frame's end, a partial last block included
                if ((rc = write_blocks((len + mbs - 1) / mbs, buf, len))) return sticky_err = rc;
                return (int64_t)total;
            }
            const size_t n = std::min(cap - src_len, len);
            if (src.size() < src_len + n) src.resize(std::min(cap, std::max(src.size() * 2, src_len + n)));   // grows with the data
            memcpy(src.data() + src_len, buf, n);
            src_len += n; buf += n; len -= n;
        }
        return (int64_t)total;
    }
    // io::Write::flush, frame/compress.rs:398-403
    int flush() {
        if (sticky_err) return sticky_err;
        if (fi.block_mode == 1 && is_frame_open) {
            const size_t pending = (size_t)(lbase + lstage_len - proc_pos);
            const int rc = pending ? write_blocks_linked(pending) : 0;
            return rc ? (sticky_err = rc) : 0;
        }
        if (src_len == 0) return 0;
        const size_t mbs = block_size_bytes(fi.block_size);
        const int rc = write_blocks((src_len + mbs - 1) / mbs);
        return rc ? (sticky_err = rc) : 0;
    }
    // try_finish, frame/compress.rs:173-187 (+ end_frame :209-230)
    int try_finish(lz4flex_err_detail* d) {
        int rc = flush();
        if (rc) return rc;
        if (!is_frame_open && !data_to_frame_written) { if ((rc = begin_frame(0))) return rc; }
        is_frame_open = false;
        if (fi.has_content_size && fi.content_size != content_len) {
            if (d) { d->expected = fi.content_size; d->actual = content_len; }
            return -LZ4FLEX_FE_CONTENT_LENGTH;
        }
        uint8_t z[4] = {0, 0, 0, 0};
        if ((rc = emit(z, 4))) return rc;
        if (fi.content_checksum) {
            uint8_t c[4]; wr32(c, content_hasher.finish());
            if ((rc = emit(c, 4))) return rc;
        }
        data_to_frame_written = true;
        return 0;
    }
};

// ------------------------------------------------------------------------------------------
// FrameDecoder
struct lz4flex_frame_decoder {
    lz4flex_read_fn r = nullptr;
    void* user = nullptr;
    bool have_frame = false;
    lz4flex_frame_info fi{};
    XxHash32 content_hasher{0};
    uint64_t content_len = 0;
    size_t batch_bytes = 64u << 20;
    bool batch_auto = true;
    // staged batch
    PinBuf comp, out;          // page-locked staging (host_pin.h)
    std::vector<uint64_t> in_off, out_off, detail;
    std::vector<uint32_t> in_len, out_cap, out_len;
    std::vector<int32_t> status;
    struct Item { size_t off, len; };     // decoded bytes waiting to be read: offsets into `out` (Independent) or `ldst` (Linked)
    std::vector<Item> ready;   // decoded pieces in stream order (len 0 = a block that decoded to nothing)
    size_t ready_idx = 0, ready_pos = 0;
    int pending_err = 0;       // surfaces after the pieces before it were delivered
    lz4flex_err_detail pending_detail{};
    bool pending_zero = false; // EndMark reached: one read() returns 0
    // Linked-mode window (frame/decompress.rs:62-72): exact mirror of the reference's dst ring
    PinBuf ldst;
    size_t ext_dict_offset = 0, ext_dict_len = 0, dst_start = 0;
    size_t lhave = 0;              // Linked: valid bytes in ldst (history + what the last call produced)

    bool io_failed = false;    // the read callback itself reported an error (not a short read)
    // read_exact; returns 0 ok, 1 clean EOF before any byte, -code on short read / error
    int read_exact(uint8_t* p, size_t n, size_t* got_out = nullptr) {
        size_t got = 0;
        while (got < n) {
            const int64_t k = r(user, p + got, n - got);
            if (k < 0) { io_failed = true; return -LZ4FLEX_FE_IO; }
            if (k == 0) break;
            got += (size_t)k;
        }
        if (got_out) *got_out = got;
        if (got == n) return 0;
        return got == 0 ? 1 : -LZ4FLEX_FE_IO;
    }
    // read_frame_info, frame/decompress.rs:109-168. 1 = EOF (Ok(0)), 0 = frame opened, <0 error
    int read_frame_info(lz4flex_err_detail* d) {
        uint8_t b[MAX_FRAME_INFO_SIZE];
        int rc = read_exact(b, 4);
        if (rc) return rc;
        size_t required;
        const uint32_t magic = rd32(b);
        if (magic == LZ4F_LEGACY_MAGIC_NUMBER) {
            required = 4;
        } else {
            rc = read_exact(b + 4, MIN_FRAME_INFO_SIZE - 4);
            if (rc) return rc;   // 1: r.read() returned 0 => Ok(0)
            // FrameInfo::read_size, header.rs:194-219
            if (magic >= 0x184D2A50u && magic <= 0x184D2A5Fu) required = 8;
            else if (magic != LZ4F_MAGIC_NUMBER) return -LZ4FLEX_FE_WRONG_MAGIC;
            else required = MIN_FRAME_INFO_SIZE + ((b[4] & FLG_CONTENT_SIZE) ? 8 : 0) + ((b[4] & FLG_DICTIONARY_ID) ? 4 : 0);
            if (required != MIN_FRAME_INFO_SIZE) {
                rc = read_exact(b + MIN_FRAME_INFO_SIZE, required - MIN_FRAME_INFO_SIZE);
                if (rc) return rc == 1 ? -LZ4FLEX_FE_IO : rc;
            }
        }
        const int64_t hs = lz4flex_frame_info_read(b, required, &fi, d);
        if (hs < 0) return (int)hs;
        have_frame = true;
        content_hasher.reset(0);
        content_len = 0;
        ext_dict_len = 0; ext_dict_offset = 0; dst_start = 0;
        lhave = 0;                                            // Linked: ldst = [history | this launch's blocks], grown on demand, never shrunk
        return 0;
    }
    bool ready_linked = false;     // the ready items index `ldst` (a Linked frame's blocks stay where they were decoded)
    const uint8_t* ready_base() const { return (ready_linked ? ldst : out).data(); }
    void fail(int code, const lz4flex_err_detail* d = nullptr) {
        pending_err = code;
        if (d) pending_detail = *d; else memset(&pending_detail, 0, sizeof pending_detail);
    }
    // EndMark handling, frame/decompress.rs:313-332
    void end_mark() {
        if (fi.has_content_size && content_len != fi.content_size) {
            lz4flex_err_detail d{}; d.expected = fi.content_size; d.actual = content_len;
            fail(-LZ4FLEX_FE_CONTENT_LENGTH, &d);
            return;
        }
        if (fi.content_checksum) {
            uint8_t c[4];
            if (read_exact(c, 4) != 0) { fail(-LZ4FLEX_FE_IO); return; }
            if (content_hasher.finish() != rd32(c)) { fail(-LZ4FLEX_FE_CONTENT_CHECKSUM); return; }
        }
        have_frame = false;
        pending_zero = true;
    }

    // Independent frames: gather blocks up to the batch size, decode them in one launch.
    // Round 6, `direct`: the reader asked for at least a block's worth of bytes (lz4flex_frame_decompress: for all of them) -- the blocks
    // are decoded STRAIGHT INTO ITS BUFFER, slot i at i * block size, and the number of bytes delivered is returned (0: nothing went that
    // way, the ready list says what there is).  Staging them and copying them out was a host copy of every decoded byte: 12 instead of
    // ~ 20 GiB/s through host buffers.  A slot that is not full (a flush() boundary, the frame's last block) is closed up by moving what
    // follows it; a block that decodes to nothing -- after which read() returns 0 once, frame/decompress.rs:344-349 -- and everything
    // behind it goes to the staging buffer as before.
    size_t read_blocks_independent(uint8_t* direct = nullptr, size_t direct_len = 0) {
        const size_t mbs = block_size_bytes(fi.block_size);
        size_t max_blocks = blocks_per_launch(batch_bytes, mbs, batch_auto);
        if (direct) max_blocks = std::min(max_blocks, direct_len / mbs);
        comp.clear(); in_off.clear(); in_len.clear(); out_off.clear(); out_cap.clear();
        ready.clear(); ready_idx = 0; ready_pos = 0;
        ready_linked = false;
        struct Slot { bool raw; size_t out_at; size_t idx; size_t raw_len; };
        std::vector<Slot> slots;
        size_t out_need = 0;
        bool saw_end = false;
        const size_t out_budget = std::max<size_t>(batch_bytes, mbs) * (batch_auto ? 4u : 1u);   // bytes of output reserved per launch
        while (slots.size() < max_blocks && (slots.empty() || out_need + mbs <= out_budget)) {
            uint8_t bi[4];
            const int rc = read_exact(bi, 4);
            if (rc != 0) {
                if (io_failed) fail(-LZ4FLEX_FE_IO);
                else pending_zero = true;   // UnexpectedEof on the block header => Ok(0), frame stays open (:231-238)
                break;
            }
            const uint32_t size = rd32(bi);
            if (size == 0) { saw_end = true; break; }   // EndMark: handled after the staged blocks are decoded
            const bool raw = (size & BLOCK_UNCOMPRESSED_SIZE_BIT) != 0;
            const size_t len = size & ~BLOCK_UNCOMPRESSED_SIZE_BIT;
            if (len > mbs) { fail(-LZ4FLEX_FE_BLOCK_TOO_BIG); break; }
            const size_t at = comp.size();
            comp.resize(at + len);
            if (len && read_exact(comp.data() + at, len) != 0) { comp.resize(at); fail(-LZ4FLEX_FE_IO); break; }
            if (fi.block_checksums) {
                uint8_t c[4];
                if (read_exact(c, 4) != 0) { comp.resize(at); fail(-LZ4FLEX_FE_IO); break; }
                if (XxHash32::oneshot(0, comp.data() + at, len) != rd32(c)) { comp.resize(at); fail(-LZ4FLEX_FE_BLOCK_CHECKSUM); break; }
            }
            Slot s{raw, out_need, in_off.size(), len};
            // An LZ4 block expands by less than 255x (one length byte adds at most 255 output bytes), so min(block size,
            // 255 len + 64) is as good a sink as the reference's block-size buffer and a frame of tiny blocks cannot make the
            // read-ahead reserve (and the device arena mirror) gigabytes.
            const size_t cap_i = direct ? mbs : std::min<size_t>(mbs, 255u * len + 64u);
            if (!raw) { in_off.push_back(at); in_len.push_back((uint32_t)len); out_off.push_back(out_need); out_cap.push_back((uint32_t)cap_i); out_need += cap_i; }
            else { s.idx = at; out_need += direct ? mbs : len; }
            slots.push_back(s);
        }
        if (!direct && out.size() < out_need) out.resize(out_need);
        uint8_t* const sink = direct ? direct : out.data();
        const size_t nb = in_off.size();
        out_len.assign(nb, 0); status.assign(nb, 0); detail.assign(2 * nb, 0);
        if (nb) {
            const int rc = lz4flex_decompress_batch(nullptr, comp.data(), in_off.data(), in_len.data(), (uint32_t)nb, sink,
                                                    out_off.data(), out_cap.data(), out_len.data(), status.data(),
                                                    detail.data(), LZ4FLEX_MEM_HOST, nullptr);
            if (rc) { fail(rc); pending_zero = false; return 0; }
        }
        size_t delivered = 0;          // direct: bytes closed up at the front of the reader's buffer
        bool staged = !direct;         // direct: a block decoded to nothing -- from there on the pieces wait in `out`
        size_t staged_at = 0;
        for (const Slot& s : slots) {
            size_t plen;
            if (s.raw) { memcpy(sink + s.out_at, comp.data() + s.idx, s.raw_len); plen = s.raw_len; }
            else {
                if (status[s.idx] != 0) {
                    lz4flex_err_detail d{}; d.expected = detail[2 * s.idx]; d.actual = detail[2 * s.idx + 1]; d.inner = status[s.idx];
                    fail(-LZ4FLEX_FE_DECOMPRESSION, &d);
                    pending_zero = false;
                    return delivered;   // pieces before this one were delivered / queued; the error surfaces after them
                }
                plen = out_len[s.idx];
            }
            content_len += plen;
            if (fi.content_checksum) content_hasher.write(sink + s.out_at, plen);
            if (direct && !staged && plen == 0) {
                staged = true;
                if (out.size() < out_need) out.resize(out_need);
            }
            if (!direct) ready.push_back({s.out_at, plen});
            else if (!staged) {
                if (s.out_at != delivered && plen) memmove(direct + delivered, direct + s.out_at, plen);
                delivered += plen;
            } else {
                if (plen) memcpy(out.data() + staged_at, direct + s.out_at, plen);
                ready.push_back({staged_at, plen});
                staged_at += plen;
            }
        }
        if (pending_err) { pending_zero = false; return delivered; }
        if (saw_end && !pending_zero) end_mark();
        return delivered;
    }

    // Linked frames (frame/decompress.rs:195-222,280-306: the reference decodes block after block into one window, a block's
    // matches may reach up to 64 KiB behind its start).  Here a run of compressed blocks is decoded by ONE launch
    // (LZ4FLEX_MEM_CHAINED): the staging buffer `ldst` holds [the frame's last <= 64 KiB | block | block | ...], block k is
    // given the position behind its predecessors as its initial sink position and the kernel makes it wait for them only where
    // a match really reaches behind its start -- the blocks of a frame written by this library's throughput encoder never do.
    // A block's decoded size is not in the frame: every block but a frame's last is taken to fill the block size, and the
    // blocks behind one that did not (a flush() boundary, frame/compress.rs:398-403) are decoded again from their real position.
    // A block stored raw ends a run (its bytes are placed by the host).
    void read_block_linked() {
        const size_t mbs = block_size_bytes(fi.block_size);
        const size_t max_blocks = blocks_per_launch(batch_bytes, mbs, batch_auto);
        ready.clear(); ready_idx = 0; ready_pos = 0;
        comp.clear(); in_off.clear(); in_len.clear();
        ready_linked = true;
        // ldst = carry (the frame's last <= 64 KiB) followed by this call's output.  The previous call's bytes were handed out
        // straight from ldst (no copy), so they are moved to the front only now that they have been consumed.
        if (lhave > WINDOW_SIZE) {
            memmove(ldst.data(), ldst.data() + (lhave - WINDOW_SIZE), WINDOW_SIZE);
            lhave = WINDOW_SIZE;
        }
        dst_start = lhave;
        const size_t carry = dst_start;                      // bytes of history at the front of ldst
        bool saw_end = false, raw_tail = false;
        size_t raw_at = 0, raw_len = 0;
        const size_t out_budget = std::max<size_t>(batch_bytes, mbs) * (batch_auto ? 4u : 1u);
        // (sink positions are 32-bit: an explicit batch size of 4 GiB or more must not wrap them)
        const size_t pos_blocks = (size_t)((0xFFFFFFFFull - carry) / mbs) - 1u;
        while (in_off.size() < std::min(max_blocks, pos_blocks) && (in_off.empty() || (in_off.size() + 1) * mbs <= out_budget)) {
            uint8_t bi[4];
            const int rc = read_exact(bi, 4);
            if (rc != 0) {
                if (io_failed) fail(-LZ4FLEX_FE_IO);
                else pending_zero = true;                    // UnexpectedEof on the block header => Ok(0), the frame stays open (:231-238)
                break;
            }
            const uint32_t size = rd32(bi);
            if (size == 0) { saw_end = true; break; }
            const bool raw = (size & BLOCK_UNCOMPRESSED_SIZE_BIT) != 0;
            const size_t len = size & ~BLOCK_UNCOMPRESSED_SIZE_BIT;
            if (len > mbs) { fail(-LZ4FLEX_FE_BLOCK_TOO_BIG); break; }
            const size_t at = comp.size();
            comp.resize(at + len);
            if (len && read_exact(comp.data() + at, len) != 0) { comp.resize(at); fail(-LZ4FLEX_FE_IO); break; }
            if (fi.block_checksums) {
                uint8_t c[4];
                if (read_exact(c, 4) != 0) { comp.resize(at); fail(-LZ4FLEX_FE_IO); break; }
                if (XxHash32::oneshot(0, comp.data() + at, len) != rd32(c)) { comp.resize(at); fail(-LZ4FLEX_FE_BLOCK_CHECKSUM); break; }
            }
            if (raw) { raw_tail = true; raw_at = at; raw_len = len; break; }
            in_off.push_back(at); in_len.push_back((uint32_t)len);
        }
        const size_t nb = in_off.size();
        if (ldst.size() < carry + (nb + 1) * mbs) ldst.resize(carry + (nb + 1) * mbs);
        size_t produced_total = 0;                           // bytes of this call behind the carry
        // ---- the run of compressed blocks: launches until every block sits behind its real predecessor
        out_len.assign(nb, 0); status.assign(nb, 0); detail.assign(2 * nb, 0);
        std::vector<uint32_t> pos(nb), cap(nb);
        std::vector<uint64_t> zero_off(nb, 0);
        size_t first = 0;
        bool failed = false;
        // A launch assumes full blocks; the blocks behind a short one have to be decoded again from their real position.  A
        // frame of many short blocks (a flush() after every small write) would cost a launch and a re-decode of everything
        // behind it per block if every launch took the whole rest of the run: the run length is cut to a quarter whenever a
        // short block invalidates the blocks behind it and doubles again while launches end clean -- launches and decoded
        // blocks stay linear in the number of blocks.
        size_t run = nb;
        while (first < nb) {
            const size_t m = std::min(run, nb - first);
            for (size_t k = 0; k < m; k++) {
                pos[k] = (uint32_t)(carry + produced_total + k * mbs);
                cap[k] = pos[k] + (uint32_t)mbs;
            }
            lz4flex_decompress_ext ext{};
            ext.out_pos = pos.data();
            const int rc = lz4flex_decompress_batch_ex(nullptr, comp.data(), in_off.data() + first, in_len.data() + first, (uint32_t)m,
                                                       ldst.data(), zero_off.data(), cap.data(), out_len.data() + first, status.data() + first,
                                                       detail.data() + 2 * first, &ext, LZ4FLEX_MEM_HOST | LZ4FLEX_MEM_CHAINED, nullptr);
            if (rc) { fail(rc); pending_zero = false; failed = true; break; }
            size_t k = 0;
            for (; k < m; k++) {
                const size_t i = first + k;
                if (status[i] != 0) {
                    lz4flex_err_detail d{}; d.expected = detail[2 * i]; d.actual = detail[2 * i + 1]; d.inner = status[i];
                    fail(-LZ4FLEX_FE_DECOMPRESSION, &d);
                    pending_zero = false;
                    failed = true;
                    break;
                }
                produced_total += out_len[i];
                if (out_len[i] != mbs) { k++; break; }        // the blocks behind a short one were given a wrong position
            }
            run = k < m ? std::max<size_t>(1, m / 4) : std::min(nb, 2 * m);
            first += k;
            if (failed) break;
        }
        // ---- a block stored raw behind the run
        if (!failed && !pending_err && raw_tail) {
            if (ldst.size() < carry + produced_total + raw_len) ldst.resize(carry + produced_total + raw_len);
            memcpy(ldst.data() + carry + produced_total, comp.data() + raw_at, raw_len);
            produced_total += raw_len;
        }
        // ---- hand the bytes out where they are (ready items index `ldst` in Linked mode); they stay there as history
        if (produced_total) ready.push_back({carry, produced_total});
        else if (!failed && !pending_err && (nb != 0 || raw_tail)) ready.push_back({carry, 0});     // (an empty block: read_more() == 0)
        content_len += produced_total;
        if (fi.content_checksum && produced_total) content_hasher.write(ldst.data() + carry, produced_total);
        lhave = carry + produced_total;
        if (pending_err) { pending_zero = false; return; }
        if (saw_end && !pending_zero) end_mark();
    }

    // io::Read::read, frame/decompress.rs:353-367
    // io::BufRead::fill_buf, frame/decompress.rs:410-416: the decoded bytes not consumed yet (decodes more when there are
    // none); *p stays valid until the next call on this decoder.  0 = end of frame / EOF, < 0 = -code.
    int64_t fill_buf(const uint8_t** p, lz4flex_err_detail* d) {
        for (;;) {
            while (ready_idx < ready.size()) {
                const Item& it = ready[ready_idx];
                if (it.len == 0 || ready_pos == it.len) {
                    const bool empty_block = it.len == 0;
                    ready_idx++; ready_pos = 0;
                    if (empty_block) { *p = out.data(); return 0; }      // read_more() == 0: an empty slice
                    continue;
                }
                *p = ready_base() + it.off + ready_pos;
                return (int64_t)(it.len - ready_pos);
            }
            if (pending_err) { if (d) *d = pending_detail; return pending_err; }
            if (pending_zero) { pending_zero = false; *p = out.data(); return 0; }
            if (!have_frame) {
                lz4flex_err_detail hd{};
                const int rc = read_frame_info(&hd);
                if (rc == 1) { *p = out.data(); return 0; }
                if (rc < 0) { if (d) *d = hd; return rc; }
            }
            if (fi.block_mode == 1) read_block_linked(); else read_blocks_independent();
        }
    }
    // io::BufRead::consume, :418-421 (the reference asserts amt <= available)
    int consume(size_t amt) {
        if (amt == 0) return 0;
        if (ready_idx >= ready.size()) return -LZ4FLEX_E_INVALID_ARG;
        const Item& it = ready[ready_idx];
        if (amt > it.len - ready_pos) return -LZ4FLEX_E_INVALID_ARG;
        ready_pos += amt;
        if (ready_pos == it.len) { ready_idx++; ready_pos = 0; }
        return 0;
    }
    int64_t read(uint8_t* buf, size_t len, lz4flex_err_detail* d) {
        for (;;) {
            while (ready_idx < ready.size()) {
                const Item& it = ready[ready_idx];
                if (it.len == 0) { ready_idx++; ready_pos = 0; return 0; }   // read_more() == 0
                const size_t n = std::min(it.len - ready_pos, len);
                if (n == 0) return 0;   // zero-length destination
                memcpy(buf, ready_base() + it.off + ready_pos, n);
                ready_pos += n;
                if (ready_pos == it.len) { ready_idx++; ready_pos = 0; }
                return (int64_t)n;
            }
            if (pending_err) { if (d) *d = pending_detail; return pending_err; }
            if (pending_zero) { pending_zero = false; return 0; }
            if (!have_frame) {
                lz4flex_err_detail hd{};
                const int rc = read_frame_info(&hd);
                if (rc == 1) return 0;
                if (rc < 0) { if (d) *d = hd; return rc; }
            }
            if (fi.block_mode == 1) read_block_linked();
            else if (len >= block_size_bytes(fi.block_size) && len >= DIRECT_MIN) {
                const size_t n = read_blocks_independent(buf, len);
                if (n) return (int64_t)n;
            } else read_blocks_independent();
        }
    }
    static constexpr size_t DIRECT_MIN = 1u << 20;    // a reader with less room than this gets its bytes from the staging buffer (small reads: one launch per read would cost more than the copy)
};

extern "C" {

lz4flex_frame_encoder* lz4flex_frame_encoder_new(const lz4flex_frame_info* info, lz4flex_write_fn w, void* user) {
    if (!w) return nullptr;
    lz4flex_frame_encoder* e = new (std::nothrow) lz4flex_frame_encoder();
    if (!e) return nullptr;
    if (info) e->fi = *info;
    e->w = w; e->user = user;
    return e;
}
int64_t lz4flex_frame_encoder_write(lz4flex_frame_encoder* e, const uint8_t* buf, size_t len) {
    if (!e || (!buf && len)) return -LZ4FLEX_E_INVALID_ARG;
    return e->write(buf, len);
}
int lz4flex_frame_encoder_flush(lz4flex_frame_encoder* e) { return e ? e->flush() : -LZ4FLEX_E_INVALID_ARG; }
int lz4flex_frame_encoder_try_finish(lz4flex_frame_encoder* e, lz4flex_err_detail* d) {
    if (d) memset(d, 0, sizeof *d);
    return e ? e->try_finish(d) : -LZ4FLEX_E_INVALID_ARG;
}
void lz4flex_frame_encoder_frame_info(lz4flex_frame_encoder* e, lz4flex_frame_info* out) { if (e && out) *out = e->fi; }
int lz4flex_frame_encoder_set_batch_bytes(lz4flex_frame_encoder* e, size_t bytes) {
    if (!e || e->is_frame_open || bytes == 0) return -LZ4FLEX_E_INVALID_ARG;
    e->batch_bytes = bytes;
    e->batch_auto = false;
    return 0;
}
void lz4flex_frame_encoder_free(lz4flex_frame_encoder* e) { delete e; }

lz4flex_frame_decoder* lz4flex_frame_decoder_new(lz4flex_read_fn r, void* user) {
    if (!r) return nullptr;
    lz4flex_frame_decoder* d = new (std::nothrow) lz4flex_frame_decoder();
    if (!d) return nullptr;
    d->r = r; d->user = user;
    return d;
}
int64_t lz4flex_frame_decoder_read(lz4flex_frame_decoder* dcd, uint8_t* buf, size_t len, lz4flex_err_detail* d) {
    if (!dcd || (!buf && len)) return -LZ4FLEX_E_INVALID_ARG;
    if (d) memset(d, 0, sizeof *d);
    return dcd->read(buf, len, d);
}
int64_t lz4flex_frame_decoder_fill_buf(lz4flex_frame_decoder* dcd, const uint8_t** buf, lz4flex_err_detail* d) {
    if (!dcd || !buf) return -LZ4FLEX_E_INVALID_ARG;
    if (d) memset(d, 0, sizeof *d);
    return dcd->fill_buf(buf, d);
}
int lz4flex_frame_decoder_consume(lz4flex_frame_decoder* dcd, size_t amt) {
    if (!dcd) return -LZ4FLEX_E_INVALID_ARG;
    return dcd->consume(amt);
}
int lz4flex_frame_decoder_set_batch_bytes(lz4flex_frame_decoder* d, size_t bytes) {
    if (!d || bytes == 0) return -LZ4FLEX_E_INVALID_ARG;
    d->batch_bytes = bytes;
    d->batch_auto = false;
    return 0;
}
void lz4flex_frame_decoder_free(lz4flex_frame_decoder* d) { delete d; }

// ---- one-shot helpers over flat buffers -------------------------------------------------------
namespace {
struct FlatW { uint8_t* p; size_t pos, cap; bool full; };
int64_t flat_write(void* u, const uint8_t* b, size_t n) {
    FlatW* f = (FlatW*)u;
    if (f->cap - f->pos < n) { f->full = true; return -1; }
    memcpy(f->p + f->pos, b, n); f->pos += n;
    return (int64_t)n;
}
struct FlatR { const uint8_t* p; size_t pos, len; };
int64_t flat_read(void* u, uint8_t* b, size_t n) {
    FlatR* f = (FlatR*)u;
    const size_t k = std::min(n, f->len - f->pos);
    memcpy(b, f->p + f->pos, k); f->pos += k;
    return (int64_t)k;
}
}  // namespace

size_t lz4flex_frame_compress_bound(size_t in_len, const lz4flex_frame_info* info) {
    int bs = info ? info->block_size : 0;
    if (bs == 0) bs = block_size_from_buf_length(in_len);
    const size_t mbs = block_size_bytes(bs) ? block_size_bytes(bs) : 65536;
    const size_t nblk = (in_len + mbs - 1) / mbs;
    return MAX_FRAME_INFO_SIZE + in_len + nblk * 8 + 8;   // raw-stored worst case: 4 B header (+4 B checksum) per block
}

int64_t lz4flex_frame_compress(const uint8_t* in, size_t in_len, const lz4flex_frame_info* info, uint8_t* out,
                               size_t out_cap, lz4flex_err_detail* detail) {
    if ((!in && in_len) || !out) return -LZ4FLEX_E_INVALID_ARG;
    if (detail) memset(detail, 0, sizeof *detail);
    FlatW fw{out, 0, out_cap, false};
    lz4flex_frame_encoder* e = lz4flex_frame_encoder_new(info, flat_write, &fw);
    if (!e) return -LZ4FLEX_E_NOMEM;
    e->final_write = true;                     // (whole blocks AND the remainder are compressed where they lie: no staging copy)
    int64_t rc = e->write(in, in_len);
    if (rc >= 0) rc = e->try_finish(detail);
    lz4flex_frame_encoder_free(e);
    if (rc < 0) return fw.full ? -LZ4FLEX_FE_OUTPUT_FULL : rc;
    return (int64_t)fw.pos;
}

int64_t lz4flex_frame_decompress(const uint8_t* in, size_t in_len, uint8_t* out, size_t out_cap, size_t* consumed,
                                 lz4flex_err_detail* detail) {
    if ((!in && in_len) || (!out && out_cap)) return -LZ4FLEX_E_INVALID_ARG;
    if (detail) memset(detail, 0, sizeof *detail);
    FlatR fr{in, 0, in_len};
    lz4flex_frame_decoder* d = lz4flex_frame_decoder_new(flat_read, &fr);
    if (!d) return -LZ4FLEX_E_NOMEM;
    size_t pos = 0;
    int64_t rc = 0;
    uint8_t scratch[1];
    for (;;) {   // read_to_end: until read() returns 0
        if (pos == out_cap) {
            // probe for more data without a destination
            rc = d->read(scratch, 1, detail);
            if (rc > 0) rc = -LZ4FLEX_FE_OUTPUT_FULL;
            break;
        }
        rc = d->read(out + pos, out_cap - pos, detail);
        if (rc <= 0) break;
        pos += (size_t)rc;
    }
    if (consumed) *consumed = fr.pos;
    lz4flex_frame_decoder_free(d);
    return rc < 0 ? rc : (int64_t)pos;
}

}  // extern "C"
