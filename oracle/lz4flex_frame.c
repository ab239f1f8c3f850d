/*
 * lz4flex_frame.c -- ORACLE (test infrastructure, never shipped): plain-C restatement of
 * lz4_flex's frame layer over flat buffers, plus XXH32.
 *
 * Follows (paths relative to /root/reference):
 *   header   src/frame/header.rs:11-34 (constants), :57-78 (BlockSize), :232-275 (write),
 *            :277-373 (read), :376-411 (BlockInfo)
 *   encoder  src/frame/compress.rs:96-118 (init), :166-187 (finish), :209-230 (end_frame),
 *            :234-257 (begin_frame), :261-371 (write_block), :375-403 (write/flush)
 *   decoder  src/frame/decompress.rs:109-168 (read_frame_info), :189-342 (read_block),
 *            :352-408 (read loop)
 *   XXH32    third-party crate twox-hash 2.x (Cargo.toml:50, not vendored, no Cargo.lock);
 *            the public XXH32 algorithm (xxHash spec, seed/primes/rounds/avalanche) restated.
 *            Pinned by the header goldens 60 40 -> 82 and 40 40 -> C0
 *            (fuzz/fuzz_targets/fuzz_decomp_corrupt_frame.rs:26-27) and cross-checked against
 *            the python `xxhash` module in tests/.
 */
#include "lz4flex_oracle.h"

#include <stdlib.h>
#include <string.h>

/* internals of lz4flex_block.c */
void *lz4o__table_new(void);
void lz4o__table_free(void *t);
void lz4o__table_clear(void *t);
void lz4o__table_reposition(void *t, uint32_t offset);
int64_t lz4o__compress_internal(const uint8_t *input, size_t input_len, size_t input_pos, uint8_t *out,
                                size_t out_cap, void *table, int use_dict, const uint8_t *ext_dict,
                                size_t ext_dict_len, size_t input_stream_offset);
int64_t lz4o__decompress_internal(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_pos, size_t out_cap,
                                  int use_dict, const uint8_t *ext_dict, size_t ext_dict_len,
                                  lz4o_err_detail *detail);

#define WINDOW_SIZE 65536u

/* ---------------------------------------------------------------------------------------- */
/* XXH32 */
#define P1 2654435761u
#define P2 2246822519u
#define P3 3266489917u
#define P4 668265263u
#define P5 374761393u
static inline uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
static inline uint32_t rd32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline uint32_t xxround(uint32_t acc, uint32_t in) { return rotl32(acc + in * P2, 13) * P1; }

uint32_t lz4o_xxh32(const uint8_t *p, size_t len, uint32_t seed) {
    const uint8_t *end = p + len;
    uint32_t h;
    if (len >= 16) {
        uint32_t v1 = seed + P1 + P2, v2 = seed + P2, v3 = seed, v4 = seed - P1;
        const uint8_t *limit = end - 16;
        do {
            v1 = xxround(v1, rd32(p)); v2 = xxround(v2, rd32(p + 4));
            v3 = xxround(v3, rd32(p + 8)); v4 = xxround(v4, rd32(p + 12));
            p += 16;
        } while (p <= limit);
        h = rotl32(v1, 1) + rotl32(v2, 7) + rotl32(v3, 12) + rotl32(v4, 18);
    } else {
        h = seed + P5;
    }
    h += (uint32_t)len;
    while (p + 4 <= end) { h = rotl32(h + rd32(p) * P3, 17) * P4; p += 4; }
    while (p < end) { h = rotl32(h + (*p) * P5, 11) * P1; p++; }
    h ^= h >> 15; h *= P2; h ^= h >> 13; h *= P3; h ^= h >> 16;
    return h;
}

/* ---------------------------------------------------------------------------------------- */
/* header */
/* src/frame/header.rs:11-34 */
#define FLG_RESERVED_MASK 0x02
#define FLG_VERSION_MASK 0xC0
#define FLG_SUPPORTED_VERSION_BITS 0x40
#define FLG_INDEPENDENT_BLOCKS 0x20
#define FLG_BLOCK_CHECKSUMS 0x10
#define FLG_CONTENT_SIZE 0x08
#define FLG_CONTENT_CHECKSUM 0x04
#define FLG_DICTIONARY_ID 0x01
#define BD_BLOCK_SIZE_MASK 0x70
#define BD_RESERVED_MASK 0x8F
#define BLOCK_UNCOMPRESSED_SIZE_BIT 0x80000000u
#define LZ4F_MAGIC_NUMBER 0x184D2204u
#define LZ4F_LEGACY_MAGIC_NUMBER 0x184C2102u
#define MIN_FRAME_INFO_SIZE 7
#define MAX_FRAME_INFO_SIZE 19

/* src/frame/header.rs:68-77 */
static size_t block_size_bytes(int code) {
    switch (code) {
        case 4: return 64u * 1024;
        case 5: return 256u * 1024;
        case 6: return 1024u * 1024;
        case 7: return 4u * 1024 * 1024;
        case 8: return 8u * 1024 * 1024;
        default: return 0;
    }
}
/* src/frame/header.rs:57-67 */
static int block_size_from_buf_length(size_t buf_len) {
    if (buf_len > 256u * 1024) return 7;
    if (buf_len > 64u * 1024) return 5;
    return 4;
}
static void wr32(uint8_t *p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }
static void wr64(uint8_t *p, uint64_t v) { wr32(p, (uint32_t)v); wr32(p + 4, (uint32_t)(v >> 32)); }
static uint64_t rd64le(const uint8_t *p) { return (uint64_t)rd32(p) | ((uint64_t)rd32(p + 4) << 32); }

/* src/frame/header.rs:232-275 */
int64_t lz4o_frame_info_write(const lz4o_frame_info *fi, uint8_t *out, size_t out_cap) {
    size_t write_size = MIN_FRAME_INFO_SIZE + (fi->has_content_size ? 8 : 0);
    if (out_cap < write_size) return -LZ4O_FE_IO;
    uint8_t b[MAX_FRAME_INFO_SIZE] = {0};
    wr32(b, LZ4F_MAGIC_NUMBER);
    b[4] = FLG_SUPPORTED_VERSION_BITS;
    if (fi->block_checksums) b[4] |= FLG_BLOCK_CHECKSUMS;
    if (fi->content_checksum) b[4] |= FLG_CONTENT_CHECKSUM;
    if (fi->block_mode == 0) b[4] |= FLG_INDEPENDENT_BLOCKS;
    b[5] = (uint8_t)(fi->block_size << 4);
    size_t off = 6;
    if (fi->has_content_size) { b[4] |= FLG_CONTENT_SIZE; wr64(b + off, fi->content_size); off += 8; }
    b[off] = (uint8_t)(lz4o_xxh32(b + 4, off - 4, 0) >> 8);
    off += 1;
    memcpy(out, b, write_size);
    return (int64_t)write_size;
}

/* src/frame/header.rs:277-373; `in` must hold the whole header (caller sized it via read_size) */
int64_t lz4o_frame_info_read(const uint8_t *in, size_t in_len, lz4o_frame_info *fi, lz4o_err_detail *d) {
    memset(fi, 0, sizeof *fi);
    if (in_len < 4) return -LZ4O_FE_IO;
    uint32_t magic = rd32(in);
    size_t p = 4;
    if (magic == LZ4F_LEGACY_MAGIC_NUMBER) { fi->block_size = 8; fi->legacy_frame = 1; return 4; }
    if (magic >= 0x184D2A50u && magic <= 0x184D2A5Fu) {
        if (in_len < 8) return -LZ4O_FE_IO;
        if (d) d->expected = rd32(in + 4);
        return -LZ4O_FE_SKIPPABLE_FRAME;
    }
    if (magic != LZ4F_MAGIC_NUMBER) return -LZ4O_FE_WRONG_MAGIC;
    if (in_len < p + 2) return -LZ4O_FE_IO;
    uint8_t flg = in[p], bd = in[p + 1];
    p += 2;
    if ((flg & FLG_VERSION_MASK) != FLG_SUPPORTED_VERSION_BITS) {
        if (d) d->expected = flg & FLG_VERSION_MASK;
        return -LZ4O_FE_UNSUPPORTED_VERSION;
    }
    if ((flg & FLG_RESERVED_MASK) != 0 || (bd & BD_RESERVED_MASK) != 0) return -LZ4O_FE_RESERVED_BITS;
    fi->block_mode = (flg & FLG_INDEPENDENT_BLOCKS) ? 0 : 1;
    fi->content_checksum = (flg & FLG_CONTENT_CHECKSUM) != 0;
    fi->block_checksums = (flg & FLG_BLOCK_CHECKSUMS) != 0;
    int bs = (bd & BD_BLOCK_SIZE_MASK) >> 4;
    if (bs <= 3) { if (d) d->expected = (uint64_t)bs; return -LZ4O_FE_UNSUPPORTED_BLOCKSIZE; }
    fi->block_size = bs;
    if (flg & FLG_CONTENT_SIZE) {
        if (in_len < p + 8) return -LZ4O_FE_IO;
        fi->has_content_size = 1; fi->content_size = rd64le(in + p); p += 8;
    }
    int has_dict = 0;
    if (flg & FLG_DICTIONARY_ID) { if (in_len < p + 4) return -LZ4O_FE_IO; has_dict = 1; p += 4; }
    if (in_len < p + 1) return -LZ4O_FE_IO;
    uint8_t expected = in[p];
    if ((uint8_t)(lz4o_xxh32(in + 4, p - 4, 0) >> 8) != expected) return -LZ4O_FE_HEADER_CHECKSUM;
    p += 1;
    if (has_dict) return -LZ4O_FE_DICTIONARY_NOT_SUPPORTED;   /* src/frame/decompress.rs:139-142 */
    return (int64_t)p;
}

/* ---------------------------------------------------------------------------------------- */
/* encoder */
typedef struct {
    uint8_t *src; size_t src_len;            /* Vec<u8> src (len), capacity reserved by init() */
    size_t src_start, src_end, ext_dict_offset, ext_dict_len, src_stream_offset;
    void *table;
    uint8_t *w; size_t wpos, wcap;           /* the io::Write sink */
    uint64_t content_len;
    uint8_t *dst; size_t dst_cap;
    int is_frame_open, data_to_frame_written;
    lz4o_frame_info fi;
    const uint8_t *content_base;              /* all writes come from one flat input: content = base[..content_len] */
    int io_err;
} enc_t;

static int w_write_all(enc_t *e, const uint8_t *p, size_t n) {
    if (e->wcap - e->wpos < n) { e->io_err = 1; return -LZ4O_FE_OUTPUT_FULL; }
    memcpy(e->w + e->wpos, p, n); e->wpos += n; return 0;
}

/* src/frame/compress.rs:234-257 (+ init :96-118) */
static int enc_begin_frame(enc_t *e, size_t buf_len) {
    e->is_frame_open = 1;
    if (e->fi.block_size == 0) e->fi.block_size = block_size_from_buf_length(buf_len);
    size_t mbs = block_size_bytes(e->fi.block_size);
    size_t src_size = e->fi.block_mode == 1 ? mbs * 2 + WINDOW_SIZE : mbs;
    if (!e->src) { e->src = (uint8_t *)malloc(src_size); if (!e->src) return -LZ4O_FE_IO; }
    size_t need = lz4o_get_maximum_output_size(mbs);
    if (!e->dst) { e->dst = (uint8_t *)malloc(need); e->dst_cap = need; if (!e->dst) return -LZ4O_FE_IO; }
    uint8_t hdr[MAX_FRAME_INFO_SIZE];
    int64_t n = lz4o_frame_info_write(&e->fi, hdr, sizeof hdr);
    if (n < 0) return (int)n;
    return w_write_all(e, hdr, (size_t)n);
}

/* src/frame/compress.rs:261-371 */
static int enc_write_block(enc_t *e) {
    size_t mbs = block_size_bytes(e->fi.block_size);
    /* :266-271 */
    if (e->src_stream_offset + mbs + WINDOW_SIZE >= (size_t)(0xFFFFFFFFu / 2)) {
        lz4o__table_reposition(e->table, (uint32_t)(e->src_stream_offset - e->ext_dict_len));
        e->src_stream_offset = e->ext_dict_len;
    }
    const uint8_t *input = e->src;            /* &self.src[..self.src_end] */
    size_t input_len = e->src_end;
    const uint8_t *src = input + e->src_start;
    size_t src_len = input_len - e->src_start;
    size_t dst_required = lz4o_get_maximum_output_size(src_len);
    int64_t r;
    if (e->ext_dict_len != 0)
        r = lz4o__compress_internal(input, input_len, e->src_start, e->dst, dst_required, e->table, 1,
                                    e->src + e->ext_dict_offset, e->ext_dict_len, e->src_stream_offset);
    else
        r = lz4o__compress_internal(input, input_len, e->src_start, e->dst, dst_required, e->table, 0,
                                    (const uint8_t *)"", 0, e->src_stream_offset);
    if (r < 0) return -LZ4O_FE_COMPRESSION;
    /* :301-306 */
    const uint8_t *block_data; size_t block_len; uint32_t info;
    if ((size_t)r < src_len) { block_data = e->dst; block_len = (size_t)r; info = (uint32_t)r; }
    else { block_data = src; block_len = src_len; info = (uint32_t)src_len | BLOCK_UNCOMPRESSED_SIZE_BIT; }
    /* BlockInfo::write header.rs:396-410: Compressed(0) is InvalidBlockInfo (cannot happen: src_len>0) */
    uint8_t bi[4]; wr32(bi, info);
    int rc;
    if ((rc = w_write_all(e, bi, 4))) return rc;
    if ((rc = w_write_all(e, block_data, block_len))) return rc;
    if (e->fi.block_checksums) {               /* :313-316 */
        uint8_t c[4]; wr32(c, lz4o_xxh32(block_data, block_len, 0));
        if ((rc = w_write_all(e, c, 4))) return rc;
    }
    /* :319-321 content hasher: deferred, content == content_base[..content_len] */
    e->content_len += src_len;                 /* :324 */
    e->src_start += src_len;
    if (e->fi.block_mode == 1) {               /* :327-356 */
        if (e->src_start >= mbs + WINDOW_SIZE) {
            e->ext_dict_offset = e->src_end - WINDOW_SIZE;
            e->ext_dict_len = WINDOW_SIZE;
            e->src_stream_offset += e->src_end;
            e->src_start = 0; e->src_end = 0;
        } else if (e->src_start + e->ext_dict_len > WINDOW_SIZE) {
            size_t over = e->src_start + e->ext_dict_len - WINDOW_SIZE;
            size_t delta = e->ext_dict_len < over ? e->ext_dict_len : over;
            e->ext_dict_offset += delta;
            e->ext_dict_len -= delta;
        }
    } else {                                   /* :357-367 */
        e->src_start = 0; e->src_end = 0;
        e->src_stream_offset += src_len;
    }
    return 0;
}

/* src/frame/compress.rs:375-396 */
static int enc_write(enc_t *e, const uint8_t *buf, size_t len) {
    int rc;
    if (!e->is_frame_open && len != 0) { if ((rc = enc_begin_frame(e, len))) return rc; }
    while (len != 0) {
        size_t src_filled = e->src_end - e->src_start;
        size_t max_fill_len = block_size_bytes(e->fi.block_size) - src_filled;
        if (max_fill_len == 0) { if ((rc = enc_write_block(e))) return rc; continue; }
        size_t fill_len = max_fill_len < len ? max_fill_len : len;
        memcpy(e->src + e->src_end, buf, fill_len);   /* vec_copy_overwriting :462-471 */
        buf += fill_len; len -= fill_len;
        e->src_end += fill_len;
    }
    return 0;
}

int64_t lz4o_frame_compress(const uint8_t *in, size_t in_len, const size_t *chunk_lens, size_t n_chunks,
                            const lz4o_frame_info *fi, uint8_t *out, size_t out_cap, lz4o_err_detail *d) {
    enc_t e; memset(&e, 0, sizeof e);
    e.fi = *fi; e.w = out; e.wcap = out_cap; e.content_base = in;
    e.table = lz4o__table_new();
    int rc = 0;
    size_t one = in_len;
    if (!chunk_lens) { chunk_lens = &one; n_chunks = 1; }
    size_t pos = 0;
    for (size_t i = 0; i < n_chunks && !rc; i++) {
        size_t n = chunk_lens[i];
        if (n > in_len - pos) n = in_len - pos;
        rc = enc_write(&e, in + pos, n);
        pos += n;
    }
    /* finish(): try_finish :173-187 */
    if (!rc && e.src_start != e.src_end) rc = enc_write_block(&e);         /* flush :398-403 */
    if (!rc && !e.is_frame_open && !e.data_to_frame_written) rc = enc_begin_frame(&e, 0);
    if (!rc) {                                                             /* end_frame :209-230 */
        e.is_frame_open = 0;
        if (e.fi.has_content_size && e.fi.content_size != e.content_len) {
            if (d) { d->expected = e.fi.content_size; d->actual = e.content_len; }
            rc = -LZ4O_FE_CONTENT_LENGTH;
        }
    }
    if (!rc) { uint8_t z[4] = {0, 0, 0, 0}; rc = w_write_all(&e, z, 4); }
    if (!rc && e.fi.content_checksum) {
        uint8_t c[4]; wr32(c, lz4o_xxh32(e.content_base, (size_t)e.content_len, 0));
        rc = w_write_all(&e, c, 4);
    }
    lz4o__table_free(e.table); free(e.src); free(e.dst);
    return rc ? rc : (int64_t)e.wpos;
}

/* ---------------------------------------------------------------------------------------- */
/* decoder */
int64_t lz4o_frame_decompress(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap,
                              size_t *consumed, lz4o_err_detail *d) {
    size_t rp = 0, wp = 0;
    int64_t ret = 0;
    uint8_t *dst = NULL;
    if (consumed) *consumed = 0;
    /* read_frame_info, src/frame/decompress.rs:109-168 */
    size_t avail = in_len - rp;
    if (avail == 0) return 0;
    if (avail < 4) return -LZ4O_FE_IO;
    uint32_t magic = rd32(in + rp);
    size_t required;
    if (magic == LZ4F_LEGACY_MAGIC_NUMBER) {
        required = 4;
    } else {
        if (avail == 4) { if (consumed) *consumed = 4; return 0; }         /* r.read() -> 0 => Ok(0) */
        if (avail < MIN_FRAME_INFO_SIZE) return -LZ4O_FE_IO;
        /* FrameInfo::read_size header.rs:194-219 */
        if (magic >= 0x184D2A50u && magic <= 0x184D2A5Fu) required = 8;
        else if (magic != LZ4F_MAGIC_NUMBER) return -LZ4O_FE_WRONG_MAGIC;
        else {
            required = MIN_FRAME_INFO_SIZE;
            if (in[rp + 4] & FLG_CONTENT_SIZE) required += 8;
            if (in[rp + 4] & FLG_DICTIONARY_ID) required += 4;
        }
        if (avail < required) return -LZ4O_FE_IO;
    }
    lz4o_frame_info fi;
    int64_t hs = lz4o_frame_info_read(in + rp, required, &fi, d);
    if (hs < 0) return hs;
    rp += required;
    size_t mbs = block_size_bytes(fi.block_size);
    size_t dst_size = fi.block_mode == 1 ? mbs * 2 + WINDOW_SIZE : mbs;
    dst = (uint8_t *)malloc(dst_size ? dst_size : 1);
    if (!dst) return -LZ4O_FE_IO;
    size_t ext_dict_offset = 0, ext_dict_len = 0, dst_start = 0, dst_end = 0;
    uint64_t content_len = 0;
    /* content hasher: deferred over out[..wp] */
    for (;;) {
        /* read_block, :189-342 */
        if (fi.block_mode == 1) {                                          /* :195-222 */
            if (dst_start + mbs > dst_size) {
                ext_dict_offset = dst_start - WINDOW_SIZE;
                ext_dict_len = WINDOW_SIZE;
                dst_start = 0; dst_end = 0;
            } else if (dst_start + ext_dict_len > WINDOW_SIZE) {
                size_t over = dst_start + ext_dict_len - WINDOW_SIZE;
                size_t delta = ext_dict_len < over ? ext_dict_len : over;
                ext_dict_offset += delta; ext_dict_len -= delta;
            }
        } else { dst_start = 0; dst_end = 0; }
        if (in_len - rp < 4) { rp = in_len; break; }                        /* UnexpectedEof => Ok(0) :231-238 */
        uint32_t size = rd32(in + rp); rp += 4;
        if (size == 0) {                                                    /* EndMark :313-332 */
            if (fi.has_content_size && content_len != fi.content_size) {
                if (d) { d->expected = fi.content_size; d->actual = content_len; }
                ret = -LZ4O_FE_CONTENT_LENGTH; goto done;
            }
            if (fi.content_checksum) {
                if (in_len - rp < 4) { ret = -LZ4O_FE_IO; goto done; }
                uint32_t expected = rd32(in + rp); rp += 4;
                if (lz4o_xxh32(out, wp, 0) != expected) { ret = -LZ4O_FE_CONTENT_CHECKSUM; goto done; }
            }
            break;
        }
        size_t produced;
        if (size & BLOCK_UNCOMPRESSED_SIZE_BIT) {                           /* :243-265 */
            size_t len = size & ~BLOCK_UNCOMPRESSED_SIZE_BIT;
            if (len > mbs) { ret = -LZ4O_FE_BLOCK_TOO_BIG; goto done; }
            if (in_len - rp < len) { ret = -LZ4O_FE_IO; goto done; }
            memcpy(dst + dst_start, in + rp, len); rp += len;
            if (fi.block_checksums) {
                if (in_len - rp < 4) { ret = -LZ4O_FE_IO; goto done; }
                uint32_t expected = rd32(in + rp); rp += 4;
                if (lz4o_xxh32(dst + dst_start, len, 0) != expected) { ret = -LZ4O_FE_BLOCK_CHECKSUM; goto done; }
            }
            produced = len;
        } else {                                                            /* :266-311 */
            size_t len = size;
            if (len > mbs) { ret = -LZ4O_FE_BLOCK_TOO_BIG; goto done; }
            if (in_len - rp < len) { ret = -LZ4O_FE_IO; goto done; }
            const uint8_t *src = in + rp; rp += len;
            if (fi.block_checksums) {
                if (in_len - rp < 4) { ret = -LZ4O_FE_IO; goto done; }
                uint32_t expected = rd32(in + rp); rp += 4;
                if (lz4o_xxh32(src, len, 0) != expected) { ret = -LZ4O_FE_BLOCK_CHECKSUM; goto done; }
            }
            int with_dict = fi.block_mode == 1 && ext_dict_len != 0;
            lz4o_err_detail bd; memset(&bd, 0, sizeof bd);
            int64_t r;
            if (with_dict)   /* sink = dst[..ext_dict_offset], pos = dst_start */
                r = lz4o__decompress_internal(src, len, dst, dst_start, ext_dict_offset, 1,
                                              dst + ext_dict_offset, ext_dict_len, &bd);
            else             /* sink = dst[..dst_start+mbs], pos = dst_start */
                r = lz4o__decompress_internal(src, len, dst, dst_start, dst_start + mbs, 0,
                                              (const uint8_t *)"", 0, &bd);
            if (r < 0) {
                if (d) { *d = bd; d->inner = (int32_t)(-r); }
                ret = -LZ4O_FE_DECOMPRESSION; goto done;
            }
            produced = (size_t)r;
        }
        dst_end = dst_start + produced;
        content_len += produced;
        if (produced == 0) break;   /* read_more() == 0 ends read_to_end (:344-349, :385-399) */
        if (out_cap - wp < produced) { ret = -LZ4O_FE_OUTPUT_FULL; goto done; }
        memcpy(out + wp, dst + dst_start, produced); wp += produced;
        dst_start = dst_end;                                                /* caller consumed it (read loop) */
    }
    ret = (int64_t)wp;
done:
    if (consumed) *consumed = rp;
    free(dst);
    return ret;
}
