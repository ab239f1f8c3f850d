/*
 * lz4flex_oracle.h -- CPU restatement of lz4_flex's LZ4 block codec and frame layer.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the parity checker ("oracle") for the HIP
 * product path.  Only tests/, __graft_entry__.smoke() and bench.py's
 * `cpu_baseline` leg may load it.  Nothing under lz4_flex_amd/ links, imports
 * or calls it.
 *
 * Every function cites the reference file:line (relative to /root/reference)
 * whose behaviour it restates.  The reference is Rust; no Rust toolchain is
 * present in this image, so oracle/_ref (a build of the real reference) does
 * not exist.  Pinning status (see DESIGN.md "Oracle"):
 *   - decoder: pinned bit-exact by the reference's own known-answer tests
 *     (src/block/decompress.rs:534-622) incl. the exact error variant;
 *   - encoder: byte output is NOT pinned by any reference golden vector (the
 *     reference holds none); it is pinned to the reference's ratio ceilings
 *     (tests/tests.rs:159-192), its end-of-block conformance tests
 *     (src/block/compress.rs:952-988) and to cross-decoding by C liblz4 1.9.3,
 *     the same library the reference tests against (tests/tests.rs:25-56).
 *   - frame header checksum: pinned by the golden headers in
 *     fuzz/fuzz_targets/fuzz_decomp_corrupt_frame.rs:26-27.
 */
#ifndef LZ4FLEX_ORACLE_H
#define LZ4FLEX_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* DecompressError variants, enum order of src/block/mod.rs:82-98. Returned negated. */
enum {
    LZ4O_OK = 0,
    LZ4O_E_OUTPUT_TOO_SMALL = 1,
    LZ4O_E_LITERAL_OUT_OF_BOUNDS = 2,
    LZ4O_E_EXPECTED_ANOTHER_BYTE = 3,
    LZ4O_E_OFFSET_ZERO = 4,
    LZ4O_E_OFFSET_OUT_OF_BOUNDS = 5,
};

/* frame::Error variants, src/frame/mod.rs:35-72 (codes continue after the block ones). */
enum {
    LZ4O_FE_COMPRESSION = 16,
    LZ4O_FE_DECOMPRESSION = 17,
    LZ4O_FE_IO = 18,
    LZ4O_FE_UNSUPPORTED_BLOCKSIZE = 19,
    LZ4O_FE_UNSUPPORTED_VERSION = 20,
    LZ4O_FE_WRONG_MAGIC = 21,
    LZ4O_FE_RESERVED_BITS = 22,
    LZ4O_FE_INVALID_BLOCK_INFO = 23,
    LZ4O_FE_BLOCK_TOO_BIG = 24,
    LZ4O_FE_HEADER_CHECKSUM = 25,
    LZ4O_FE_BLOCK_CHECKSUM = 26,
    LZ4O_FE_CONTENT_CHECKSUM = 27,
    LZ4O_FE_SKIPPABLE_FRAME = 28,
    LZ4O_FE_DICTIONARY_NOT_SUPPORTED = 29,
    LZ4O_FE_CONTENT_LENGTH = 30,
    LZ4O_FE_OUTPUT_FULL = 31, /* oracle-only: caller's flat output buffer too small */
};

typedef struct {
    uint64_t expected; /* OutputTooSmall{expected,..} / ContentLengthError{expected,..} / SkippableFrame(len) */
    uint64_t actual;   /* OutputTooSmall{..,actual}   / ContentLengthError{..,actual}   */
    int32_t inner;     /* frame: the DecompressError code wrapped by DecompressionError */
} lz4o_err_detail;

/* ---- block ---- */

/* src/block/compress.rs:588-590 */
size_t lz4o_get_maximum_output_size(size_t input_len);

/* src/block/compress.rs:599-601 (compress_into).  >=0 bytes written, <0 = -LZ4O_E_OUTPUT_TOO_SMALL */
int64_t lz4o_compress_into(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap);

/* src/block/compress.rs:610-616 (compress_into_with_dict) */
int64_t lz4o_compress_into_with_dict(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap,
                                     const uint8_t *dict, size_t dict_len);

/* What FrameEncoder::write_block runs for block k of an Independent frame
 * (src/frame/compress.rs:261-371 with :357-367): compress_internal<HashTable4K,false> on a
 * table whose entries are all unreachable (first_block == 0) or freshly zeroed with
 * stream offset 0 (first_block != 0).  See SURVEY.md N3. */
int64_t lz4o_compress_frame_block(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap,
                                  int first_block);

/* src/block/decompress.rs:454-456 (decompress_into), unsafe flavour check order. */
int64_t lz4o_decompress_into(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap,
                             lz4o_err_detail *detail);

/* src/block/decompress.rs:462-468 (decompress_into_with_dict) */
int64_t lz4o_decompress_into_with_dict(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap,
                                       const uint8_t *dict, size_t dict_len, lz4o_err_detail *detail);

/* src/block/decompress.rs:192-195 */
int lz4o_does_token_fit(uint8_t token);

/* src/block/compress.rs:156-216 (count_same_bytes); returns the count and advances *cur. */
size_t lz4o_count_same_bytes(const uint8_t *input, size_t input_len, size_t *cur,
                             const uint8_t *source, size_t source_len, size_t candidate);

/* ---- XXH32 (third-party twox-hash 2.x in the reference; public XXH32 spec restated) ---- */
uint32_t lz4o_xxh32(const uint8_t *data, size_t len, uint32_t seed);

/* ---- frame ---- */
typedef struct {
    int has_content_size;
    uint64_t content_size;
    int block_size;  /* 0 = Auto, 4 = 64KB, 5 = 256KB, 6 = 1MB, 7 = 4MB, 8 = 8MB (legacy decode only) */
    int block_mode;  /* 0 = Independent, 1 = Linked */
    int block_checksums;
    int content_checksum;
    int legacy_frame;
} lz4o_frame_info;

/* FrameInfo::write, src/frame/header.rs:232-275.  Returns bytes written or -code. */
int64_t lz4o_frame_info_write(const lz4o_frame_info *fi, uint8_t *out, size_t out_cap);
/* FrameInfo::read, src/frame/header.rs:277-373. Returns header size consumed or -code. */
int64_t lz4o_frame_info_read(const uint8_t *in, size_t in_len, lz4o_frame_info *fi, lz4o_err_detail *d);

/* One FrameEncoder lifetime: with_frame_info; write_all(chunks...); finish()
 * (src/frame/compress.rs:95-404).  `chunk_lens` gives the sizes of the successive
 * write() calls (NULL => a single write of in_len bytes).  Returns bytes or -code. */
int64_t lz4o_frame_compress(const uint8_t *in, size_t in_len, const size_t *chunk_lens, size_t n_chunks,
                            const lz4o_frame_info *fi, uint8_t *out, size_t out_cap, lz4o_err_detail *d);

/* FrameDecoder::new(in).read_to_end() ONCE (src/frame/decompress.rs:352-408): decodes the
 * first frame, stops at its EndMark.  *consumed = input bytes read.  Returns bytes or -code. */
int64_t lz4o_frame_decompress(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap,
                              size_t *consumed, lz4o_err_detail *d);

/* ---- cpu_baseline helper: batch over independent blocks with `threads` pthreads ---- */
/* dir: 0 = compress, 1 = decompress.  Returns wall seconds of the best of `reps` passes. */
double lz4o_bench_batch(int dir, const uint8_t *in_base, const uint64_t *in_off, const uint32_t *in_len,
                        uint8_t *out_base, const uint64_t *out_off, const uint32_t *out_cap,
                        uint32_t *out_len, uint32_t n_blocks, int threads, int reps);

/* the same batch loop around a foreign codec with liblz4's signature int f(const char*, char*, int, int)
 * (LZ4_compress_default / LZ4_decompress_safe of the system liblz4 1.9.3, loaded by the caller) */
double lz4o_bench_batch_fn(int dir, void *fn, const uint8_t *in_base, const uint64_t *in_off, const uint32_t *in_len,
                           uint8_t *out_base, const uint64_t *out_off, const uint32_t *out_cap,
                           uint32_t *out_len, uint32_t n_blocks, int threads, int reps);

/* The measurement proper: a pool of `threads` workers created once, `reps` timed passes of >= min_pass_s seconds each (a
 * pass repeats the sweep over the batch as often as that takes), returns the best time of ONE sweep.  *used (nullable) =
 * worker threads that really ran. */
double lz4o_bench_pool(int dir, void *fn, const uint8_t *in_base, const uint64_t *in_off, const uint32_t *in_len,
                       uint8_t *out_base, const uint64_t *out_off, const uint32_t *out_cap, uint32_t *out_len,
                       uint32_t n_blocks, int threads, int reps, double min_pass_s, int *used);

#ifdef __cplusplus
}
#endif
#endif
