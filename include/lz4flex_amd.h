/*
 * lz4flex_amd.h -- C ABI of the MI355X-native LZ4 block codec (drop-in boundary).
 *
 * lz4_flex has no FFI of its own: its boundary IS its public Rust API.  Each entry point
 * below names the lz4_flex item it replaces (paths relative to the lz4_flex v0.12.0 tree);
 * INTEGRATION.md shows the Rust `extern "C"` shim a maintainer would add on the reference
 * side.  Plain pointers and sizes only; no torch / C++ types.
 *
 * Every compute entry point runs hand-written HIP kernels on the GPU.  There is NO CPU
 * fallback: without a usable HIP device the calls return -LZ4FLEX_E_NO_DEVICE / -LZ4FLEX_E_HIP.
 *
 * Return convention for scalar calls: >= 0 is the byte count (Ok(usize)), < 0 is -code (Err).
 */
#ifndef LZ4FLEX_AMD_H
#define LZ4FLEX_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- error codes ------------------------------------------------------------------------- */
/* block::DecompressError, variant order of src/block/mod.rs:82-98 */
#define LZ4FLEX_OK 0
#define LZ4FLEX_E_OUTPUT_TOO_SMALL 1       /* also block::CompressError::OutputTooSmall, mod.rs:103-106 */
#define LZ4FLEX_E_LITERAL_OUT_OF_BOUNDS 2
#define LZ4FLEX_E_EXPECTED_ANOTHER_BYTE 3
#define LZ4FLEX_E_OFFSET_ZERO 4
#define LZ4FLEX_E_OFFSET_OUT_OF_BOUNDS 5
/* frame::Error, src/frame/mod.rs:35-72 */
#define LZ4FLEX_FE_COMPRESSION 16
#define LZ4FLEX_FE_DECOMPRESSION 17
#define LZ4FLEX_FE_IO 18
#define LZ4FLEX_FE_UNSUPPORTED_BLOCKSIZE 19
#define LZ4FLEX_FE_UNSUPPORTED_VERSION 20
#define LZ4FLEX_FE_WRONG_MAGIC 21
#define LZ4FLEX_FE_RESERVED_BITS 22
#define LZ4FLEX_FE_INVALID_BLOCK_INFO 23
#define LZ4FLEX_FE_BLOCK_TOO_BIG 24
#define LZ4FLEX_FE_HEADER_CHECKSUM 25
#define LZ4FLEX_FE_BLOCK_CHECKSUM 26
#define LZ4FLEX_FE_CONTENT_CHECKSUM 27
#define LZ4FLEX_FE_SKIPPABLE_FRAME 28
#define LZ4FLEX_FE_DICTIONARY_NOT_SUPPORTED 29
#define LZ4FLEX_FE_CONTENT_LENGTH 30
#define LZ4FLEX_FE_OUTPUT_FULL 31          /* one-shot helpers only: caller's flat buffer too small */
/* library/runtime errors (API misuse or device failure; the reference would panic) */
#define LZ4FLEX_E_INVALID_ARG 64
#define LZ4FLEX_E_NO_DEVICE 65
#define LZ4FLEX_E_HIP 66
#define LZ4FLEX_E_NOMEM 67
#define LZ4FLEX_E_UNSUPPORTED 68         /* entry point declared but its GPU path is not built yet */

typedef struct lz4flex_err_detail {
    uint64_t expected; /* OutputTooSmall{expected}, ContentLengthError{expected}, SkippableFrame(len), ... */
    uint64_t actual;   /* OutputTooSmall{actual},   ContentLengthError{actual} */
    int32_t inner;     /* frame: block error code wrapped by DecompressionError */
    int32_t hip_error; /* hipError_t when the code is LZ4FLEX_E_HIP */
} lz4flex_err_detail;

/* ---- context ------------------------------------------------------------------------------ */
/* Owns the device workspace (the throughput encoder's 164 MiB of candidate slots and segment bodies, allocated HERE so
 * that no compress call allocates; the staging arena and descriptor arrays of MEM_HOST calls) and a HIP stream.
 * One context per thread; distinct contexts are independent (reference: no global state, all fns reentrant).  MEM_DEVICE
 * compress batches of ONE context share its encoder workspace: the library orders them on the device (a batch enqueued on
 * another stream than the previous one waits for it), so they never overlap -- use one context per stream for concurrency.
 * The scalar calls below use a lazily created per-thread default context on the current HIP device. */
typedef struct lz4flex_ctx lz4flex_ctx;
int lz4flex_ctx_create(lz4flex_ctx **ctx, int device /* -1 = current */);
void lz4flex_ctx_destroy(lz4flex_ctx *ctx);
int lz4flex_device_count(void);
const char *lz4flex_version(void);
/* hash of the sources (csrc/ + include/ + compiler flags) this binary was built from; lz4_flex_amd/build.py
 * recomputes it from the tree, so a stale library is detectable */
const char *lz4flex_build_id(void);
/* The round of this header: 6.  Changes a caller built against an earlier header has to know: round 5 appended chain_prev / n_chains to
 * lz4flex_decompress_ext (read only for LZ4FLEX_MEM_DEVICE | LZ4FLEX_MEM_CHAINED batches; a struct of the four older members is fine for
 * every other call); round 6 removed "decompress_variant" 5 / 6 (the wave decoder; 13 took its place) and moved 12 to tools builds. */
int lz4flex_abi_version(void);
/* last HIP error string seen by this thread (diagnostics) */
const char *lz4flex_last_error(void);

/* ---- block: scalar, lz4_flex signatures ---------------------------------------------------- */
/* block::get_maximum_output_size, src/block/compress.rs:588-590 */
size_t lz4flex_get_maximum_output_size(size_t input_len);
/* WHAT A SCALAR CALL COSTS.  The lz4_flex-shaped scalar entries below are 1-block batches through the same kernels: one transfer
 * up (descriptors + input from page-locked staging), the kernels, one transfer down, one synchronisation.  Measured on an MI355X
 * (profiles/r06_scalar_latency.txt): a 1 KiB block 0.06 ms either way, a 64 KiB block 0.32 - 0.36 ms to compress and 0.15 ms to
 * decompress (one block occupies one of 256 CUs: the kernel, not PCIe, is the cost), 16 MiB 10 / 25 ms -- a CPU core does a 64 KiB
 * block in 0.04 / 0.012 ms.  These entries exist so that code written against lz4_flex links and runs; the throughput of this
 * library is in the BATCH entries (lz4flex_compress_batch / lz4flex_decompress_batch: thousands of blocks per launch, device-
 * resident buffers) and the frame encoder / decoder, which batch internally.  Batch or lose. */
/* block::compress_into, src/block/compress.rs:599-601.  Err(OutputTooSmall) up front iff
 * out_cap < get_maximum_output_size(in_len) (:338-340).
 * CONTRACT OF THE OUTPUT BYTES (this entry point, compress_prepend_size, compress_into_with_table, lz4flex_compress_batch and
 * the frame encoder): in the DEFAULT compress_mode (0, "fast") the block is a valid LZ4 block with this library's own parse --
 * lz4_flex's decoder (and any other LZ4 decoder) returns the input, the ratio is within a percent of lz4_flex's (usually
 * better) -- but the bytes are NOT lz4_flex's.  Callers that hash, deduplicate or golden-file compressed output must select
 * compress_mode 1 ("exact": lz4flex_set_tuning / LZ4FLEX_COMPRESS_MODE=exact), which reproduces the reference encoder byte
 * for byte as restated by oracle/ (no Rust toolchain exists here: "oracle-exact, reference byte-unpinned"). */
int64_t lz4flex_compress_into(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap);
/* block::compress_into_with_dict, src/block/compress.rs:610-616 */
int64_t lz4flex_compress_into_with_dict(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap,
                                        const uint8_t *dict, size_t dict_len);
/* block::compress_prepend_size, src/block/compress.rs:673-675 (LE u32 length prefix) */
int64_t lz4flex_compress_prepend_size(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap);
/* block::compress_prepend_size_with_dict, src/block/compress.rs:692-694 (dictionaries of <= 3 bytes are ignored, :626-628) */
int64_t lz4flex_compress_prepend_size_with_dict(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap,
                                                const uint8_t *dict, size_t dict_len);
/* block::CompressTable + compress_into_with_table, src/block/compress.rs:710-766.  The reference clears the table on every
 * call; the handle avoids re-allocating it and carries its variant: Small (u16 entries, 4-byte hash: what compress_into
 * uses below 65 535 bytes) or Large (u32 entries, 5-byte hash).  An input of >= 65 535 bytes upgrades a Small table to
 * Large for good (:752-754), which changes the bytes later small inputs compress to -- reproduced here.  The handle owns
 * the device workspace reused across calls. */
typedef struct lz4flex_compress_table lz4flex_compress_table;
lz4flex_compress_table *lz4flex_compress_table_new(int large /* 0 = CompressTable::small() / default, 1 = large() */);
void lz4flex_compress_table_free(lz4flex_compress_table *t);
int lz4flex_compress_table_is_large(const lz4flex_compress_table *t);
int64_t lz4flex_compress_into_with_table(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap,
                                         lz4flex_compress_table *table);
/* block::decompress_into, src/block/decompress.rs:454-456 */
int64_t lz4flex_decompress_into(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap,
                                lz4flex_err_detail *detail /* nullable */);
/* block::decompress_into_with_dict, src/block/decompress.rs:462-468 */
int64_t lz4flex_decompress_into_with_dict(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap,
                                          const uint8_t *dict, size_t dict_len, lz4flex_err_detail *detail);
/* block::uncompressed_size, src/block/mod.rs:151-157: returns the LE u32 prefix or -EXPECTED_ANOTHER_BYTE */
int64_t lz4flex_uncompressed_size(const uint8_t *in, size_t in_len);
/* block::decompress_size_prepended, src/block/decompress.rs:493-496; out_cap must be >= the prefix */
int64_t lz4flex_decompress_size_prepended(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap,
                                          lz4flex_err_detail *detail);

/* block::decompress_size_prepended_with_dict, src/block/decompress.rs:521-527 */
int64_t lz4flex_decompress_size_prepended_with_dict(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap,
                                                    const uint8_t *dict, size_t dict_len, lz4flex_err_detail *detail);

/* ---- block: batched (the hot entry; the frame layer's per-block calls
 *      src/frame/compress.rs:282-298 and src/frame/decompress.rs:288-305, batched) ------------ */
#define LZ4FLEX_MEM_HOST 0   /* every pointer is host memory; the call stages through the ctx arena and is synchronous */
#define LZ4FLEX_MEM_DEVICE 1 /* every pointer (data AND descriptor/result arrays) is device memory; asynchronous on `stream` */
/* OR into mem_kind for DEVICE batches that may hold blocks > 64 KiB: the lengths live in device memory where the host cannot see
 * them (HOST batches detect it themselves).  compress, reference-exact mode: selects the u32 hash table; decompress: a hint
 * that the blocks are large, so any number of them goes to the one-workgroup-per-block decoder (lz4_decompress_pcd.hip), which
 * otherwise serves batches of up to 1 024 blocks.  Results do not depend on the hint. */
#define LZ4FLEX_MEM_BIG_BLOCKS 0x100

/* OR into mem_kind for lz4flex_decompress_batch_ex with ext->out_pos: the batch is a CHAIN -- every block has the same out_off,
 * and block i's prefix [out_off, out_off + out_pos[i]) is what blocks 0 .. i-1 of this batch produce (plus whatever lay before
 * out_pos[0] when the call was made).  This is how a Linked frame's blocks are decoded in one launch
 * (src/frame/decompress.rs:195-222,280-306: the reference decodes them one after the other into one window): the token chains
 * of all blocks are parsed at once, a block waits for its predecessors only where a match really reaches behind its start.
 * At most 65 536 blocks per call; no external dictionaries. */
#define LZ4FLEX_MEM_CHAINED 0x200

/* per-block compress flags.  They select among the REFERENCE's hash tables and therefore only have a meaning in compress_mode
 * exact; the throughput encoder (compress_mode fast, the default) has one table layout of its own and ignores them, as it
 * ignores the Small / Large state of a lz4flex_compress_table. */
#define LZ4FLEX_BLOCK_DEFAULT 0u            /* block::compress_into: table/hash picked by length (compress.rs:559-566) */
/* bit 1: the FrameEncoder's table (HashTable4K + 5-byte hash whatever the length,
 * src/frame/compress.rs:76,141) with a freshly zeroed table: block 0 of a frame */
#define LZ4FLEX_BLOCK_FRAME_FIRST 2u
/* bits 1|0: block k>0 of an Independent frame: same table, every entry unreachable, position 0
 * is probed (src/frame/compress.rs:357-367, src/block/compress.rs:353-359,422-429) */
#define LZ4FLEX_BLOCK_FRAME_CONTINUATION 3u
/* bits 8..31, any compress_mode's flag word: h bytes of the SAME STREAM lie in front of the block, readable at
 * in_base[in_off[i] - h .. in_off[i]) -- block k > 0 of a Linked frame (src/frame/compress.rs:280-299,327-356: the reference
 * keeps the previous 64 KiB of input as the next block's dictionary).  The throughput encoder then lets the block's matches
 * reach into them: with h >= 32 768 every position sees between 32 and 64 KiB of the stream behind it (a smaller h is not
 * used), all blocks of the batch still encode side by side, and the block can only be decoded behind those bytes (a Linked
 * frame, LZ4FLEX_MEM_CHAINED, or lz4flex_decompress_batch_ex with them as out_pos prefix / dictionary).  compress_mode exact
 * ignores the bits: the reference's bytes for dependent blocks come from lz4flex_compress_chains. */
#define LZ4FLEX_BLOCK_HISTORY(h) ((uint32_t)(h) << 8)

/* Compress n independent blocks.  Block i reads in_base[in_off[i] .. +in_len[i]] and writes at
 * out_base[out_off[i] ..], capacity out_cap[i] (must be >= get_maximum_output_size(in_len[i]),
 * else status[i] = LZ4FLEX_E_OUTPUT_TOO_SMALL and nothing is written).  out_len[i] = bytes
 * written.  flags may be NULL.  hip_stream: the hipStream_t a MEM_DEVICE batch is enqueued on (NULL = HIP's
 * null stream); MEM_HOST batches ignore it and use the context's own stream.
 * Returns 0 or -code for call-level failures; per-block results are in status[]. */
int lz4flex_compress_batch(lz4flex_ctx *ctx, const void *in_base, const uint64_t *in_off, const uint32_t *in_len,
                           const uint32_t *flags, uint32_t n, void *out_base, const uint64_t *out_off,
                           const uint32_t *out_cap, uint32_t *out_len, int32_t *status, int mem_kind,
                           void *hip_stream);

/* Chains of DEPENDENT blocks: the full compress_internal signature (src/block/compress.rs:289-325) -- a
 * prefix before in_pos, an external dictionary, a stream offset and ONE hash table that persists across
 * the blocks of a chain (Linked frames, src/frame/compress.rs:280-299,327-356; compress_into_with_dict,
 * :554-583).  Chains run in parallel, the blocks of a chain in order.  All offsets index in_base. */
typedef struct lz4flex_chain_block {
    uint64_t in_off;    /* start of `input` (prefix included) */
    uint64_t dict_off;  /* start of ext_dict */
    uint32_t in_len;    /* input.len() */
    uint32_t in_pos;    /* input_pos: first byte to compress */
    uint32_t dict_len;
    uint32_t so;        /* input_stream_offset */
    uint32_t repos;     /* HashTable4K::reposition(repos) before this block (hashtable.rs:113-117); 0 = none */
    uint32_t flags;     /* bit0: 4-byte hash (the HashTable4KU16 case of compress.rs:559-562); bit1: clear the table and init_dict (:571-583) */
} lz4flex_chain_block;
/* blocks[chain_first[c] .. +chain_count[c]) form chain c; out_off/out_cap/out_len/status are per block.
 * tbl_state (nullable): 4096 u32 per chain, read before the chain's first block and written back after its
 * last one, so a chain can continue in a later call.  MEM_HOST or MEM_DEVICE as for the batches. */
int lz4flex_compress_chains(lz4flex_ctx *ctx, const void *in_base, const lz4flex_chain_block *blocks, uint32_t n_blocks,
                            const uint32_t *chain_first, const uint32_t *chain_count, uint32_t n_chains,
                            void *out_base, const uint64_t *out_off, const uint32_t *out_cap, uint32_t *out_len,
                            int32_t *status, uint32_t *tbl_state, int mem_kind, void *hip_stream);

/* Decompress n independent blocks.  out_cap[i] >= true size (larger allowed, as
 * decompress_into).  status[i] = 0 or a DecompressError code; detail (nullable, 2*n u64:
 * expected, actual) is filled for OutputTooSmall. */
int lz4flex_decompress_batch(lz4flex_ctx *ctx, const void *in_base, const uint64_t *in_off, const uint32_t *in_len,
                             uint32_t n, void *out_base, const uint64_t *out_off, const uint32_t *out_cap,
                             uint32_t *out_len, int32_t *status, uint64_t *detail, int mem_kind,
                             void *hip_stream);

/* Optional per-block extras for decoding: an external dictionary (block::decompress_into_with_dict,
 * src/block/decompress.rs:462-468) and/or an initial sink position: the output region
 * [out_off, out_off+out_pos) already holds earlier bytes that matches may reference (the prefix mode
 * of Linked frames, src/frame/decompress.rs:293-306); out_len counts only the new bytes. */
typedef struct lz4flex_decompress_ext {
    const void *dict_base;     /* nullable */
    const uint64_t *dict_off;
    const uint32_t *dict_len;
    const uint32_t *out_pos;   /* nullable */
    /* nullable; LZ4FLEX_MEM_DEVICE | LZ4FLEX_MEM_CHAINED batches only: the batch holds SEVERAL chains (N Linked frames decoded side
     * by side, lz4flex_frame_decompress_many).  Block i's predecessor in its chain is block chain_prev[i] -- an index BELOW i -- or
     * 0xFFFFFFFF for the first block of a chain; the blocks of one chain share an out_off, different chains have regions of their
     * own.  Order the blocks level by level (every chain's block k before any chain's block k + 1) and all chains advance together. */
    const uint32_t *chain_prev;
    uint32_t n_chains;         /* with chain_prev: how many chains the batch holds (a hint for the kernel geometry: few chains get a
                                * large workgroup per block, many chains small ones); 0 = unknown */
} lz4flex_decompress_ext;
int lz4flex_decompress_batch_ex(lz4flex_ctx *ctx, const void *in_base, const uint64_t *in_off, const uint32_t *in_len,
                                uint32_t n, void *out_base, const uint64_t *out_off, const uint32_t *out_cap,
                                uint32_t *out_len, int32_t *status, uint64_t *detail,
                                const lz4flex_decompress_ext *ext, int mem_kind, void *hip_stream);

/* Settings (ctx NULL = the default context the scalar / frame entry points use):
 * "compress_mode": 0 = throughput encoder (default; lz4_compress_wave.hip: a valid LZ4 block with this library's own
 *   parse -- any LZ4 decoder returns the input; ratio within a percent of the reference's, usually better), 1 = the
 *   reference's exact bytes (src/block/compress.rs:318-489 restated; about 3x slower).  Blocks with a dictionary / prefix
 *   always use the exact encoder.  A Linked frame (src/frame/compress.rs:261-371) written in mode 0 holds blocks whose matches
 *   reach up to 64 KiB back into the blocks before them (LZ4FLEX_BLOCK_HISTORY below: 32 KiB of the stream in front of every
 *   block are its history) -- a dependency-carrying Linked frame that any decoder returns to the input, with a ratio below the
 *   reference's and still one launch per batch of blocks; in mode 1 it holds the reference's bytes (one dependency chain,
 *   milliseconds per block).
 *   Environment: LZ4FLEX_COMPRESS_MODE=exact|fast.
 * "compress_sliding_window" (throughput encoder): how far the 64 KiB windows of a block LONGER than 64 KiB advance.  2 (default
 *   since round 5) = by 48 KiB, so every window start has 16 ... 64 KiB of the block behind it -- the reference's window slides
 *   continuously (src/block/compress.rs:403-405); 4 MiB log blocks: ratio 0.2940 (the reference: 0.2947) for a third more indexing;
 *   1 = by 32 KiB (round 4's default: 0.2929, every byte indexed twice); 0 = by 64 KiB (0.3027, fastest, round 3's bytes).
 *   Blocks of <= 64 KiB are one window either way.  Environment: LZ4FLEX_SLIDING_WINDOW=0|1|2.
 * "compress_subwindows" (throughput encoder): 0 (default) = by batch size -- a batch that leaves most of the encoder's persistent
 *   workgroups ("compress_workgroups", read-only: two per CU) without a block cuts every block of at most 64 KiB into 4 (n * 4 <=
 *   workgroups), 3 or 2 sub-windows that different workgroups encode side by side: a scalar compress_into and small batches take
 *   about half the time, at a ratio a few tenths of a percent higher; the BYTES of a block therefore depend on the size of the batch
 *   it travels in (always a valid block) and on the device's CU count; 1 = never, 2 / 3 / 4 = always.
 * "compress_deterministic": 1 = the bytes of a block are a function of the block and of the settings above alone -- never of the batch
 *   it travels in or of the device ("compress_subwindows" is taken as 1; small batches and the scalar compress_into then cost about twice
 *   the time).  What a caller sets that stores, hashes, deduplicates or golden-files compressed blocks (src/block/compress.rs:599-601
 *   is a pure function of its input; "compress_mode" 1 is the one that also gives the REFERENCE's bytes).  0 (default).
 * Kernel selection, for measurements only (every choice produces the same bytes / lengths / error variants):
 * "decompress_variant": 0 = by batch size (default), 7 = one block per WORKGROUP, token chain and copies parallel inside the
 *   block (lz4_decompress_pcd.hip: few, large blocks; 8 = the same with its small test geometry, 10 / 11 = with 256 / 512 lanes
 *   per block: what 0 picks for 513 ... 640 / 257 ... 512 blocks), 13 = one block per WAVEFRONT, one lane per sequence
 *   (lz4_decompress_seq.hip, round 6: what 0 picks for 641 ... 14 336 blocks; it replaced 5 / 6, the wave decoder and its
 *   two-wavefront form, which are gone), 4 = parser / copier split decoder (larger batches), 12 = the split decoder's parser feeding
 *   a piece cutter and the replay decoder's copy engine inside one workgroup (lz4_decompress_fused.hip: level with 4 on JSON, ahead on
 *   text, behind on incompressible data and runs; -DLZ4FLEX_TOOLS builds only since round 6), 9 = plan / replay
 *   (lz4_decompress_plan.hip + lz4_decompress_replay.hip; -DLZ4FLEX_TOOLS builds only since round 5: slower than 0 on every shape
 *   measured), 1 = decoder whose window lives in HBM/L2 (always used for dictionary / prefix blocks); "decompress_blocks_per_wg" (variant 4: 0 = 64, the default at every batch size; 8/16/32: the older narrow geometries, tests); "decompress_lanes"
 *   (16; 8/32/64 in -DLZ4FLEX_ALL_VARIANTS builds, variant 1); exact encoder: "compress_lanes" (8/16 lanes of a wavefront per
 *   block), "compress_variant" (1 = group encoder + emitter wavefront; 3 = group encoder alone, -DLZ4FLEX_ALL_VARIANTS builds);
 *   "decompress_second_pass" (tests: 0 leaves the blocks that variants 7 ... 13 hand to the reference-order kernel marked with
 *   status 0x7F000001 instead of decoding them again); "decompress_pcd_pair" (variants 7 / 8: 1 = batches of at most 128 LARGE
 *   blocks get a parser and a copier workgroup per block -- parse and copy of a block overlap, a single huge block uses two
 *   CUs' worth of time instead of one; 0 = never; 2 = every batch of at most 128 blocks: tests); "compress_carry_wait" (tests: 0 = a 64 KiB window of the throughput
 *   encoder that has to wait for the window before it -- few, large blocks: a block's windows run on different workgroups --
 *   gives up at once instead of after a fraction of a second; such a block is encoded again by the launch that follows, to
 *   the same bytes: a time-sliced GPU costs time, never an error); "decompress_level_chains" (default 1024; lz4flex_frame_decompress_many:
 *   a call that holds at least this many Linked streams decodes block k of every stream in ONE launch -- a plain batch whose prefixes
 *   the launches before it have written -- instead of a workgroup per block that polls its predecessor: thousands of short streams,
 *   4 096 x 256 KiB 6.7 -> 3.1 ms per GiB; 0 = never; tests set 1).
 * Keys that start with "debug_" inject faults for this library's own tests; they are unsupported and refused
 * (-LZ4FLEX_E_INVALID_ARG) unless the process runs with LZ4FLEX_TEST_HOOKS=1. */
int lz4flex_set_tuning(lz4flex_ctx *ctx, const char *key, int value);
/* the current value of a setting (>= 0), or -LZ4FLEX_E_INVALID_ARG for an unknown key.  Two read-only lists need no device and no
 * context: "dispatch_threshold_<i>" (the batch sizes at which the default decoder dispatch changes kernel or geometry, ascending) and
 * "decoder_config_<i>" (every decoder configuration this build can be pinned to: variant * 1000 + "decompress_lanes" (variant 1) /
 * "decompress_blocks_per_wg" (variant 4) / 0); both end where the key is refused */
int lz4flex_get_tuning(lz4flex_ctx *ctx, const char *key);

/* ---- frame (src/frame/) ------------------------------------------------------------------ */
typedef struct lz4flex_frame_info {   /* frame::FrameInfo, src/frame/header.rs:130-149 */
    int32_t has_content_size;
    uint64_t content_size;
    int32_t block_size;       /* frame::BlockSize: 0 Auto, 4 Max64KB, 5 Max256KB, 6 Max1MB, 7 Max4MB, 8 Max8MB */
    int32_t block_mode;       /* frame::BlockMode: 0 Independent, 1 Linked */
    int32_t block_checksums;
    int32_t content_checksum;
    int32_t legacy_frame;
} lz4flex_frame_info;

/* io::Write / io::Read stand-ins: return bytes written/read, or < 0 for an I/O error */
typedef int64_t (*lz4flex_write_fn)(void *user, const uint8_t *buf, size_t len);
typedef int64_t (*lz4flex_read_fn)(void *user, uint8_t *buf, size_t len);

typedef struct lz4flex_frame_encoder lz4flex_frame_encoder;
/* FrameEncoder::with_frame_info, src/frame/compress.rs:133-151 (info NULL => FrameEncoder::new) */
lz4flex_frame_encoder *lz4flex_frame_encoder_new(const lz4flex_frame_info *info, lz4flex_write_fn w, void *user);
/* io::Write::write, :375-396 */
int64_t lz4flex_frame_encoder_write(lz4flex_frame_encoder *e, const uint8_t *buf, size_t len);
/* io::Write::flush, :398-403 */
int lz4flex_frame_encoder_flush(lz4flex_frame_encoder *e);
/* try_finish, :173-187 */
int lz4flex_frame_encoder_try_finish(lz4flex_frame_encoder *e, lz4flex_err_detail *detail);
/* frame_info(), :159-161 */
void lz4flex_frame_encoder_frame_info(lz4flex_frame_encoder *e, lz4flex_frame_info *out);
/* how many uncompressed bytes the encoder gathers per kernel launch (default 64 MiB); before the first write */
int lz4flex_frame_encoder_set_batch_bytes(lz4flex_frame_encoder *e, size_t bytes);
void lz4flex_frame_encoder_free(lz4flex_frame_encoder *e);

typedef struct lz4flex_frame_decoder lz4flex_frame_decoder;
/* FrameDecoder::new, src/frame/decompress.rs:76-89 */
lz4flex_frame_decoder *lz4flex_frame_decoder_new(lz4flex_read_fn r, void *user);
/* io::Read::read, :353-367: bytes read, 0 at end of frame / EOF, < 0 = -code */
int64_t lz4flex_frame_decoder_read(lz4flex_frame_decoder *d, uint8_t *buf, size_t len, lz4flex_err_detail *detail);
/* io::BufRead::fill_buf, :410-416: *buf = the decoded bytes not consumed yet (valid until the next call on this decoder),
 * return value = their count; decodes the next batch of blocks when there are none; 0 at end of frame / EOF, < 0 = -code */
int64_t lz4flex_frame_decoder_fill_buf(lz4flex_frame_decoder *d, const uint8_t **buf, lz4flex_err_detail *detail);
/* io::BufRead::consume, :418-421; amt must not exceed what the last fill_buf returned (the reference asserts) */
int lz4flex_frame_decoder_consume(lz4flex_frame_decoder *d, size_t amt);
int lz4flex_frame_decoder_set_batch_bytes(lz4flex_frame_decoder *d, size_t bytes);
void lz4flex_frame_decoder_free(lz4flex_frame_decoder *d);

/* One-shot helpers over flat host buffers: FrameEncoder::with_frame_info + write_all + finish,
 * and FrameDecoder::new + read_to_end (first frame only; *consumed = input bytes read). */
int64_t lz4flex_frame_compress(const uint8_t *in, size_t in_len, const lz4flex_frame_info *info,
                               uint8_t *out, size_t out_cap, lz4flex_err_detail *detail);
int64_t lz4flex_frame_decompress(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap,
                                 size_t *consumed, lz4flex_err_detail *detail);
/* worst-case frame size for `in_len` bytes under `info` (header + per-block overhead + EndMark) */
size_t lz4flex_frame_compress_bound(size_t in_len, const lz4flex_frame_info *info);
/* frame::FrameInfo::write / read, src/frame/header.rs:232-373 */
int64_t lz4flex_frame_info_write(const lz4flex_frame_info *info, uint8_t *out, size_t out_cap);
int64_t lz4flex_frame_info_read(const uint8_t *in, size_t in_len, lz4flex_frame_info *info,
                                lz4flex_err_detail *detail);
/* XXH32 as used by the frame format (twox-hash in the reference); host */
uint32_t lz4flex_xxh32(const uint8_t *data, size_t len, uint32_t seed);
/* batched XXH32 of n DEVICE-resident buffers base[off[i] .. +len[i]) -> out[i] (device), on hip_stream: the block
 * checksums of src/frame/compress.rs:313-316 / src/frame/decompress.rs:178-187 for device-resident frames */
int lz4flex_xxh32_batch_device(const void *base, const uint64_t *off, const uint32_t *len, uint32_t n, uint32_t seed,
                               uint32_t *out, void *hip_stream);

/* ---- frame segments on the device (multi-GPU frame path: every rank assembles the segment of its block range) ---------
 * What FrameEncoder::write_block does around the codec call for each block (src/frame/compress.rs:282-316): the 4-byte
 * BlockInfo (frame/header.rs:108-124), the store-raw rule (compress.rs:301-306: a block that did not shrink is stored
 * uncompressed, high bit set), the payload and the optional XXH32 of the payload -- for n blocks already compressed by
 * lz4flex_compress_batch (MEM_DEVICE), back to back into `seg`.  Every pointer is device memory, the work is enqueued
 * on hip_stream.  seg_off: n + 1 u64, filled with each block's offset in seg and, last, the bytes written (read it
 * back to size the exchange).  seg must hold sum(in_len) + 8 n bytes.  scratch: 16 n bytes, needed with
 * block_checksums only. */
int lz4flex_frame_assemble_device(const void *src_base, const uint64_t *src_off, const uint32_t *in_len,
                                  const void *comp_base, const uint64_t *comp_off, const uint32_t *comp_len, uint32_t n,
                                  int block_checksums, void *seg, uint64_t *seg_off, void *scratch, void *hip_stream);
/* n independent byte ranges src_base[src_off[i] .. +len[i]) -> dst_base[dst_off[i] ..) in one launch (decode side: blocks
 * stored raw go straight to their place in the output, src/frame/decompress.rs:262-271) */
int lz4flex_copy_batch_device(const void *src_base, const uint64_t *src_off, const uint32_t *len, void *dst_base,
                              const uint64_t *dst_off, uint32_t n, void *hip_stream);

/* The block-header walk of FrameDecoder::read_block (src/frame/decompress.rs:231-247) for a frame in DEVICE memory: follows
 * the BlockInfo words from header_len to the EndMark and writes, per block, the payload offset and the length word (high
 * bit = stored uncompressed).  info (4 x u32, device): [0] blocks found, [1] 0 ok / 1 truncated frame / 2 BlockTooBig
 * (a block longer than block_size, :242-247) / 3 more than max_blocks, [2..3] offset behind the EndMark.  Only the few
 * result words travel to the host; the frame stays where it is. */
int lz4flex_frame_walk_device(const void *frame, uint64_t frame_len, uint32_t header_len, int block_checksums, uint32_t block_size,
                              uint32_t max_blocks, uint64_t *payload_off, uint32_t *len_word, uint32_t *info, void *hip_stream);

/* ---- many frames at once (BASELINE configs[4] in the shape that has parallelism in it: N streams, a frame each) ------------------
 * What N FrameEncoders / FrameDecoders do, one per stream (src/frame/compress.rs:261-371, src/frame/decompress.rs:189-342), as ONE
 * batch: the blocks of all streams are encoded by one launch; on the way back the BlockInfo words of all frames are walked on the
 * device (a thread per frame) and all blocks are decoded by one launch -- for BlockMode::Linked frames a chained batch of N chains
 * that advance together (src/frame/decompress.rs:195-222: a Linked frame is a dependency chain; one stream is serial, N streams are
 * N-fold parallel).  Stream i is in_base[in_off[i] .. + in_len[i]) and its result goes to out_base[out_off[i] .. + out_cap[i]);
 * out_len[i] = bytes written, status[i] = 0 or the negative code lz4flex_frame_compress / lz4flex_frame_decompress would have
 * returned for that stream alone (detail: nullable, n entries).  in_off / in_len / out_off / out_cap / out_len / status / detail
 * are HOST arrays; in_base / out_base are DEVICE memory (LZ4FLEX_MEM_DEVICE: work on hip_stream) or HOST memory (LZ4FLEX_MEM_HOST:
 * staged through the context's scratch; hip_stream ignored).  The calls BLOCK: they return after the work has completed (they synchronise
 * hip_stream), and they may allocate: the context's scratch grows with the largest job seen (hipMalloc / hipFree, a device-wide
 * synchronisation, whenever a larger one arrives).  The scratch lives on the CONTEXT's device whatever the calling thread's current one is.
 * compress_many: info NULL = FrameInfo::default(); BlockSize::Auto is resolved per stream from its length (frame/header.rs:57-67);
 * has_content_size: every frame's header carries ITS stream's length (info->content_size is not read).  compress_mode fast: the
 * frames hold this library's own parse (a Linked frame's blocks reach into the 32 KiB in front of them); exact: the reference's
 * bytes (Linked: lz4flex_compress_chains, a chain per stream).  out_cap[i] >= lz4flex_frame_compress_bound(in_len[i], info) always fits.
 * decompress_many: out_cap[i] should be the stream's size or little more (a frame's block table is sized from it).  The batch takes
 * frames whose blocks all fill the block size but the last; anything else (skippable frames, flush() boundaries,
 * checksum mismatches, corrupt blocks, buffers too small, legacy frames) is decoded by lz4flex_frame_decompress, stream by stream,
 * which also names the error -- same results, one stream's speed.  A stream's FIRST frame is decoded; bytes behind it are not read
 * (FrameDecoder::read returns 0 at the end of a frame, tests/tests.rs:633-647). */
int lz4flex_frame_compress_many(lz4flex_ctx *ctx, const void *in_base, const uint64_t *in_off, const uint64_t *in_len, uint32_t n,
                                const lz4flex_frame_info *info, void *out_base, const uint64_t *out_off, const uint64_t *out_cap,
                                uint64_t *out_len, int32_t *status, int mem_kind, void *hip_stream);
int lz4flex_frame_decompress_many(lz4flex_ctx *ctx, const void *in_base, const uint64_t *in_off, const uint64_t *in_len, uint32_t n,
                                  void *out_base, const uint64_t *out_off, const uint64_t *out_cap, uint64_t *out_len, int32_t *status,
                                  lz4flex_err_detail *detail, int mem_kind, void *hip_stream);

/* ---- the frame across the GPUs of one node (one process per GPU, an RCCL communicator; BASELINE configs[3]) ----------------
 * FrameEncoder / FrameDecoder for BlockMode::Independent frames whose blocks are spread over `world` ranks: the per-block
 * work of src/frame/compress.rs:261-371 / src/frame/decompress.rs:189-342 runs on every rank for its contiguous block
 * range, ONE exchange step moves the bytes (compress: ncclAllGather of the segment sizes + grouped ncclSend / ncclRecv of the
 * segments to `root`; decompress: the root walks the block headers on its device, broadcasts the table and sends every rank
 * its byte range).  nccl_comm: an ncclComm_t (NULL with world == 1: no RCCL call is made and RCCL is not even loaded).  All
 * data pointers are DEVICE memory; work is enqueued on hip_stream and the calls return after it has completed (they own
 * temporaries).  Linked frames, content checksums / sizes and BlockSize::Auto do not shard: -LZ4FLEX_E_UNSUPPORTED /
 * -LZ4FLEX_E_INVALID_ARG.  Verdicts that depend on the data or on one rank's buffers (a block that failed to compress, a frame
 * that does not parse, a root buffer too small) reach every rank before the exchange they would break: all ranks return the
 * same code and nobody is left waiting.  With more than one rank these entry points have run against a mock communicator
 * (ranks as threads on one device: tests/test_gpu_sharded_native.py; LZ4FLEX_RCCL_LIB names the library that provides the
 * ncclXxx entry points, RCCL by default), never against RCCL on several GPUs (no multi-GPU node was available to this build);
 * lz4_flex_amd/sharded.py is the same exchange over torch.distributed. */
/* bytes a rank's segment can take at most (and the root must be able to receive from it) */
uint64_t lz4flex_frame_segment_bound(uint64_t local_len, const lz4flex_frame_info *info);
/* local[0 .. local_len): this rank's blocks (a multiple of the block size except on the last rank); first_block: the global
 * index of its first block.  On `root`, frame receives header + all segments + EndMark and *frame_len their size (0 elsewhere). */
int lz4flex_frame_compress_sharded(lz4flex_ctx *ctx, void *nccl_comm, int rank, int world, int root, const void *local,
                                   uint64_t local_len, uint64_t first_block, const lz4flex_frame_info *info, void *frame,
                                   uint64_t frame_cap, uint64_t *frame_len, void *hip_stream);
/* frame / frame_bytes: needed on `root` only.  Every rank decodes blocks [*first_block, *first_block + *n_blocks) into out (block i of
 * the range at i * block size), *out_len bytes; info_out (nullable) receives the frame's block size and checksum flag. */
int lz4flex_frame_decompress_sharded(lz4flex_ctx *ctx, void *nccl_comm, int rank, int world, int root, const void *frame,
                                     uint64_t frame_bytes, void *out, uint64_t out_cap, uint64_t *out_len, uint64_t *first_block,
                                     uint64_t *n_blocks, lz4flex_frame_info *info_out, lz4flex_err_detail *detail, void *hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* LZ4FLEX_AMD_H */
