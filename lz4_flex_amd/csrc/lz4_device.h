// lz4_device.h -- kernel argument blocks and launch entry points shared by the HIP kernels
// and the host C ABI (capi.cpp).  Device-side status codes mirror include/lz4flex_amd.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define LZ4FLEX_DEV_E_OUTPUT_TOO_SMALL 1
#define LZ4FLEX_DEV_E_LITERAL_OUT_OF_BOUNDS 2
#define LZ4FLEX_DEV_E_EXPECTED_ANOTHER_BYTE 3
#define LZ4FLEX_DEV_E_OFFSET_ZERO 4
#define LZ4FLEX_DEV_E_OFFSET_OUT_OF_BOUNDS 5

struct lz4flex_ctx;

namespace lz4flex_dev {

// All pointers are device pointers.
struct DecompressArgs {
    const uint8_t* in_base;
    const uint64_t* in_off;
    const uint32_t* in_len;
    uint8_t* out_base;
    const uint64_t* out_off;
    const uint32_t* out_cap;
    const uint32_t* out_pos;   // nullable: initial sink position (prefix mode of Linked frames)
    const uint8_t* dict_base;  // nullable: external dictionaries
    const uint64_t* dict_off;
    const uint32_t* dict_len;
    uint32_t* out_len;
    int32_t* status;
    uint64_t* detail;          // nullable: 2 per block (expected, actual)
    uint32_t n;
    int32_t only_status;       // lz4_decompress_blocks_kernel: 0 = every block, else only blocks whose status equals it (second pass)
    // lz4_decompress_pcd_kernel with out_pos: nullable; n words, zero before the launch.  Set => the batch is a CHAIN: the blocks share
    // one output region and block i's prefix [.., out_pos[i]) is what blocks 0..i-1 of this batch write (Linked frames,
    // src/frame/decompress.rs:195-222,280-306).  chain_done[i] becomes 1 (done, and every block before it) or 2 (given up).
    uint32_t* chain_done;
    // with chain_done: nullable; n words.  Set => the batch holds SEVERAL chains (N Linked frames side by side): block i's predecessor
    // is block chain_prev[i] (< i), 0xFFFFFFFF = the first block of its chain (nothing of the batch lies before it).  Null: i - 1.
    const uint32_t* chain_prev;
    uint32_t n_chains;         // with chain_prev: the number of chains (0 = unknown); picks the workgroup size (capi.cpp)
    // lz4_decompress_pcd_kernel: nullable.  Set (n <= PCD_PAIR_MAX_BLOCKS, the first 64 n bytes zero before the launch) => every block
    // gets TWO workgroups: one parses its tiles and hands the token lists over through this workspace, the other copies
    // (lz4_decompress_pcd.hip "roles"); decompress_pcd_pair_ws_bytes() bytes
    uint8_t* pair_ws;
    // lz4_decompress_pcd_kernel, tests only: block debug_giveup - 1 of a chained batch gives up as if a wait had timed out (0: none)
    uint32_t debug_giveup;
};
constexpr uint32_t PCD_PAIR_MAX_BLOCKS = 128u;
// The batch sizes at which launch_decompress_fast (capi.cpp) changes decoder and launch_decompress_split its geometry: ONE
// table, read by the dispatch AND (through lz4flex_get_tuning "dispatch_threshold_<i>") by the tests, whose decoder matrix is
// every threshold and its successor -- a threshold edit cannot leave a size class untested (a wrong result lived a round in
// batches of 5 121 ... 16 383 blocks because one geometry was never run on real data).
constexpr uint32_t DISPATCH_PCD_1024 = 256u;         // <= : a workgroup of 1 024 lanes per block (one per CU); also for large blocks and chains at any count
constexpr uint32_t DISPATCH_PCD_512 = 512u;          // <= : 512 lanes per block (two per CU)
constexpr uint32_t DISPATCH_PCD_256 = 640u;          // <= : 256 lanes per block (four per CU)
// round 6: above, a wavefront per block and a lane per sequence (lz4_decompress_seq.hip) -- its time grows with the batch (16 wavefronts per
// CU: 4 096 blocks are one round), the split decoder's is one block's chain whatever the batch: JSON tiles 768 / 4 096 / 8 192 / 12 288 /
// 14 336 / 16 384 blocks 0.32 / 0.53 / 1.00 / 1.44 / 1.63 / 1.82 ms against 0.35 (256 lanes per block) / 1.01 / 1.63 / 1.64 / 1.65 / 1.67;
// text and log tiles are ahead at every size (16 384 blocks: 2.76 against 3.53, 1.57 against 2.25 ms) -- the threshold is the JSON one
// (profiles/r06_decoder_shapes.txt)
constexpr uint32_t DISPATCH_SEQ_MAX = 14336u;        // <= : a wavefront per block, a lane per sequence; above: the split decoder
constexpr uint32_t DISPATCH_SPLIT_FULL = 64u * 256u;  // (not a change of kernel: from here on every CU holds a workgroup of the split decoder; the tests want this size too)
size_t decompress_pcd_pair_ws_bytes();

struct CompressArgs {
    const uint8_t* in_base;
    const uint64_t* in_off;
    const uint32_t* in_len;
    const uint32_t* flags;     // nullable
    uint8_t* out_base;
    const uint64_t* out_off;
    const uint32_t* out_cap;
    uint32_t* out_len;
    int32_t* status;
    uint32_t n;
    uint32_t slide;            // throughput encoder: 0, or the bytes the windows of a block longer than 64 KiB advance by (32 768 or 49 152; lz4_compress_wave.hip Item)
    uint32_t sub;              // throughput encoder: 2 / 4 = blocks of at most 64 KiB are cut into that many sub-windows (small batches: lz4_compress_wave.hip Item::sub); else one window
};

// plan / replay decoder (lz4_decompress_plan.hip, lz4_decompress_replay.hip; record format: lz4_plan_common.h)
namespace plan { struct BlockPlan; }
struct ReplayArgs {
    const uint8_t* in_base;
    uint8_t* out_base;
    const plan::BlockPlan* plans;   // n per-block headers
    const uint32_t* words;          // the plan array
    uint32_t n;
    uint32_t max_turns;             // no block's plan is longer than this many turns (a plan without its K_END must not hang the kernel)
};
hipError_t launch_replay(const ReplayArgs& a, hipStream_t s);
struct PlanArgs {
    const uint8_t* in_base;
    const uint64_t* in_off;
    const uint32_t* in_len;
    const uint64_t* out_off;
    const uint32_t* out_cap;
    plan::BlockPlan* plans;         // n per-block headers (written)
    uint32_t* words;                // the plan array: slot_words per block (written)
    uint32_t* out_len;
    int32_t* status;                // 0, or redo_code: the block has no plan (irregular) and is left to the reference-order kernel
    uint32_t n;
    uint32_t slot_words;
    int32_t redo_code;
};
size_t plan_slot_words();
hipError_t launch_plan(const PlanArgs& a, hipStream_t s);

hipError_t launch_decompress(const DecompressArgs& a, int lanes_per_block, hipStream_t s);
// the second pass of a CHAINED batch: the blocks whose status equals a.only_status, one after the other in chain order (one wavefront)
hipError_t launch_decompress_chain_redo(const DecompressArgs& a, hipStream_t s);
// one block per wavefront, one LANE PER SEQUENCE (lz4_decompress_seq.hip, round 6): speculative part walks give the token positions,
// 64 sequences at a time are placed by a prefix sum and copied by their lanes; irregular blocks are left with status redo_code for a
// second pass of launch_decompress (only_status = redo_code), which decodes them in the reference's check order
hipError_t launch_decompress_seq(const DecompressArgs& a, int32_t redo_code, hipStream_t s);
hipError_t launch_decompress_split(const DecompressArgs& a, hipStream_t s, int blocks_per_wg = 0);   // parser / copier wavefronts, no dict/prefix
// parser -> emitter -> quad wavefronts (lz4_decompress_fused.hip: the split decoder's parser, the replay decoder's copy engine, no dict/prefix);
// blocks of 512 KiB or more are left with status redo_code for a second pass of launch_decompress.  -DLZ4FLEX_TOOLS builds only (round 6)
hipError_t launch_decompress_fused(const DecompressArgs& a, int32_t redo_code, hipStream_t s);
// one WORKGROUP per block, token chain and copies parallel inside the block (lz4_decompress_pcd.hip: few, large blocks); irregular
// blocks are left with status redo_code like behind launch_decompress_seq.  test_geometry: tiny tiles / batches (tests only)
hipError_t launch_decompress_pcd(const DecompressArgs& a, int32_t redo_code, hipStream_t s, int geometry = 0);   // 0 production (1 024 lanes), 1 tests, 2 / 3 medium batches (256 / 512 lanes)
hipError_t launch_compress(const CompressArgs& a, int variant, hipStream_t s);
// throughput ("wave") encoder, lz4_compress_wave.hip: persistent workgroups, `workspace` holds
// compress_wave_workspace_bytes(n_workgroups) bytes (cand[] slots + segment bodies, L2 / Infinity Cache resident)
size_t compress_wave_workspace_bytes(int n_workgroups);
hipError_t launch_compress_wave(const CompressArgs& a, void* workspace, int n_workgroups, hipStream_t s,
                                unsigned long long* prof = nullptr,    // prof: 8 cycle counters (tools), nullable
                                bool carry_wait = true);               // tests: false = a window that has to wait for its predecessor gives up at once

// chains of dependent blocks (dictionary / Linked frames); `blocks` is an array of the 40-byte ChainBlock
// records laid out as {u64 in_off, u64 dict_off, u32 in_len, in_pos, dict_len, so, repos, flags}
hipError_t launch_compress_chain(const uint8_t* in_base, const void* blocks, const uint32_t* chain_first,
                                 const uint32_t* chain_count, uint32_t n_chains, uint8_t* out_base,
                                 const uint64_t* out_off, const uint32_t* out_cap, uint32_t* out_len, int32_t* status,
                                 uint32_t* tbl_state, hipStream_t s);

hipError_t launch_xxh32_batch(const uint8_t* base, const uint64_t* off, const uint32_t* len, uint32_t n, uint32_t seed,
                              uint32_t* out, hipStream_t s);

// frame_kernels.hip: a rank's block range -> [header | payload | (checksum)]* on the device; seg_off holds n + 1 offsets
// (the last one = bytes written); pay_off / pay_len / sums are n-element scratch arrays, needed with block_checksums only
hipError_t launch_frame_assemble(const uint8_t* src_base, const uint64_t* src_off, const uint32_t* in_len, const uint8_t* comp_base,
                                 const uint64_t* comp_off, const uint32_t* comp_len, uint32_t n, int block_checksums, uint8_t* seg,
                                 uint64_t* seg_off, uint64_t* pay_off, uint32_t* pay_len, uint32_t* sums, hipStream_t s);
hipError_t launch_frame_walk(const uint8_t* f, uint64_t n, uint32_t hdr, uint32_t tail, uint32_t block_size, uint32_t max_blocks, uint64_t* off,
                             uint32_t* len, uint32_t* info, hipStream_t s);
// many frames at once (frame_many.cpp).  ManyStream: one stream to encode -- its blocks are [first, first + count) of the block arrays,
// its frame goes to out_base[out_off .. + out_cap); hdr = FrameInfo::write's bytes; flags bit 0 block checksums, bit 1 content checksum.
struct ManyStream { uint64_t out_off, out_cap; uint32_t first, count, hdr_len, flags; uint8_t hdr[24]; };
// ManyFrame: one frame to decode -- base[off .. + len), header of hdr_len bytes, table slots [slot, slot + slot_cap); skip: not walked
struct ManyFrame { uint64_t off, len; uint32_t hdr_len, block_size, flags, slot, slot_cap, skip; };
// verdict[s]: 0 written, 1 a block failed to compress, 2 the frame does not fit; frame_len[s] = bytes written.  dst_off (n_blocks) is
// scratch, pay_off / pay_len / sums (n_blocks) with block_checksums only, content_sum (n_streams) with content checksums only.
hipError_t launch_frame_many_assemble(const ManyStream* st, uint32_t n_streams, const uint8_t* src_base, const uint64_t* src_off, const uint32_t* in_len,
                                      const uint8_t* comp_base, const uint64_t* comp_off, const uint32_t* comp_len, const int32_t* comp_st, uint32_t n_blocks,
                                      int block_checksums, const uint32_t* content_sum, uint8_t* out_base, uint64_t* dst_off, uint64_t* pay_off,
                                      uint32_t* pay_len, uint32_t* sums, uint64_t* frame_len, int32_t* verdict, hipStream_t s);
hipError_t launch_frame_many_heads(const uint8_t* base, const uint64_t* off, const uint64_t* len, uint32_t n, uint8_t* heads, hipStream_t s);
hipError_t launch_frame_many_walk(const uint8_t* base, const ManyFrame* fr, uint32_t n, uint64_t* pay_off, uint32_t* word, uint32_t* info, hipStream_t s);
hipError_t launch_frame_sums_check(const uint8_t* base, const uint64_t* pay_off, const uint32_t* pay_len, const uint32_t* sums, uint32_t n, uint32_t* bad,
                                   hipStream_t s);
hipError_t launch_copy_batch(const uint8_t* src_base, const uint64_t* src_off, const uint32_t* len, uint8_t* dst_base, const uint64_t* dst_off,
                             uint32_t n, hipStream_t s);


// capi.cpp, for frame_many.cpp: *ctx = the calling thread's default context if null; the context's device / own stream / compress_mode;
// grow-only device scratch in 4 slots (valid until the next ctx_scratch of the same slot; the caller runs to completion before it returns)
int ctx_resolve(struct ::lz4flex_ctx** ctx);
int ctx_device(struct ::lz4flex_ctx* c);
hipStream_t ctx_stream(struct ::lz4flex_ctx* c);
int ctx_comp_mode(struct ::lz4flex_ctx* c);
int ctx_scratch(struct ::lz4flex_ctx* c, int slot, size_t bytes, void** out);

}  // namespace lz4flex_dev
