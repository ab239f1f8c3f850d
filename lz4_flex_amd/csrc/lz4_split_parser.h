// lz4_split_parser.h -- the PARSER half of the split decoder (lz4_decompress_split.hip): constants, the LDS record
// queue and the per-lane token-chain walker.  The code is per-lane scalar code (its only wave-level operation is
// __any), so the same source also compiles for the host with -DLZ4FLEX_HOST_SIM: tests/sim/ runs it against the
// oracle on the CPU (test infrastructure; the product path is the HIP kernel).
#pragma once
#include <stdint.h>

#ifdef LZ4FLEX_HOST_SIM
// status codes of lz4_device.h (which needs the HIP runtime headers)
#define LZ4FLEX_DEV_E_OUTPUT_TOO_SMALL 1
#define LZ4FLEX_DEV_E_LITERAL_OUT_OF_BOUNDS 2
#define LZ4FLEX_DEV_E_EXPECTED_ANOTHER_BYTE 3
#define LZ4FLEX_DEV_E_OFFSET_ZERO 4
#define LZ4FLEX_DEV_E_OFFSET_OUT_OF_BOUNDS 5
#define LZ4_LDS
#define LZ4_FN inline
#define LZ4_COLD_FN inline
// one lane per run on the host: "some other lane of the wavefront wants this path" is simulated by a coin (0 = never)
static uint32_t lz4_sim_chaos_state = 0u;
static inline bool lz4_sim_chaos() {
    if (lz4_sim_chaos_state == 0u) return false;
    lz4_sim_chaos_state = lz4_sim_chaos_state * 1664525u + 1013904223u;
    return (lz4_sim_chaos_state >> 29) == 0u;
}
#define LZ4_ANY(x) ((x) || lz4_sim_chaos())
static inline uint32_t lz4_alignbyte(uint32_t hi, uint32_t lo, uint32_t sh) {
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> (8u * (sh & 3u)));
}
#else
#include "lz4_device.h"
#define LZ4_LDS __attribute__((address_space(3)))
#define LZ4_FN __device__ __forceinline__
#define LZ4_COLD_FN __device__
#define LZ4_ANY(x) __any(x)
#define lz4_alignbyte(hi, lo, sh) __builtin_amdgcn_alignbyte(hi, lo, sh)
#endif
// base + off for a base that is aligned to more than off can reach (layouts with RING_ALIGNED: every block's ring is 64 B aligned):
// an OR on the device (one v_and_or with the masking).  Used inside ParserT<L> only.
#ifdef LZ4FLEX_HOST_SIM
#define LZ4_ALIGNED_PLUS(base, off) ((base) + (off))
#else
#define LZ4_ALIGNED_PLUS(base, off) (L::RING_ALIGNED ? (lds_u8*)(uintptr_t)((uint32_t)(uintptr_t)(base) | (uint32_t)(off)) : (lds_u8*)((base) + (off)))
#endif

namespace lz4flex_dev {
namespace v5 {

typedef uint32_t __attribute__((ext_vector_type(4))) u32x4;
typedef uint8_t LZ4_LDS lds_u8;
typedef volatile uint32_t LZ4_LDS lds_vu32;
typedef volatile u32x4 LZ4_LDS lds_vu128;
typedef uint32_t LZ4_LDS lds_u32;

constexpr uint32_t OUT_SLACK = 32;
constexpr uint32_t TAILB = 48;        // bytes of the block's end staged in LDS
constexpr uint32_t TAIL_BUF = 80;     // + zero padding: a 24-byte window read at any tail position stays inside
constexpr uint32_t RING = 64;         // compressed bytes around the parse position, per block (+ 8 mirrored bytes behind it)
constexpr uint32_t RING_BYTES = 80;
// LDS layout of one block: output buffer | record queue | head, tail | tail copy | sink | ring (64 B aligned) | PAD
// PAD: one lane per block (parser) and four lanes per block (copiers) address LDS with the block's size as their stride.  LDS has 32
// banks of 4 bytes: with a stride of 2 496 bytes (624 dwords, 16 mod 32) the parser's 64 lanes hit TWO banks -- every one of its
// accesses is executed 32 lanes per bank, one after the other -- and the copiers' 16-byte accesses two of eight 16-byte bank groups.
// 16 bytes of padding (628 dwords: a multiple of 4 that is 4 x odd) spread both over all eight groups; the rings are then only 16-byte
// aligned (RING_ALIGNED false: the parser adds where it could OR).
template <uint32_t OUT_CAP_, uint32_t OUT_H_, uint32_t QD_, uint32_t PAD_ = 0u>
struct Layout {
    static constexpr bool RING_ALIGNED = PAD_ % 64u == 0u;
    static constexpr uint32_t QD = QD_;             // records per queue (power of two)
    static constexpr uint32_t OUT_H = OUT_H_;       // history kept in LDS after a write-back
    static constexpr uint32_t OUT_CAP = OUT_CAP_;
    static constexpr uint32_t FLUSH_AT = (OUT_CAP - OUT_SLACK - OUT_H) / 2u - 8u;   // service: write back below this much space
    static constexpr uint32_t Q_OFF = OUT_CAP;
    static constexpr uint32_t CTL_OFF = Q_OFF + 16u * QD;   // head, tail
    static constexpr uint32_t TAIL_OFF = CTL_OFF + 16u;
    static constexpr uint32_t SINK_OFF = TAIL_OFF + TAIL_BUF;   // 16 bytes nobody reads: target of the parser's record store when it has nothing to push
    static constexpr uint32_t RING_OFF = (SINK_OFF + 16u + 63u) & ~63u;
    static constexpr uint32_t BLK_LDS = ((RING_OFF + RING_BYTES + 63u) & ~63u) + PAD_;   // without PAD a multiple of 64: every block's ring is 64 B aligned
    static_assert(BLK_LDS % 16 == 0 && RING_OFF % 64 == 0 && OUT_H % 16 == 0 && (QD & (QD - 1u)) == 0 && QD >= 8, "layout");
};
#ifndef LZ4S_LDS_PAD
#define LZ4S_LDS_PAD 16
#endif
using LayoutBig = Layout<1984, 512, 16, LZ4S_LDS_PAD>;     // 2 496 + 16 B per block: 64 blocks = one workgroup per CU (157 KiB of its 160 KiB of LDS)
static_assert(LayoutBig::BLK_LDS == 2496 + LZ4S_LDS_PAD && LayoutBig::FLUSH_AT == 712 && 64u * LayoutBig::BLK_LDS <= 163840u, "LDS per block");
using LayoutSmall = Layout<1024, 256, 8>;    // 1 408 B per block (host simulation of a short queue; not launched)
static_assert(LayoutSmall::BLK_LDS == 1408, "LDS per block");
// the default layout's constants at namespace level (host simulation, tools)
constexpr uint32_t QD = LayoutBig::QD, OUT_H = LayoutBig::OUT_H, OUT_CAP = LayoutBig::OUT_CAP, FLUSH_AT = LayoutBig::FLUSH_AT;
constexpr uint32_t Q_OFF = LayoutBig::Q_OFF, CTL_OFF = LayoutBig::CTL_OFF, TAIL_OFF = LayoutBig::TAIL_OFF, SINK_OFF = LayoutBig::SINK_OFF, RING_OFF = LayoutBig::RING_OFF;
constexpr uint32_t BLK_LDS = LayoutBig::BLK_LDS;
constexpr uint32_t PF_AHEAD = 512;    // compressed bytes kept warm ahead of the records being copied

// record.w = offset | kind << 16; a non-zero kind takes the copier group out of its steady loop:
constexpr uint32_t R_RARE = 2u;       // match whose offset is below the copier's bytes per lane (periodic copy)
constexpr uint32_t R_CAREFUL = 3u;    // literals whose source may end at the block's last byte (no wild reads)
constexpr uint32_t R_FINISH = 4u;     // same, and the block ends after them (or with an error)

#ifdef LZ4FLEX_HOST_SIM
static uint8_t g_pad[64] __attribute__((aligned(16)));
#else
__device__ __attribute__((aligned(16))) uint8_t g_pad[64];
#endif   // always-readable target of loads that fetch nothing

LZ4_FN uint32_t ld32(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
LZ4_FN uint32_t ld32l(const lds_u8* p) { uint32_t v; __builtin_memcpy(&v, (const void*)p, 4); return v; }

template <class L>
struct QueueT {
    static constexpr uint32_t QD = L::QD, Q_OFF = L::Q_OFF, CTL_OFF = L::CTL_OFF;
    lds_u8* blk;   // the block's LDS area
    LZ4_FN uint32_t head() const { return *reinterpret_cast<lds_vu32*>(blk + CTL_OFF); }
    LZ4_FN void set_head(uint32_t v) const { *reinterpret_cast<lds_vu32*>(blk + CTL_OFF) = v; }
    LZ4_FN uint32_t tail() const { return *reinterpret_cast<lds_vu32*>(blk + CTL_OFF + 4u); }
    LZ4_FN void set_tail(uint32_t v) const { *reinterpret_cast<lds_vu32*>(blk + CTL_OFF + 4u) = v; }
    LZ4_FN void put(uint32_t slot, uint32_t a, uint32_t b, uint32_t c, uint32_t d) const {
        const u32x4 v = {a, b, c, d};
        *reinterpret_cast<lds_vu128*>(blk + Q_OFF + 16u * (slot & (QD - 1u))) = v;
    }
    LZ4_FN u32x4 get(uint32_t slot) const {
        return *reinterpret_cast<lds_vu128*>(blk + Q_OFF + 16u * (slot & (QD - 1u)));
    }
};

// =====================================================================================================
// PARSER: one lane = one block
// =====================================================================================================
template <class L>
struct ParserT {
    static constexpr uint32_t QD = L::QD, Q_OFF = L::Q_OFF, CTL_OFF = L::CTL_OFF, TAIL_OFF = L::TAIL_OFF, SINK_OFF = L::SINK_OFF, RING_OFF = L::RING_OFF;
    const uint8_t* gin;      // compressed block
    const uint8_t* gal;      // gin rounded down to 4 bytes: the ring holds bytes of this "aligned space"
    uint32_t A;              // gin - gal
    uint32_t ilen, cap;
    uint32_t ip, op;
    uint32_t tok_over;       // 0, or 0x100 | low nibble of the token whose literals were pushed separately: the next
                             // step parses a synthetic token (no literals, that match nibble) at ip, the byte before the offset
    uint32_t done;
    int32_t status;
    uint64_t expected;
    uint32_t qtail;
    uint32_t tstart;         // the LDS tail copy holds compressed positions [tstart, ilen)
    QueueT<L> q;
    // The compressed bytes around the parse position live in a 64-byte LDS ring per block: bytes [vhi - 64, vhi) of the
    // aligned space, byte x at ring offset x & 63 (the first 8 bytes are mirrored behind the ring, so a dword pair never
    // wraps).  N = the 16-byte chunk at vhi, in flight from the previous step.  Round 1 kept a 48-byte window in
    // registers: sliding it and selecting 24 bytes at a per-lane offset cost ~85 of a step's 172 instructions; two aligned
    // LDS reads at computed addresses cost 10.
    uint32_t vhi;
    uint32_t rare_below;     // matches with a smaller offset are marked R_RARE (the copier's periodic path)
    uint32_t lit_slack;      // >= 3: a copier lane reads up to this many bytes behind the end of a plain record's literals (its
                             // word width - 1); literals closer to the block's end go through the exact path (R_CAREFUL)
    uint32_t climit;         // start of the last chunk that lies entirely inside the block (loads are clamped to it)
    u32x4 N;
#ifdef LZ4FLEX_SPLIT_DEBUG
    uint32_t dbg_ip, dbg_w0, dbg_w1, dbg_kb;   // state at the first sequence that left the fast path
#endif
#ifdef LZ4FLEX_PROFILE_PHASES
    uint32_t pr_steps, pr_noroom, pr_bubble, pr_slow, pr_pushed;
#endif

    static LZ4_FN uint32_t bfi(uint32_t m, uint32_t a, uint32_t b) { return (a & m) | (b & ~m); }
    static LZ4_FN int32_t imin3(int32_t a, int32_t b, int32_t c) { const int32_t m = a < b ? a : b; return m < c ? m : c; }
    static LZ4_FN uint32_t sel4(uint32_t m0, uint32_t m1, uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3) {
        return bfi(m1, bfi(m0, a3, a2), bfi(m0, a1, a0));
    }
    // Window geometry for a block at `g` of `n` bytes: chunks are fetched only if they lie entirely inside the block (a
    // clamped load returns the last such chunk; by then the lane reads the LDS tail copy instead); a block without
    // any full chunk points the window at g_pad.
    LZ4_FN void init_window(const uint8_t* g, uint32_t n) {
        gin = g;
        ilen = n;
        A = (uint32_t)(reinterpret_cast<uintptr_t>(g) & 3u);
        const bool any = n + A >= 16u;
        gal = any ? g - A : g_pad;
        climit = any ? ((n + A) & ~15u) - 16u : 0u;
        tstart = n > TAILB ? n - TAILB : 0u;
        vhi = 0u;
    }
    LZ4_FN lds_u8* ring() const { return q.blk + RING_OFF; }
    LZ4_FN void ring_put(const u32x4& c) {                  // the chunk at vhi enters the ring
        const uint32_t slot = vhi & (RING - 1u);
        *reinterpret_cast<lds_vu128*>(LZ4_ALIGNED_PLUS(ring(), slot)) = c;
        if (slot == 0u) {
            *reinterpret_cast<lds_vu32*>(ring() + RING) = c.x;
            *reinterpret_cast<lds_vu32*>(ring() + RING + 4u) = c.y;
        }
        vhi += 16u;
    }
    // fill the ring with the block's first 64 bytes, request the chunk behind them.  Only lanes that own a block: with fewer
    // than 64 blocks per workgroup the spare lanes point at block 0's LDS area, and their (zero) chunks landed in ITS ring --
    // which dword of which lane survives such a write is the hardware's business; a sequence in block 0's first 64 bytes
    // whose length byte was zeroed that way decoded 3 bytes short with status 0 (round 3: block 2 096 of the JSON batch with
    // 8 / 16 blocks per workgroup).
    LZ4_FN void prime() {
        for (uint32_t c = 0u; c < RING; c += 16u) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(chunk_addr(c));
            if (done == 0u) ring_put(v); else vhi += 16u;
        }
        N = *reinterpret_cast<const u32x4*>(chunk_addr(vhi));
    }
    LZ4_FN const uint8_t* chunk_addr(uint32_t c) const { return gal + (c < climit ? c : climit); }
    LZ4_FN uint32_t rd8(uint32_t pos) const {   // pos < ilen
        return pos >= tstart ? (uint32_t)q.blk[TAIL_OFF + (pos - tstart)] : (uint32_t)gin[pos];
    }
    // branch-free push: the record goes to the queue slot or to the sink, the tail is rewritten either way
    LZ4_FN void push_if(bool c, uint32_t lsrc, uint32_t ln, uint32_t ml, uint32_t off_flags) {
        const u32x4 v = {lsrc, ln, ml, off_flags};
        const uint32_t at = c ? Q_OFF + 16u * (qtail & (QD - 1u)) : SINK_OFF;
        *reinterpret_cast<lds_vu128*>(q.blk + at) = v;
        qtail += c ? 1u : 0u;
        *reinterpret_cast<lds_vu32*>(q.blk + (c ? CTL_OFF + 4u : SINK_OFF)) = qtail;   // lanes without a block share block 0's area: they must not touch its tail
    }
    LZ4_FN void push(uint32_t lsrc, uint32_t ln, uint32_t ml, uint32_t off_flags) {
        q.put(qtail, lsrc, ln, ml, off_flags);
        qtail += 1u;
        q.set_tail(qtail);
    }
    LZ4_FN void fail(int32_t code) {
        status = code;
        done = 1u;
        push(0u, 0u, 0u, R_FINISH << 16);
    }

    // Exact handling of one sequence (or of the offset half when need_off is set), byte by byte, every check in the
    // reference's order (src/block/decompress.rs:244-444).  Needs three free queue slots.
    LZ4_COLD_FN void exact_step() {
        uint32_t mlc;
        if (tok_over == 0u) {
            const uint32_t tok = rd8(ip);
            ip += 1u;
            uint32_t lit = tok >> 4;
            mlc = tok & 15u;
            if (lit == 15u) {
                uint64_t acc = lit;   // usize in the reference: > 16 MiB of 0xFF length bytes must not wrap a 32-bit sum
                for (;;) {   // read_integer_ptr :126-157
                    if (ip >= ilen) return fail(LZ4FLEX_DEV_E_EXPECTED_ANOTHER_BYTE);
                    const uint32_t e = rd8(ip);
                    ip += 1u;
                    acc += e;
                    if (e != 0xFFu) break;
                }
                if (acc > (uint64_t)(ilen - ip)) return fail(LZ4FLEX_DEV_E_LITERAL_OUT_OF_BOUNDS);
                lit = (uint32_t)acc;
            }
            if (lit > ilen - ip) return fail(LZ4FLEX_DEV_E_LITERAL_OUT_OF_BOUNDS);
            if (lit > cap - op) { expected = (uint64_t)op + lit; return fail(LZ4FLEX_DEV_E_OUTPUT_TOO_SMALL); }
            const uint32_t lsrc = ip;
            ip += lit;
            op += lit;
            if (ip >= ilen) {   // :366-368 the block's last sequence
                push(lsrc, lit, 0u, R_FINISH << 16);
                done = 1u;
                return;
            }
            if (lit != 0u) push(lsrc, lit, 0u, R_CAREFUL << 16);
        } else {
            mlc = tok_over & 15u;
            tok_over = 0u;
            ip += 1u;   // ip was parked on the byte before the offset
        }
        if (ilen - ip < 2u) return fail(LZ4FLEX_DEV_E_EXPECTED_ANOTHER_BYTE);   // :373-375
        const uint32_t offset = rd8(ip) | (rd8(ip + 1u) << 8);
        ip += 2u;
        if (offset == 0u) return fail(LZ4FLEX_DEV_E_OFFSET_ZERO);
        uint32_t ml = 4u + mlc;
        if (ml == 19u) {
            uint64_t acc = ml;
            for (;;) {
                if (ip >= ilen) return fail(LZ4FLEX_DEV_E_EXPECTED_ANOTHER_BYTE);
                const uint32_t e = rd8(ip);
                ip += 1u;
                acc += e;
                if (e != 0xFFu) break;
            }
            if (acc > 0xFFFFFFFFull) {
                if (offset > op) return fail(LZ4FLEX_DEV_E_OFFSET_OUT_OF_BOUNDS);
                expected = (uint64_t)op + acc;
                return fail(LZ4FLEX_DEV_E_OUTPUT_TOO_SMALL);
            }
            ml = (uint32_t)acc;
        }
        if (offset > op) return fail(LZ4FLEX_DEV_E_OFFSET_OUT_OF_BOUNDS);       // :398-408, unsafe-flavour order
        if (ml > cap - op) { expected = (uint64_t)op + ml; return fail(LZ4FLEX_DEV_E_OUTPUT_TOO_SMALL); }
        push(ip, 0u, ml, offset | (offset < rare_below ? R_RARE << 16 : 0u));
        op += ml;
        if (ip >= ilen) return fail(LZ4FLEX_DEV_E_EXPECTED_ANOTHER_BYTE);       // :439-443
    }

    // One step = window() ; [patch_tail() if any lane of the wave reads its tail copy] ; parse() ; [exact_step() for
    // lanes whose sequence left the fast path].
    uint32_t qhead_, ahead_, rel_;     // carried from window() to parse()
    const lds_u8* a_;                  // LDS address of the dword that holds the byte at ip
    uint32_t sh_;
    bool tailmode, slow;

    LZ4_FN void step() {
        window();
        if (LZ4_ANY(tailmode)) { patch_tail(); parse<true>(); } else parse<false>();
        if (LZ4_ANY(slow)) {
            if (slow) exact_step();
        }
    }
    LZ4_FN void window() {
        qhead_ = q.head();
        // ---- ring: take the chunk that was in flight if its slot lies before the parse position, fetch the next one
        const uint32_t ipa = ip + A;
        const uint32_t ipa4 = ipa & ~3u;
        const bool rebase = (int32_t)(ipa - vhi) >= 16;      // a long literal run led behind the chunk in flight: restart at the target (the steps until the ring is full again do not parse)
        const bool slide = !rebase && (int32_t)(ipa4 - (vhi - 48u)) >= 0;   // slot [vhi - 64, vhi - 48) is behind the dwords at ip
        if (slide && done == 0u) ring_put(N);            // (a lane without a block, or past its block's end, owns no ring)
        vhi = rebase ? (ipa & ~15u) : vhi;
        N = *reinterpret_cast<const u32x4*>(chunk_addr(vhi));
        ahead_ = vhi - ipa;                                  // valid bytes from ip on (signed)
        tailmode = (ip + 48u > ilen) & (done == 0u);         // (a finished lane reads whatever its ring holds: nothing of it is used)
        a_ = LZ4_ALIGNED_PLUS(ring(), ipa4 & (RING - 1u));
        sh_ = ipa & 3u;
        rel_ = 0u;
    }
    // lanes within 48 bytes of their block's end read the LDS tail copy instead of the ring
    LZ4_FN void patch_tail() {
        rel_ = tailmode ? ip - tstart : 0u;
        a_ = tailmode ? q.blk + TAIL_OFF + (rel_ & ~3u) : a_;
        sh_ = tailmode ? (rel_ & 3u) : sh_;
    }
    LZ4_FN uint32_t rd4(const lds_u8* a, uint32_t sh) const {      // 4 bytes at byte sh of the dword pair at a
        const lds_u32* w = reinterpret_cast<const lds_u32*>(a);
        return lz4_alignbyte(w[1], w[0], sh);
    }
    // TAIL: some lane of the wavefront reads its tail copy (patch_tail() has run); false: every lane reads its ring
    template <bool TAIL>
    LZ4_FN void parse() {
        const uint32_t qhead = qhead_;
        const uint32_t W0 = rd4(a_, sh_);
        // ---- token (or the synthetic one left by a long literal run), literal length with at most one extension byte
        const uint32_t tokb = tok_over != 0u ? tok_over : W0;
        const uint32_t lc = (tokb >> 4) & 15u;
        const uint32_t mlc = tokb & 15u;
        const uint32_t lc15 = lc == 15u ? 1u : 0u;
        const uint32_t lit = lc + (lc15 ? (W0 >> 8) & 0xFFu : 0u);       // 270 = the extension continues (exact path)
        const uint32_t pos_off = 1u + lc15 + lit;                         // window index of the offset
        // ---- offset and match-length extension, pos_off bytes behind ip (used when pos_off <= 17: inside the 24 valid bytes)
        const uint32_t po = pos_off < 20u ? pos_off : 20u;                   // (a longer run is not read here: keep the address inside the buffers)
        const uint32_t x2 = ip + A + po;
        const lds_u8* b = LZ4_ALIGNED_PLUS(ring(), x2 & (RING - 4u));
        uint32_t shb = x2 & 3u;
        if (TAIL) {
            const uint32_t r2 = rel_ + po;
            b = tailmode ? q.blk + TAIL_OFF + (r2 & ~3u) : b;
            shb = tailmode ? (r2 & 3u) : shb;
        }
        // the second sequence's token sits behind the first one's offset (and one length byte if its match nibble is 15):
        // its address is known before the offset is, both reads share one LDS round trip
        const uint32_t lit_src = ip + 1u + lc15;
        const uint32_t lit_end = lit_src + lit;
        const uint32_t seq_end = lit_end + 2u + (mlc == 15u ? 1u : 0u);
        const uint32_t xa = seq_end + A;
        const uint32_t t = rd4(b, shb);
        uint32_t V0 = 0u;
        if (!TAIL) V0 = rd4(LZ4_ALIGNED_PLUS(ring(), xa & (RING - 4u)), xa & 3u);
        const uint32_t offset = t & 0xFFFFu;
        const uint32_t ee = mlc == 15u ? (t >> 16) & 0xFFu : 0u;          // 255 = the extension continues (exact path)
        const uint32_t ml = 4u + mlc + ee;
        const uint32_t mstart = op + lit;
        // ---- classify.  Every condition is a signed slack (>= 0 holds), folded with min: sizes are below 2 GiB, larger
        // values only send a sequence to the exact path.  decompress.rs:334-408 in one go for the plain sequence:
        const int32_t common = imin3((int32_t)(0u - done), (TAIL && tailmode) ? 0 : (int32_t)(ahead_ - 24u),   // live, the ring holds 24 bytes from ip on
                                     (int32_t)(QD - 3u - (qtail - qhead)));                              // queue has room
        const uint32_t tail_need = lit_slack - 2u;                                                       // >= 1; seq_end >= lit_end + 2
        const int32_t short_s = imin3(imin3((int32_t)(ilen - tail_need - seq_end),                       // a byte follows the sequence (lit_slack bytes its literals)
                                            (int32_t)(mstart - offset), (int32_t)(offset - 1u)),         // 1 <= offset <= output so far
                                      imin3((int32_t)(cap - mstart - ml), (int32_t)(17u - pos_off),      // fits; offset inside the window
                                            (int32_t)(254u - ee)), common);
        // a literal run too long for the window: push it alone, parse the offset next time
        const int32_t long_s = imin3(imin3((int32_t)(pos_off - 18u), (int32_t)(269u - lit), (int32_t)(ilen - lit_slack - lit_end)),
                                     imin3((int32_t)(cap - mstart), (int32_t)(0u - tok_over), common), 0);
        const bool is_short = short_s >= 0;
        const bool is_long = long_s >= 0;
        slow = common >= 0 && !is_short && !is_long;
#ifdef LZ4FLEX_PROFILE_PHASES
        pr_steps += done == 0u; pr_noroom += done == 0u && (int32_t)(QD - 3u - (qtail - qhead)) < 0;
        pr_bubble += done == 0u && !tailmode && (int32_t)(ahead_ - 24u) < 0; pr_slow += slow; pr_pushed += (is_short | is_long);
#endif
#ifdef LZ4FLEX_SPLIT_DEBUG
        if (slow && dbg_ip == 0xFFFFFFFFu) { dbg_ip = ip; dbg_w0 = W0; dbg_w1 = t; dbg_kb = (ahead_ << 24) | (vhi & 0xFFFFFFu); }
#endif
        // ---- a second sequence in the same step, when both are plain short ones (the usual case): the block's token chain is
        // what the whole kernel waits for (its longest block), and the second sequence's token can be read as soon as the
        // first one's length bytes are known -- three dependent LDS round trips for two sequences instead of four, and a
        // step's fixed instructions once.  Needs 44 valid bytes in the ring, room for a second record, and the second
        // sequence itself away from the block's tail copy; anything else is simply left to the next step.
        bool two = false;
        uint32_t lit_src2 = 0u, lit2 = 0u, ml2 = 0u, offset2 = 0u, seq_end2 = 0u, op_end2 = 0u;
        if (!TAIL) {
            const uint32_t ip1 = seq_end, op1 = mstart + ml;
            const uint32_t lcb = (V0 >> 4) & 15u, mlcb = V0 & 15u;
            const uint32_t l15 = lcb == 15u ? 1u : 0u;
            lit2 = lcb + (l15 ? (V0 >> 8) & 0xFFu : 0u);
            const uint32_t pob = 1u + l15 + lit2;
            const uint32_t xb = xa + (pob < 20u ? pob : 20u);
            lit_src2 = ip1 + 1u + l15;
            const uint32_t lit_end2 = lit_src2 + lit2;
            seq_end2 = lit_end2 + 2u + (mlcb == 15u ? 1u : 0u);
            const uint32_t t2 = rd4(LZ4_ALIGNED_PLUS(ring(), xb & (RING - 4u)), xb & 3u);
            offset2 = t2 & 0xFFFFu;
            const uint32_t ee2 = mlcb == 15u ? (t2 >> 16) & 0xFFu : 0u;
            ml2 = 4u + mlcb + ee2;
            const uint32_t mstart2 = op1 + lit2;
            op_end2 = mstart2 + ml2;
            const int32_t s2 = imin3(imin3((int32_t)(ilen - tail_need - seq_end2), (int32_t)(mstart2 - offset2), (int32_t)(offset2 - 1u)),
                                     imin3((int32_t)(cap - mstart2 - ml2), (int32_t)(17u - pob), (int32_t)(254u - ee2)),
                                     imin3(short_s, imin3((int32_t)(ahead_ - 44u), (int32_t)(QD - 4u - (qtail - qhead)), (int32_t)(ilen - 48u - ip1)), 0));
            two = s2 >= 0;
        }
        // ---- commit
        push_if(is_short | is_long, lit_src, lit, is_short ? ml : 0u, is_short ? (offset | (offset < rare_below ? R_RARE << 16 : 0u)) : 0u);
        if (!TAIL) push_if(two, lit_src2, lit2, ml2, offset2 | (offset2 < rare_below ? R_RARE << 16 : 0u));
        // is_short and is_long exclude each other; one select per case (a nested ?: chain here became exec-mask branches)
        uint32_t ip2 = is_short ? seq_end : ip, op2 = is_short ? mstart + ml : op, to2 = is_short ? 0u : tok_over;
        ip = is_long ? lit_end - 1u : ip2;
        op = is_long ? mstart : op2;
        tok_over = is_long ? (0x100u | mlc) : to2;
        if (!TAIL) { ip = two ? seq_end2 : ip; op = two ? op_end2 : op; }
    }
};
using Queue = QueueT<LayoutBig>;
using Parser = ParserT<LayoutBig>;

}  // namespace v5
}  // namespace lz4flex_dev
