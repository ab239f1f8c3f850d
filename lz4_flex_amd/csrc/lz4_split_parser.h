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
#define LZ4_ANY(x) (x)
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

namespace lz4flex_dev {
namespace v5 {

typedef uint32_t __attribute__((ext_vector_type(4))) u32x4;
typedef uint8_t LZ4_LDS lds_u8;
typedef volatile uint32_t LZ4_LDS lds_vu32;
typedef volatile u32x4 LZ4_LDS lds_vu128;
typedef uint32_t LZ4_LDS lds_u32;

constexpr uint32_t G = 8;             // copier lanes per block
constexpr uint32_t PIECE = 4u * G;    // bytes per copy step
constexpr uint32_t QD = 16;           // records per queue
constexpr uint32_t OUT_H = 512;       // history kept in LDS after a write-back
constexpr uint32_t OUT_SLACK = 32;
constexpr uint32_t OUT_CAP = 2080;
constexpr uint32_t FLUSH_AT = 760;
constexpr uint32_t TAILB = 48;        // bytes of the block's end staged in LDS
constexpr uint32_t TAIL_BUF = 80;     // + zero padding: a 24-byte window read at any tail position stays inside
constexpr uint32_t Q_OFF = OUT_CAP;
constexpr uint32_t CTL_OFF = Q_OFF + 16u * QD;   // head, tail
constexpr uint32_t TAIL_OFF = CTL_OFF + 16u;
constexpr uint32_t BLK_LDS = TAIL_OFF + TAIL_BUF;
static_assert(BLK_LDS == 2432 && BLK_LDS % 16 == 0, "LDS per block");
constexpr uint32_t PF_AHEAD = 512;    // compressed bytes kept warm ahead of the records being copied

constexpr uint32_t F_FIN = 1u << 16;      // record: the block ends after these literals (or with an error)
constexpr uint32_t F_CAREFUL = 1u << 17;  // record: literal source may end at the block's last byte (no wild reads)

#ifdef LZ4FLEX_HOST_SIM
static uint8_t g_pad[64] __attribute__((aligned(16)));
#else
__device__ __attribute__((aligned(16))) uint8_t g_pad[64];
#endif   // always-readable target of loads that fetch nothing

LZ4_FN uint32_t ld32(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
LZ4_FN uint32_t ld32l(const lds_u8* p) { uint32_t v; __builtin_memcpy(&v, (const void*)p, 4); return v; }

struct Queue {
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
struct Parser {
    const uint8_t* gin;      // compressed block
    const uint8_t* gal;      // gin rounded down to 4 bytes: the register window lives in this "aligned space"
    uint32_t A;              // gin - gal
    uint32_t ilen, cap;
    uint32_t ip, op;
    uint32_t need_off, mlc_saved;
    uint32_t done;
    int32_t status;
    uint64_t expected;
    uint32_t qtail;
    uint32_t tstart;         // the LDS tail copy holds compressed positions [tstart, ilen)
    Queue q;
    // register window: stream bytes [base, base + 48) of the aligned space, N = the chunk at base + 48
    uint32_t base;
    u32x4 C0, C1, C2, N;
#ifdef LZ4FLEX_SPLIT_DEBUG
    uint32_t dbg_ip, dbg_w0, dbg_w1, dbg_kb;   // state at the first sequence that left the fast path
#endif

    static LZ4_FN uint32_t bfi(uint32_t m, uint32_t a, uint32_t b) { return (a & m) | (b & ~m); }
    static LZ4_FN uint32_t sel4(uint32_t m0, uint32_t m1, uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3) {
        return bfi(m1, bfi(m0, a3, a2), bfi(m0, a1, a0));
    }
    LZ4_FN const uint8_t* chunk_addr(uint32_t c) const {
        // a chunk is fetched only if it lies entirely inside the block (the tail copy serves the rest)
        return (c + 16u <= ilen + A) ? gal + c : g_pad;
    }
    LZ4_FN uint32_t rd8(uint32_t pos) const {   // pos < ilen
        return pos >= tstart ? (uint32_t)q.blk[TAIL_OFF + (pos - tstart)] : (uint32_t)gin[pos];
    }
    LZ4_FN void push(uint32_t lsrc, uint32_t ln, uint32_t ml, uint32_t off_flags) {
        q.put(qtail, lsrc, ln, ml, off_flags);
        qtail += 1u;
        q.set_tail(qtail);
    }
    LZ4_FN void fail(int32_t code) {
        status = code;
        done = 1u;
        push(0u, 0u, 0u, F_FIN);
    }

    // Exact handling of one sequence (or of the offset half when need_off is set), byte by byte, every check in the
    // reference's order (src/block/decompress.rs:244-444).  Needs three free queue slots.
    LZ4_COLD_FN void exact_step() {
        uint32_t mlc;
        if (!need_off) {
            const uint32_t tok = rd8(ip);
            ip += 1u;
            uint32_t lit = tok >> 4;
            mlc = tok & 15u;
            if (lit == 15u) {
                for (;;) {   // read_integer_ptr :126-157
                    if (ip >= ilen) return fail(LZ4FLEX_DEV_E_EXPECTED_ANOTHER_BYTE);
                    const uint32_t e = rd8(ip);
                    ip += 1u;
                    lit += e;
                    if (e != 0xFFu) break;
                }
            }
            if (lit > ilen - ip) return fail(LZ4FLEX_DEV_E_LITERAL_OUT_OF_BOUNDS);
            if (lit > cap - op) { expected = (uint64_t)op + lit; return fail(LZ4FLEX_DEV_E_OUTPUT_TOO_SMALL); }
            const uint32_t lsrc = ip;
            ip += lit;
            op += lit;
            if (ip >= ilen) {   // :366-368 the block's last sequence
                push(lsrc, lit, 0u, F_FIN | F_CAREFUL);
                done = 1u;
                return;
            }
            if (lit != 0u) push(lsrc, lit, 0u, F_CAREFUL);
        } else {
            mlc = mlc_saved;
            need_off = 0u;
        }
        if (ilen - ip < 2u) return fail(LZ4FLEX_DEV_E_EXPECTED_ANOTHER_BYTE);   // :373-375
        const uint32_t offset = rd8(ip) | (rd8(ip + 1u) << 8);
        ip += 2u;
        if (offset == 0u) return fail(LZ4FLEX_DEV_E_OFFSET_ZERO);
        uint32_t ml = 4u + mlc;
        if (ml == 19u) {
            for (;;) {
                if (ip >= ilen) return fail(LZ4FLEX_DEV_E_EXPECTED_ANOTHER_BYTE);
                const uint32_t e = rd8(ip);
                ip += 1u;
                ml += e;
                if (e != 0xFFu) break;
            }
        }
        if (offset > op) return fail(LZ4FLEX_DEV_E_OFFSET_OUT_OF_BOUNDS);       // :398-408, unsafe-flavour order
        if (ml > cap - op) { expected = (uint64_t)op + ml; return fail(LZ4FLEX_DEV_E_OUTPUT_TOO_SMALL); }
        push(ip, 0u, ml, offset);
        op += ml;
        if (ip >= ilen) return fail(LZ4FLEX_DEV_E_EXPECTED_ANOTHER_BYTE);       // :439-443
    }

    LZ4_FN void step() {
        const uint32_t qhead = q.head();
        // ---- register window: take the chunk that was in flight, slide by at most one chunk, fetch the next one
        const uint32_t ipa = ip + A;
        uint32_t k = ipa - base;
        const bool rebase = k >= 64u;            // a long literal run was skipped: restart three chunks before the target
        const bool slide = !rebase && k >= 16u;
        if (slide) { C0 = C1; C1 = C2; C2 = N; }
        base = rebase ? (ipa & ~15u) - 48u : (slide ? base + 16u : base);
        N = *reinterpret_cast<const u32x4*>(chunk_addr(base + 48u));
        k = ipa - base;
        const bool tailmode = ip + 48u > ilen;
        const bool have = k < 16u || tailmode;
        // ---- 24 bytes at ip: D0..D5 dwords -> W0..W4
        const uint32_t i = (k >> 2) & 3u;
        // 4-way selects written as bit-field inserts on the two index bits (v_bfi_b32): three instructions per dword,
        // no control flow (nested ?: chains on six values were lowered to exec-mask branches)
        const uint32_t m0 = 0u - (i & 1u), m1 = 0u - (i >> 1);
        uint32_t D0 = sel4(m0, m1, C0.x, C0.y, C0.z, C0.w);
        uint32_t D1 = sel4(m0, m1, C0.y, C0.z, C0.w, C1.x);
        uint32_t D2 = sel4(m0, m1, C0.z, C0.w, C1.x, C1.y);
        uint32_t D3 = sel4(m0, m1, C0.w, C1.x, C1.y, C1.z);
        uint32_t D4 = sel4(m0, m1, C1.x, C1.y, C1.z, C1.w);
        uint32_t D5 = sel4(m0, m1, C1.y, C1.z, C1.w, C2.x);
        uint32_t sh = k & 3u;
        if (LZ4_ANY(tailmode)) {
            const uint32_t rel = tailmode ? ip - tstart : 0u;
            const lds_u32* tw = reinterpret_cast<const lds_u32*>(q.blk + TAIL_OFF + (rel & ~3u));
            const uint32_t t0 = tw[0], t1 = tw[1], t2 = tw[2], t3 = tw[3], t4 = tw[4], t5 = tw[5];
            D0 = tailmode ? t0 : D0; D1 = tailmode ? t1 : D1; D2 = tailmode ? t2 : D2;
            D3 = tailmode ? t3 : D3; D4 = tailmode ? t4 : D4; D5 = tailmode ? t5 : D5;
            sh = tailmode ? (rel & 3u) : sh;
        }
        const uint32_t W0 = lz4_alignbyte(D1, D0, sh);
        const uint32_t W1 = lz4_alignbyte(D2, D1, sh);
        const uint32_t W2 = lz4_alignbyte(D3, D2, sh);
        const uint32_t W3 = lz4_alignbyte(D4, D3, sh);
        const uint32_t W4 = lz4_alignbyte(D5, D4, sh);
        // ---- token fields (meaningful when !need)
        const bool need = need_off != 0u;
        const uint32_t lc = (W0 >> 4) & 15u;
        const uint32_t mlc_t = W0 & 15u;
        const uint32_t e1 = (W0 >> 8) & 0xFFu;
        const bool lc15 = lc == 15u;
        const uint32_t lit_t = lc15 ? 15u + e1 : lc;
        const uint32_t hdr = lc15 ? 2u : 1u;
        const bool rare_t = lc15 && e1 == 0xFFu;
        const uint32_t lit_end = ip + hdr + lit_t;                 // blocks are far below 4 GiB - 272: no wrap
        const bool lit_fits = lit_t <= cap - op;
        const bool hdr_in = ilen - ip >= hdr;                      // ip < ilen always; the extension byte needs one more
        const bool lit_in = hdr_in && lit_t <= ilen - ip - hdr;
        // ---- offset / extension byte at window index pos_off (<= 17)
        const uint32_t lit_s = need ? 0u : lit_t;
        const uint32_t pos_off = need ? 0u : hdr + lit_t;
        const uint32_t mlc = need ? mlc_saved : mlc_t;
        const bool longlit = pos_off > 17u;
        const uint32_t wi = pos_off >> 2;
        const uint32_t lo = wi == 0u ? W0 : (wi == 1u ? W1 : (wi == 2u ? W2 : (wi == 3u ? W3 : W4)));
        const uint32_t hi = wi == 0u ? W1 : (wi == 1u ? W2 : (wi == 2u ? W3 : (wi == 3u ? W4 : 0u)));
        const uint32_t t = lz4_alignbyte(hi, lo, pos_off & 3u);
        const uint32_t offset = t & 0xFFFFu;
        const uint32_t e = (t >> 16) & 0xFFu;
        const bool ext = mlc == 15u;
        const uint32_t ml = 4u + mlc + (ext ? e : 0u);
        const uint32_t seq_end = ip + pos_off + 2u + (ext ? 1u : 0u);
        const uint32_t mstart = op + lit_s;
        // ---- classify
        const bool room = qtail - qhead <= QD - 3u;   // the exact path pushes up to three records
        const bool active = done == 0u && have && room;
        const bool tok_ok = need || (!rare_t && hdr_in);
        const bool fin = active && !need && tok_ok && lit_in && lit_end == ilen && lit_fits;
        const bool start_long = active && !need && tok_ok && longlit && lit_in && lit_end + 3u <= ilen && lit_fits;
        const bool do_short = active && tok_ok && !longlit && !(ext && e == 0xFFu) && seq_end < ilen &&
                              (need || lit_fits) && offset != 0u && offset <= mstart && ml <= cap - mstart;
        const bool slow = active && !fin && !start_long && !do_short;
        // ---- commit
        if (fin | start_long | do_short) {
            const uint32_t flags = fin ? (F_FIN | F_CAREFUL) : 0u;
            push(ip + hdr, do_short ? lit_s : lit_t, do_short ? ml : 0u, (do_short ? offset : 0u) | flags);
        }
        ip = do_short ? seq_end : ((fin | start_long) ? lit_end : ip);
        op = do_short ? mstart + ml : ((fin | start_long) ? op + lit_t : op);
        need_off = do_short ? 0u : (start_long ? 1u : need_off);
        mlc_saved = start_long ? mlc_t : mlc_saved;
        done = fin ? 1u : done;
#ifdef LZ4FLEX_SPLIT_DEBUG
        if (slow && dbg_ip == 0xFFFFFFFFu) { dbg_ip = ip; dbg_w0 = W0; dbg_w1 = W1; dbg_kb = (k << 24) | (base & 0xFFFFFFu); }
#endif
        if (LZ4_ANY(slow)) {
            if (slow) exact_step();
        }
    }
};

}  // namespace v5
}  // namespace lz4flex_dev
