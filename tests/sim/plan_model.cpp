// plan_model.cpp -- HOST MODEL of the plan / replay decoder (lz4_flex_amd/csrc/lz4_decompress_plan.hip,
// lz4_decompress_replay.hip).  Test infrastructure: compiled by tests/plan_model.py with gcc, never shipped, never timed.
//
//   plan_compile        what the plan kernel must produce for one block: the reference's parse
//                       (src/block/decompress.rs:244-443 through lz4_pcd_common.h::parse_seq) fed to the SAME record
//                       emitter the kernel uses (lz4_plan_common.h).
//   plan_replay         what the replay kernel's four lanes do with a plan, byte for byte: 16-byte moves whatever the
//                       piece's length, sources requested LOOKAHEAD records early, a 2 KiB ring with 16 bytes of pad, and
//                       guards that fail the run when a lane reads outside the compressed block / the written output or
//                       writes outside the sink.
//   plan_compile_batch  plans for a whole batch in the kernel's array layout (tools: the replay kernel can be run and
//                       timed on plans compiled here).
#define LZ4FLEX_HOST_SIM 1
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../lz4_flex_amd/csrc/lz4_pcd_common.h"
#include "../../lz4_flex_amd/csrc/lz4_plan_common.h"

using namespace lz4flex_dev;
using namespace lz4flex_dev::plan;

namespace {

struct Rd {
    const uint8_t* p;
    uint32_t operator()(uint32_t i) const { return p[i]; }
    uint32_t u32(uint32_t i) const { uint32_t v; memcpy(&v, p + i, 4); return v; }
};

struct VecSink {
    std::vector<uint32_t> mainv, tailv;
    uint32_t tail_op = 0u, main_bytes = 0u;
    void main(uint32_t r) { mainv.push_back(r); main_bytes += rec_n(r); }
    void tail(uint32_t r) { if (tailv.empty()) tail_op = main_bytes; tailv.push_back(r); }
};

// -> 0 and the plan, or 1: irregular block (any DecompressError, sink too small, lengths the records cannot hold)
int compile_block(const uint8_t* in, uint32_t in_len, uint32_t cap, VecSink& sink, uint32_t* E_out) {
    if (in_len > MAX_FIELD) return 1;
    Rd rd{in};
    // pass 1: the decoded length (the plan kernel has it from its prefix sums)
    uint64_t E = 0;
    {
        uint32_t p = 0;
        for (;;) {
            pcd::Seq s;
            const uint32_t nx = pcd::parse_seq(rd, in_len, p, s);
            if (nx == pcd::X_ERR) return 1;
            if (s.ml != 0u && s.off > E + s.lit) return 1;              // :398-402
            E += (uint64_t)s.lit + s.ml;
            if (E > cap || E > MAX_FIELD) return 1;
            if (nx == pcd::X_END) break;
            p = nx;
        }
    }
    Emit e{0u, (uint32_t)E, in_len, 0u};
    uint32_t p = 0;
    for (;;) {
        pcd::Seq s;
        const uint32_t nx = pcd::parse_seq(rd, in_len, p, s);
        emit_literals(e, s.lit_src, s.lit, sink);
        if (s.ml != 0u) emit_match(e, s.off, s.ml, sink);
        if (nx == pcd::X_END) break;
        p = nx;
    }
    if (e.op != (uint32_t)E || sink.tailv.size() > MAX_TAIL) return 1;
    if (sink.tailv.empty()) sink.tail_op = sink.main_bytes;
    *E_out = (uint32_t)E;
    return 0;
}

uint32_t padded_main_words(uint32_t n_main) {   // records + K_END up to the line's end + END_LINES lines of K_END
    const uint32_t with_end = n_main + 1u;
    return (with_end + LINE_WORDS - 1u) / LINE_WORDS * LINE_WORDS + END_LINES * LINE_WORDS;
}

}  // namespace

extern "C" {

// words: [main records | K_END padding | tail records]; returns total words written, 0 = irregular block, -1 = no room
int64_t plan_compile(const uint8_t* in, uint32_t in_len, uint32_t cap, uint32_t* words, uint32_t max_words, uint32_t* n_main,
                     uint32_t* tail_word, uint32_t* n_tail, uint32_t* E) {
    VecSink s;
    if (compile_block(in, in_len, cap, s, E)) return 0;
    const uint32_t pm = padded_main_words((uint32_t)s.mainv.size());
    if (pm + s.tailv.size() > max_words) return -1;
    for (uint32_t i = 0; i < pm; ++i) words[i] = i < s.mainv.size() ? s.mainv[i] : END_REC;
    for (size_t i = 0; i < s.tailv.size(); ++i) words[pm + i] = s.tailv[i];
    *n_main = (uint32_t)s.mainv.size();
    *tail_word = pm;
    *n_tail = (uint32_t)s.tailv.size();
    return (int64_t)(pm + s.tailv.size());
}

// The replay kernel's lanes, byte for byte.  `out` has cap bytes.  Returns 0, or a negative code naming the guard that fired.
int plan_replay(const uint8_t* in, uint32_t in_len, const uint32_t* words, uint32_t tail_word, uint32_t n_tail, uint32_t E,
                uint8_t* out, uint32_t cap) {
    if (E > cap) return -1;
    std::vector<uint8_t> ring(RING_STRIDE, 0xEE);
    std::vector<uint8_t> written(E + 1u, 0);      // which output bytes have been stored (final value or not)
    struct Slot { uint32_t r; uint8_t v[4][16]; };
    std::vector<Slot> slots(LOOKAHEAD);
    uint32_t op = 0, F = 0;
    bool done = false;
    std::vector<uint8_t> ringfinal(E + 128u, 0);   // is the ring's copy of this output position the final byte
    auto fe = [&](uint32_t idx, Slot& sl) -> int {      // request the record's bytes
        const uint32_t r = words[idx];
        sl.r = r;
        const uint32_t kind = rec_kind(r), n = rec_n(r), field = rec_field(r);
        for (uint32_t g = 0; g < 4u; ++g) {
            const uint32_t lane_off = 16u * g < n ? 16u * g : 0u;
            if (kind == K_LIT) {
                if (field + lane_off + 16u > in_len) return -2;                 // a lane reads behind the compressed block
                memcpy(sl.v[g], in + field + lane_off, 16);
            } else if (kind == K_FAR) {
                if (field + lane_off + 16u > E) return -3;
                for (uint32_t k = 0; k < 16u; ++k)
                    if (16u * g < n && lane_off + k < n && written[field + lane_off + k] != 2) return -4;   // a byte the piece needs is not final yet
                memcpy(sl.v[g], out + field + lane_off, 16);
            } else {
                memset(sl.v[g], 0xAB, 16);                                     // (idle lanes read plan words)
            }
        }
        return 0;
    };
    // prologue: the first LOOKAHEAD records
    for (uint32_t i = 0; i < LOOKAHEAD; ++i) { const int e = fe(i, slots[i]); if (e) return e; }
    uint32_t idx = 0;
    while (!done) {
        Slot& sl = slots[idx % LOOKAHEAD];
        const uint32_t r = sl.r, kind = rec_kind(r), n = rec_n(r), field = rec_field(r);
        if (kind == K_END) { done = true; break; }
        // back end: all lanes read, then all lanes write
        uint8_t x[4][16];
        for (uint32_t g = 0; g < 4u; ++g) {
            if (16u * g >= n) continue;
            if (kind == K_NEAR) {
                const uint32_t a = field + 16u * g;
                if (a + 16u > RING_STRIDE) return -5;
                memcpy(x[g], &ring[a], 16);
            } else {
                memcpy(x[g], sl.v[g], 16);
            }
        }
        const uint32_t d = op & MASK;
        for (uint32_t g = 0; g < 4u; ++g) {      // (lanes of a piece never overlap: 16-byte strides)
            if (16u * g >= n) continue;
            if (d + 16u * g + 16u > RING_STRIDE) return -6;
            memcpy(&ring[d + 16u * g], x[g], 16);
            for (uint32_t k = 0; k < 16u && op + 16u * g + k < E + 64u; ++k) {
                const uint32_t pos = op + 16u * g + k;
                if (pos < ringfinal.size()) ringfinal[pos] = (16u * g + k < n) ? 1 : 0;
            }
        }
        op += n;
        const uint32_t fl = op & ~(PIECE - 1u);
        if (fl != F) {
            if (fl != F + PIECE) return -15;
            if (F + PIECE > E) return -7;                                       // a store behind the block's output
            for (uint32_t k = 0; k < PIECE; ++k) {
                if (!ringfinal[F + k]) return -8;                               // a byte that is not final leaves the ring
                out[F + k] = ring[(F + k) & MASK];
                written[F + k] = 2;
            }
            F = fl;
        }
        // front end for the record LOOKAHEAD ahead
        const int e = fe(idx + LOOKAHEAD, slots[idx % LOOKAHEAD]);
        if (e) return e;
        idx++;
    }
    // the bytes behind the last full line, then the tail: byte by byte in global memory
    for (uint32_t k = F; k < op; ++k) {
        if (!ringfinal[k]) return -8;
        out[k] = ring[k & MASK];
        written[k] = 2;
    }
    for (uint32_t t = 0; t < n_tail; ++t) {
        const uint32_t r = words[tail_word + t], kind = rec_kind(r), n = rec_n(r), field = rec_field(r);
        for (uint32_t k = 0; k < n; ++k) {
            if (op >= E) return -9;
            if (kind == K_LIT) {
                if (field + k >= in_len) return -10;
                out[op] = in[field + k];
            } else if (kind == K_FAR) {
                if (field == 0u || field > op) return -11;
                out[op] = out[op - field];
            } else {
                return -12;
            }
            written[op] = 2;
            op++;
        }
    }
    if (op != E) return -13;
    for (uint32_t k = 0; k < E; ++k) if (written[k] != 2) return -14;
    return 0;
}

// plans for n blocks in the kernel's layout.  plans: n BlockPlan records (32 bytes each); words: the plan array; returns words used
// (first_word of every block is a multiple of LINE_WORDS), or -1 when max_words is too small.  Irregular blocks get flags = 1 and no records.
int64_t plan_compile_batch(const uint8_t* in_base, const uint64_t* in_off, const uint32_t* in_len, const uint64_t* out_off,
                           const uint32_t* out_cap, uint32_t n, void* plans, uint32_t* words, uint64_t max_words, uint32_t* out_len) {
    BlockPlan* bp = (BlockPlan*)plans;
    uint64_t w = 0;
    for (uint32_t b = 0; b < n; ++b) {
        VecSink s;
        uint32_t E = 0;
        bp[b].in_off = in_off[b]; bp[b].out_off = out_off[b];
        bp[b].first_word = (uint32_t)w; bp[b].tail_word = 0u; bp[b].tail_op = 0u; bp[b].n_tail = 0u; bp[b].flags = 0u;
        out_len[b] = 0u;
        if (compile_block(in_base + in_off[b], in_len[b], out_cap[b], s, &E)) { bp[b].flags = 1u; continue; }
        const uint32_t pm = padded_main_words((uint32_t)s.mainv.size());
        const uint64_t total = ((uint64_t)pm + s.tailv.size() + LINE_WORDS - 1u) / LINE_WORDS * LINE_WORDS;
        if (w + total > max_words || w + total > 0xFFFFFFFFull) return -1;
        for (uint32_t i = 0; i < pm; ++i) words[w + i] = i < s.mainv.size() ? s.mainv[i] : END_REC;
        for (size_t i = 0; i < s.tailv.size(); ++i) words[w + pm + i] = s.tailv[i];
        for (uint64_t i = pm + s.tailv.size(); i < total; ++i) words[w + i] = END_REC;
        bp[b].tail_word = (uint32_t)(w + pm);
        bp[b].n_tail = (uint16_t)s.tailv.size();
        bp[b].tail_op = s.tail_op;
        out_len[b] = E;
        w += total;
    }
    return (int64_t)w;
}

}  // extern "C"
