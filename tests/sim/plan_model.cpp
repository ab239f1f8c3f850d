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
    std::vector<uint32_t> steps;     // 4 words per step
    std::vector<uint32_t> tailv;
    uint32_t tail_op = 0u, main_bytes = 0u;
    void step(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3) {
        steps.push_back(w0); steps.push_back(w1); steps.push_back(w2); steps.push_back(w3);
        main_bytes += rec_n(w0) + rec_n(w1) + rec_n(w2) + rec_n(w3);
    }
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
    Emit e;
    emit_init(e, in_len);
    uint32_t p = 0;
    for (;;) {
        pcd::Seq s;
        const uint32_t nx = pcd::parse_seq(rd, in_len, p, s);
        emit_literals(e, s.lit_src, s.lit, sink);
        if (s.ml != 0u) emit_match(e, s.off, s.ml, sink);
        if (nx == pcd::X_END) break;
        p = nx;
    }
    emit_end(e, sink);
    if (e.op != (uint32_t)E || sink.tailv.size() > MAX_TAIL) return 1;
    if (sink.tailv.empty()) sink.tail_op = sink.main_bytes;
    *E_out = (uint32_t)E;
    return 0;
}

// steps + a step of K_END, padded with K_END to whole turns, + END_TURNS turns of K_END
uint32_t padded_turns(uint32_t n_steps) { return (n_steps + 1u + TURN_STEPS - 1u) / TURN_STEPS + END_TURNS; }

// the kernel's layout: lz4_plan_common.h word_of
void lay_out(const VecSink& s, uint32_t* words) {
    const uint32_t n_steps = (uint32_t)(s.steps.size() / 4u);
    const uint32_t total = padded_turns(n_steps) * TURN_STEPS;
    for (uint32_t st = 0; st < total; ++st)
        for (uint32_t g = 0; g < G; ++g) words[word_of(st, g)] = st < n_steps ? s.steps[4u * st + g] : END_REC;
}

}  // namespace

extern "C" {

// words: [turns: steps, then K_END | tail records]; returns total words written, 0 = irregular block, -1 = no room
int64_t plan_compile(const uint8_t* in, uint32_t in_len, uint32_t cap, uint32_t* words, uint32_t max_words, uint32_t* n_steps,
                     uint32_t* tail_word, uint32_t* n_tail, uint32_t* E) {
    VecSink s;
    if (compile_block(in, in_len, cap, s, E)) return 0;
    const uint32_t ns = (uint32_t)(s.steps.size() / 4u);
    const uint32_t pm = padded_turns(ns) * TURN_WORDS;
    if (pm + s.tailv.size() > max_words) return -1;
    lay_out(s, words);
    for (size_t i = 0; i < s.tailv.size(); ++i) words[pm + i] = s.tailv[i];
    *n_steps = ns;
    *tail_word = pm;
    *n_tail = (uint32_t)s.tailv.size();
    return (int64_t)(pm + s.tailv.size());
}

// The replay kernel's lanes, byte for byte.  `out` has cap bytes.  Returns 0, or a negative code naming the guard that fired.
int plan_replay(const uint8_t* in, uint32_t in_len, const uint32_t* words, uint32_t tail_word, uint32_t n_tail, uint32_t E,
                uint8_t* out, uint32_t cap) {
    if (E > cap) return -1;
    std::vector<uint8_t> ring(RING_STRIDE, 0xEE);
    std::vector<uint8_t> written(E + 1u, 0);         // which output bytes are in memory
    std::vector<uint8_t> ringfinal(E + 128u, 0);     // is the ring's copy of this output position the final byte
    struct Slot { uint32_t r[G]; uint8_t v[G][16]; };
    std::vector<Slot> slots(LOOKAHEAD);
    uint32_t op = 0, F = 0;
    auto word = [&](uint32_t step, uint32_t g) { return words[word_of(step, g)]; };
    auto fe = [&](uint32_t step, Slot& sl) -> int {      // request the step's bytes
        for (uint32_t g = 0; g < G; ++g) {
            const uint32_t r = word(step, g);
            sl.r[g] = r;
            const uint32_t kind = rec_kind(r), n = rec_n(r), field = rec_field(r);
            memset(sl.v[g], 0xAB, 16);
            if (n == 0u) continue;
            if (kind == K_LIT) {
                if (field + 16u > in_len) return -2;                            // a lane reads behind the compressed block
                memcpy(sl.v[g], in + field, 16);
            } else if (kind == K_FAR) {
                if (field + 16u > E) return -3;
                for (uint32_t k = 0; k < n; ++k) if (written[field + k] != 2) return -4;   // a byte the lane needs is not in memory yet
                memcpy(sl.v[g], out + field, 16);
            }
        }
        return 0;
    };
    for (uint32_t i = 0; i < LOOKAHEAD; ++i) { const int e = fe(i, slots[i]); if (e) return e; }
    uint32_t idx = 0;
    for (;;) {
        Slot& sl = slots[idx % LOOKAHEAD];
        if (rec_kind(sl.r[0]) == K_END) {
            for (uint32_t g = 0; g < G; ++g) if (sl.r[g] != END_REC) return -16;
            break;
        }
        // all lanes read ...
        uint8_t x[G][16];
        uint32_t total = 0;
        for (uint32_t g = 0; g < G; ++g) {
            const uint32_t r = sl.r[g], n = rec_n(r);
            if (n > 16u) return -17;
            if (n != 0u && rec_rel(r) != total) return -18;       // lanes in ascending, gapless order
            total += n;
            if (n == 0u) continue;
            if (rec_kind(r) == K_NEAR) {
                const uint32_t a = rec_field(r);
                if (a + 16u > RING_STRIDE) return -5;
                memcpy(x[g], &ring[a], 16);
            } else {
                memcpy(x[g], sl.v[g], 16);
            }
        }
        // ... then write in lane order, 16 bytes each
        for (uint32_t g = 0; g < G; ++g) {
            const uint32_t r = sl.r[g], n = rec_n(r);
            if (n == 0u) continue;
            const uint32_t pos = op + rec_rel(r), d = pos & MASK;
            if (d + 16u > RING_STRIDE) return -6;
            if (d + n > W) return -19;                             // the lane's own bytes never wrap
            memcpy(&ring[d], x[g], 16);
            for (uint32_t k = 0; k < 16u; ++k) if (pos + k < ringfinal.size()) ringfinal[pos + k] = k < n ? 1 : 0;
        }
        op += total;
        if (idx % FLUSH_EVERY == FLUSH_EVERY - 1u) {
            while (F + PIECE <= op) {
                if (F + PIECE > E) return -7;                                       // a store behind the block's output
                for (uint32_t k = 0; k < PIECE; ++k) {
                    if (!ringfinal[F + k]) return -8;                               // a byte that is not final leaves the ring
                    out[F + k] = ring[(F + k) & MASK];
                    written[F + k] = 2;
                }
                F += PIECE;
            }
        }
        const int e = fe(idx + LOOKAHEAD, slots[idx % LOOKAHEAD]);
        if (e) return e;
        idx++;
    }
    // every complete line that is still in the ring, the bytes behind the last full line, then the tail: byte by byte in global memory
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
// (first_word of every block is a multiple of TURN_WORDS), or -1 when max_words is too small.  Irregular blocks get flags = 1 and no records.
int64_t plan_compile_batch(const uint8_t* in_base, const uint64_t* in_off, const uint32_t* in_len, const uint64_t* out_off,
                           const uint32_t* out_cap, uint32_t n, void* plans, uint32_t* words, uint64_t max_words, uint32_t* out_len,
                           uint64_t* n_steps_total) {
    BlockPlan* bp = (BlockPlan*)plans;
    uint64_t w = 0, steps = 0;
    for (uint32_t b = 0; b < n; ++b) {
        VecSink s;
        uint32_t E = 0;
        bp[b].in_off = in_off[b]; bp[b].out_off = out_off[b];
        bp[b].first_word = (uint32_t)w; bp[b].tail_word = 0u; bp[b].tail_op = 0u; bp[b].n_tail = 0u; bp[b].flags = 0u;
        out_len[b] = 0u;
        if (compile_block(in_base + in_off[b], in_len[b], out_cap[b], s, &E)) { bp[b].flags = 1u; continue; }
        const uint32_t ns = (uint32_t)(s.steps.size() / 4u);
        const uint32_t pm = padded_turns(ns) * TURN_WORDS;
        const uint64_t total = ((uint64_t)pm + s.tailv.size() + TURN_WORDS - 1u) / TURN_WORDS * TURN_WORDS;
        if (w + total > max_words || w + total > 0xFFFFFFFFull) return -1;
        lay_out(s, words + w);
        for (size_t i = 0; i < s.tailv.size(); ++i) words[w + pm + i] = s.tailv[i];
        for (uint64_t i = pm + s.tailv.size(); i < total; ++i) words[w + i] = END_REC;
        bp[b].tail_word = (uint32_t)(w + pm);
        bp[b].n_tail = (uint16_t)s.tailv.size();
        bp[b].tail_op = s.tail_op;
        out_len[b] = E;
        w += total;
        steps += ns;
    }
    if (n_steps_total) *n_steps_total = steps;
    return (int64_t)w;
}

}  // extern "C"
