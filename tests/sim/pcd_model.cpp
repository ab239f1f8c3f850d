// pcd_model.cpp -- host model of the PARALLEL-CHAIN decoder (lz4_flex_amd/csrc/lz4_decompress_pcd.hip).  TEST INFRASTRUCTURE:
// it runs the kernel's ALGORITHM -- tiles, parts, speculative walks, exit following, re-walks until nothing changes, batches
// of sequences on an output window with history, dependency ranges between the matches of a batch, the giant-sequence path --
// sequentially on the CPU, with every geometry constant a run-time parameter so that tiny geometries put tile, part, batch
// and window boundaries everywhere in small inputs.  tests/test_pcd_model.py checks it against the oracle (the reference's
// decoder restated): same bytes for every regular block, "irregular" (-> the reference-order kernel) for everything else.
// The per-lane sequence walker is the kernel's own (csrc/lz4_pcd_common.h).
//
// What "a lane" does in the kernel is a loop body here; the matches of a batch are executed in a RANDOM order among the ones
// whose producers are done (the kernel's wavefronts poll "done" bits and run in no particular order), so a dependency range
// that is too small shows up as wrong bytes.
#define LZ4FLEX_HOST_SIM 1
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "../../lz4_flex_amd/csrc/lz4_pcd_common.h"

using namespace lz4flex_dev::pcd;

extern "C" {

struct pcd_params {
    uint32_t ct, p, batch, hist, wnew, max_iters;
};
struct pcd_stats {
    uint64_t tiles, iters, part_walks, hops, batches, seqs, giants, far_bytes, near_bytes, depth_sum, depth_max, dirty_after_first;
};

}  // extern "C"

namespace {

struct Rd {
    const uint8_t* c;
    uint32_t operator()(uint32_t pos) const { return c[pos]; }
    uint32_t u32(uint32_t pos) const { uint32_t v; memcpy(&v, c + pos, 4); return v; }
};

struct Rng {
    uint64_t s;
    uint32_t next() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(s >> 33); }
};

}  // namespace

extern "C" void pcd_model_defaults(pcd_params* p) {
    p->ct = CT; p->p = P; p->batch = BATCH; p->hist = HIST; p->wnew = WNEW; p->max_iters = MAX_ITERS;
}

// Returns the decoded length, -1: irregular (the kernel would leave the block to the reference-order kernel).
extern "C" long pcd_model_decode(const uint8_t* c, uint32_t n, uint8_t* out, uint64_t cap, const pcd_params* prm, pcd_stats* st,
                                 uint64_t seed) {
    const uint32_t ct = prm->ct, pp = prm->p, np = ct / pp;
    if (n == 0 || pp == 0 || ct % pp != 0 || pp % 32 != 0) return -1;
    pcd_stats dummy;
    if (!st) st = &dummy;
    memset(st, 0, sizeof *st);
    Rd rd{c};
    Rng rng{seed * 2654435761ull + 12345};
    std::vector<uint32_t> e(np), x(np), tok;
    std::vector<uint8_t> dirty(np), live(np), marks(ct);
    std::vector<uint8_t> win((size_t)prm->hist + prm->wnew + 64);
    uint64_t OP = 0;       // output position
    uint32_t hist = 0;     // valid history bytes in the window: win[0 .. hist) = out[OP - hist .. OP)
    uint32_t cbase = 0;    // the tile's first byte: a true token position
    bool ended = false;
    while (!ended) {
        st->tiles++;
        // ---- parse: parts of the tile that start inside the block
        const uint32_t parts = std::min<uint64_t>(np, ((uint64_t)(n - cbase) + pp - 1) / pp);
        for (uint32_t k = 0; k < parts; k++) { e[k] = cbase + k * pp; dirty[k] = 1; x[k] = X_ERR; }
        std::fill(marks.begin(), marks.end(), 0);
        uint32_t tile_exit = X_ERR;
        bool converged = false;
        for (uint32_t it = 0; it < prm->max_iters && !converged; it++) {
            st->iters++;
            bool exit_changed = it == 0;
            for (uint32_t k = 0; k < parts; k++) {            // every lane with a dirty part: walk it from its entry
                if (!dirty[k]) continue;
                dirty[k] = 0;
                st->part_walks++;
                const uint32_t pend = cbase + (k + 1) * pp;
                for (uint32_t i = k * pp; i < (k + 1) * pp; i++) marks[i] = 0;
                uint32_t p = e[k], xn;
                for (;;) {
                    if (p >= pend) { xn = p; break; }
                    marks[p - cbase] = 1;
                    Seq s;
                    const uint32_t nx = parse_seq<Rd, false>(rd, n, p, s);      // (the walk does not look at offsets: lz4_pcd_common.h)
                    st->hops++;
                    if (nx >= X_ERR) { xn = nx; break; }
                    p = nx;
                }
                exit_changed |= xn != x[k];
                x[k] = xn;
            }
            // (the kernel: no exit changed since the exits were last followed => entries and live parts stand, the tile is settled)
            if (!exit_changed) { converged = true; break; }
            // follow the exits from part 0 (its entry is the tile's true start)
            std::fill(live.begin(), live.end(), 0);
            uint32_t nd = 0;
            for (uint32_t k = 0;;) {
                live[k] = 1;
                const uint32_t nx = x[k];
                if (nx >= X_ERR) { tile_exit = nx; break; }
                const uint32_t k2 = (nx - cbase) / pp;
                if (k2 >= parts) { tile_exit = nx; break; }     // behind the tile (or behind the block's last part: cannot be, nx < n)
                if (e[k2] != nx) { e[k2] = nx; dirty[k2] = 1; nd++; }
                k = k2;
            }
            if (it == 0) st->dirty_after_first += nd;
            converged = nd == 0;
        }
        if (!converged || tile_exit == X_ERR) return -1;
        ended = tile_exit == X_END;
        // ---- the tile's sequences: set bits of the live parts, in order
        tok.clear();
        for (uint32_t k = 0; k < parts; k++)
            if (live[k])
                for (uint32_t i = k * pp; i < (k + 1) * pp; i++)
                    if (marks[i]) tok.push_back(cbase + i);
        // ---- copy: batches of consecutive sequences
        size_t idx = 0;
        std::vector<Seq> sq(prm->batch);
        std::vector<uint64_t> start(prm->batch + 1);     // output position of each sequence of the batch (+ the end)
        std::vector<uint8_t> done(prm->batch);
        std::vector<uint32_t> level(prm->batch), order;
        while (idx < tok.size()) {
            const uint32_t m = (uint32_t)std::min<size_t>(prm->batch, tok.size() - idx);
            uint64_t acc = 0;
            uint32_t cnt = 0;
            for (uint32_t i = 0; i < m; i++) {
                const uint32_t nx = parse_seq(rd, n, tok[idx + i], sq[i]);
                if (nx == X_ERR) return -1;                  // (an offset of zero: the walk went past it)
                if (nx == X_END && !(ended && idx + i + 1 == tok.size())) return -1;
                const uint64_t len = (uint64_t)sq[i].lit + sq[i].ml;
                if (acc + len > prm->wnew || len > (prm->wnew < 16384u ? prm->wnew : 16384u)) break;     // (the kernel's Geo::GIANT)
                start[i] = OP + acc;
                acc += len;
                cnt = i + 1;
            }
            if (cnt == 0) {
                // ---- a sequence longer than the window: alone, on the output itself (everything before OP is written back)
                st->giants++;
                const Seq& s = sq[0];
                if ((uint64_t)s.lit > cap - OP) return -1;
                memcpy(out + OP, c + s.lit_src, s.lit);
                OP += s.lit;
                if (s.ml) {
                    if (s.off > OP) return -1;
                    if ((uint64_t)s.ml > cap - OP) return -1;
                    // non-overlapping steps of growing size (the kernel's cooperative copy): n = min(rest, done + off)
                    uint64_t donem = 0;
                    while (donem < s.ml) {
                        const uint64_t step = std::min<uint64_t>(s.ml - donem, donem + s.off);
                        memcpy(out + OP + donem, out + OP - s.off, step);
                        donem += step;
                    }
                    OP += s.ml;
                }
                hist = 0;
                idx += 1;
                st->seqs++;
                continue;
            }
            st->batches++;
            st->seqs += cnt;
            start[cnt] = OP + acc;
            if (acc > cap - OP) return -1;                   // OutputTooSmall somewhere in this batch
            for (uint32_t i = 0; i < cnt; i++)
                if (sq[i].ml && sq[i].off > start[i] + sq[i].lit) return -1;   // OffsetOutOfBounds
            const uint64_t Lo = OP - hist;                   // window byte j = output position Lo + j
            // literals: all at once
            for (uint32_t i = 0; i < cnt; i++) memcpy(&win[(size_t)(start[i] - Lo)], c + sq[i].lit_src, sq[i].lit);
            // matches: dependency range per match, then a random order among the ready ones
            std::vector<uint32_t> lo(cnt), hi(cnt);
            order.clear();
            uint32_t dmax = 0;
            for (uint32_t i = 0; i < cnt; i++) {
                done[i] = sq[i].ml == 0;
                level[i] = 0;
                lo[i] = 1; hi[i] = 0;                        // empty range
                if (!sq[i].ml) continue;
                order.push_back(i);
                const uint64_t ms = start[i] + sq[i].lit;
                const uint64_t s0 = ms - sq[i].off, s1 = std::min<uint64_t>(s0 + sq[i].ml, ms);   // the source outside its own output
                if (s1 <= OP) { level[i] = 1; continue; }   // history / already written back: no producer in this batch
                const uint64_t a = std::max<uint64_t>(s0, OP);
                // largest j with start[j] <= a; largest j with start[j] <= s1 - 1
                uint32_t l = (uint32_t)(std::upper_bound(start.begin(), start.begin() + cnt, a) - start.begin()) - 1;
                uint32_t h = (uint32_t)(std::upper_bound(start.begin(), start.begin() + cnt, s1 - 1) - start.begin()) - 1;
                if (l >= i) { level[i] = 1; continue; }      // the source lies in its own literals only (placed already)
                if (h >= i) h = i - 1;
                else if (s1 <= start[h] + sq[h].lit) {       // the source ends inside sequence h's literals: its match is no producer
                    if (h == l) { level[i] = 1; continue; }
                    h -= 1;
                }
                lo[i] = l; hi[i] = h;
                uint32_t lv = 0;
                for (uint32_t j = l; j <= h; j++) lv = std::max(lv, level[j]);
                level[i] = lv + 1;
            }
            for (uint32_t i = 0; i < cnt; i++) dmax = std::max(dmax, level[i]);
            st->depth_sum += dmax;
            st->depth_max = std::max<uint64_t>(st->depth_max, dmax);
            size_t left = order.size();
            while (left) {
                // pick a random undone match; take it if its producers are done (the kernel's lanes poll the same condition)
                const size_t r = rng.next() % left;
                const uint32_t i = order[r];
                bool ready = true;
                for (uint32_t j = lo[i]; j <= hi[i] && lo[i] <= hi[i]; j++) ready &= done[j] != 0;
                if (!ready) continue;
                const uint64_t ms = start[i] + sq[i].lit;
                for (uint32_t b = 0; b < sq[i].ml; b++) {    // byte-serial forward copy: decompress.rs:57-82 semantics
                    const uint64_t s = ms - sq[i].off + b;
                    uint8_t v;
                    if (s < Lo) { v = out[s]; st->far_bytes++; }
                    else { v = win[(size_t)(s - Lo)]; st->near_bytes++; }
                    win[(size_t)(ms + b - Lo)] = v;
                }
                done[i] = 1;
                order[r] = order[left - 1];
                left--;
            }
            // write back, slide the history
            memcpy(out + OP, &win[hist], (size_t)acc);
            OP += acc;
            const uint32_t have = hist + (uint32_t)acc;
            const uint32_t keep = std::min<uint32_t>(prm->hist, have);
            memmove(&win[0], &win[have - keep], keep);
            hist = keep;
            idx += cnt;
        }
        if (!ended) cbase = tile_exit;
    }
    return (long)OP;
}
