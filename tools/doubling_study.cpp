// study (CPU, no GPU): would a POINTER-DOUBLING copy phase pay for the workgroup decoder?
// For every batch of the decoder (<= 2 048 sequences / 48 KiB of output) every match byte x gets src[x] = x - offset; a source is
// TERMINAL when it lies in front of the batch (history, written back) or inside a literal run (placed before the matches).  One round
// of pointer doubling replaces every non-terminal src[x] by src[src[x]]; the rounds needed are ceil(log2(depth in bytes' hops)).
// A PIECE is a maximal run of bytes of one match whose sources are consecutive -- what a piece-wise kernel would hold as one record:
// the number of pieces after each round is the work of the next round and the number of copies at the end; "cuts" counts how often a
// piece had to be split in a round (its source straddled two records).  Matches that overlap their own output (offset < length) are
// periodic: their tail is reported separately (a kernel would resolve the first period and splat the rest).
//   g++ -O2 -o /tmp/doubling tools/doubling_study.cpp; /tmp/doubling block.lz4 [batch_seqs [wnew]]
// (block.lz4: one raw LZ4 block, e.g. written by tests/wave_model.compress or oracle_api.compress; tools/doubling_study.py drives it)
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
struct Seq { uint64_t start, ms; uint32_t lit, ml, off; };
int main(int argc, char** argv) {
    if (argc < 2) return 1;
    FILE* f = fopen(argv[1], "rb"); if (!f) return 1;
    std::vector<uint8_t> c(1 << 26); size_t n = fread(c.data(), 1, c.size(), f); fclose(f);
    const uint32_t BN = argc > 2 ? atoi(argv[2]) : 2048, WNEW = argc > 3 ? atoi(argv[3]) : 49152;
    std::vector<Seq> all; uint64_t op = 0; size_t p = 0;
    while (p < n) {
        uint32_t t = c[p++], lit = t >> 4, ml = t & 15;
        if (lit == 15) { uint32_t e; do { e = c[p++]; lit += e; } while (e == 255); }
        Seq s; s.start = op; s.lit = lit; p += lit; op += lit; s.ms = op;
        if (p >= n) { s.ml = 0; s.off = 0; all.push_back(s); break; }
        s.off = c[p] | (c[p + 1] << 8); p += 2;
        if (ml == 15) { uint32_t e; do { e = c[p++]; ml += e; } while (e == 255); }
        s.ml = ml + 4; op += s.ml; all.push_back(s);
    }
    uint64_t batches = 0, seqs = 0, matches = 0, mbytes = 0, periodic = 0, periodic_bytes = 0;
    uint64_t rounds_sum = 0, rounds_max = 0, levels_sum = 0, levels_max = 0;
    uint64_t pieces_r[16] = {0}, open_r[16] = {0}, cuts_r[16] = {0};     // after round r: pieces in all, pieces still open, cuts made in round r
    for (size_t b0 = 0; b0 < all.size();) {
        size_t cnt = 0; const uint64_t OP = all[b0].start;
        while (b0 + cnt < all.size() && cnt < BN && all[b0 + cnt].start + all[b0 + cnt].lit + all[b0 + cnt].ml - OP <= WNEW) cnt++;
        if (cnt == 0) { b0++; continue; }
        const Seq* q = &all[b0];
        const uint64_t END = q[cnt - 1].ms + q[cnt - 1].ml;
        const size_t N = (size_t)(END - OP);
        // per byte of the batch: owner match (or -1 for literal), source (absolute), terminal?
        std::vector<int32_t> owner(N, -1);
        std::vector<int64_t> src(N, -1);
        std::vector<uint8_t> is_match(N, 0), tail(N, 0);
        for (size_t i = 0; i < cnt; i++) {
            if (!q[i].ml) continue;
            matches++; mbytes += q[i].ml;
            const bool per = q[i].off < q[i].ml;
            if (per) { periodic++; periodic_bytes += q[i].ml - q[i].off; }
            for (uint32_t k = 0; k < q[i].ml; k++) {
                const size_t x = (size_t)(q[i].ms + k - OP);
                owner[x] = (int32_t)i; is_match[x] = 1;
                if (per && k >= q[i].off) { tail[x] = 1; src[x] = -2; }        // the periodic tail: splatted behind the first period, not doubled
                else src[x] = (int64_t)(q[i].ms + k) - q[i].off;
            }
        }
        auto terminal = [&](int64_t s) { return s < (int64_t)OP || !is_match[(size_t)(s - (int64_t)OP)] ; };
        // sources inside a periodic tail: map into the first period of that match (what the splat reproduces)
        auto fold = [&](int64_t s) {
            if (s < (int64_t)OP) return s;
            const size_t x = (size_t)(s - (int64_t)OP);
            if (!tail[x]) return s;
            const Seq& m = q[owner[x]];
            return (int64_t)m.ms + (int64_t)((s - (int64_t)m.ms) % m.off);
        };
        for (size_t x = 0; x < N; x++) if (is_match[x] && !tail[x]) src[x] = fold(src[x]);
        // dependency levels in sequences (what the kernel pays today, without relinking): depth in hops of the deepest byte
        {
            std::vector<uint32_t> hop(N, 0); uint32_t dm = 0;
            for (size_t x = 0; x < N; x++) {
                if (!is_match[x] || tail[x]) continue;
                const int64_t s = src[x];
                hop[x] = terminal(s) ? 1u : hop[(size_t)(s - (int64_t)OP)] + 1u;
                dm = std::max(dm, hop[x]);
            }
            levels_sum += dm; levels_max = std::max<uint64_t>(levels_max, dm);
        }
        auto src_owner = [&](int64_t sx) -> int32_t { return sx < (int64_t)OP ? -2 : owner[(size_t)(sx - (int64_t)OP)]; };
        auto count_pieces = [&](uint64_t* all_p, uint64_t* open_p) {
            uint64_t np = 0, no = 0;
            for (size_t x = 0; x < N; x++) {
                if (!is_match[x] || tail[x]) continue;
                const bool first = x == 0 || owner[x - 1] != owner[x] || tail[x - 1] || src[x - 1] + 1 != src[x] ||
                                   terminal(src[x - 1]) != terminal(src[x]) ||
                                   // a record's source lies in ONE record: consecutive sources in different matches / literal runs are two pieces
                                   src_owner(src[x]) != src_owner(src[x - 1]);
                if (first) { np++; no += !terminal(src[x]); }
            }
            *all_p += np; *open_p += no;
        };
        count_pieces(&pieces_r[0], &open_r[0]);
        uint32_t r = 0;
        for (;;) {
            bool any = false;
            std::vector<int64_t> nsrc(src);
            for (size_t x = 0; x < N; x++) {
                if (!is_match[x] || tail[x]) continue;
                const int64_t s = src[x];
                if (terminal(s)) continue;
                nsrc[x] = fold(src[(size_t)(s - (int64_t)OP)]);
                any = true;
            }
            if (!any) break;
            // cuts: pieces before the round whose bytes no longer form one run afterwards
            uint64_t before_all = 0, before_open = 0; count_pieces(&before_all, &before_open);
            src.swap(nsrc);
            r++;
            uint64_t after_all = 0, after_open = 0; count_pieces(&after_all, &after_open);
            if (r < 16) { pieces_r[r] += after_all; open_r[r] += after_open; cuts_r[r] += after_all > before_all ? after_all - before_all : 0; }
            if (r >= 15) break;
        }
        for (uint32_t k = r + 1; k < 16; k++) { uint64_t a = 0, o = 0; count_pieces(&a, &o); pieces_r[k] += a; }
        rounds_sum += r; rounds_max = std::max<uint64_t>(rounds_max, r);
        batches++; seqs += cnt;
        b0 += cnt;
    }
    printf("%s: %zu sequences, %llu bytes, %llu batches of <= %u sequences / %u bytes\n", argv[1], all.size(), (unsigned long long)op,
           (unsigned long long)batches, BN, WNEW);
    if (!batches) return 0;
    printf("  per batch: %.0f sequences, %.0f matches (%.0f bytes), periodic matches %.1f (their tails %.0f bytes)\n", (double)seqs / batches,
           (double)matches / batches, (double)mbytes / batches, (double)periodic / batches, (double)periodic_bytes / batches);
    printf("  dependency depth (byte hops = levels the polling copy phase walks without relinking): mean %.1f, max %llu\n",
           (double)levels_sum / batches, (unsigned long long)levels_max);
    printf("  pointer-doubling rounds until every source is terminal: mean %.2f, max %llu\n", (double)rounds_sum / batches, (unsigned long long)rounds_max);
    printf("  pieces per batch (all / still open / cuts made in the round):\n");
    for (uint32_t k = 0; k <= std::min<uint64_t>(rounds_max, 15); k++)
        printf("    after round %2u: %8.0f / %8.0f / %8.0f   (x %.2f of the matches)\n", k, (double)pieces_r[k] / batches, (double)open_r[k] / batches,
               (double)cuts_r[k] / batches, (double)pieces_r[k] / std::max<uint64_t>(matches, 1));
    return 0;
}
