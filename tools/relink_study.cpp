// study (CPU, no GPU): dependency depth of the matches of a workgroup-decoder batch (2 048 sequences / 48 KiB), before and after
// RELINKING sources through the matches that produce them (lz4_decompress_pcd.hip match_slot), and the share of
// matches that relink in each round.  g++ -O2 -o /tmp/relink tools/relink_study.cpp; /tmp/relink block.lz4 [batch [rounds]]
// (block.lz4: one raw LZ4 block, e.g. written by tests/wave_model.compress or oracle_api.compress)
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
struct Seq { uint64_t start, ms; uint32_t lit, ml, off; };
int main(int argc, char** argv) {
    FILE* f = fopen(argv[1], "rb"); std::vector<uint8_t> c(1 << 25); size_t n = fread(c.data(), 1, c.size(), f); fclose(f);
    const uint32_t BN = argc > 2 ? atoi(argv[2]) : 2048, WNEW = 49152, ROUNDS = argc > 3 ? atoi(argv[3]) : 8;
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
    printf("%s: %zu sequences, %llu bytes\n", argv[1], all.size(), (unsigned long long)op);
    uint64_t batches = 0, depth_sum = 0, depth2_sum = 0, nmatch = 0, free0 = 0, free1 = 0, relinked = 0; uint32_t dmax = 0, d2max = 0;
    uint64_t hist_lv[8] = {0}, hist_lv2[8] = {0}, relinked_r[8] = {0};
    for (size_t b0 = 0; b0 < all.size();) {
        size_t cnt = 0; uint64_t OP = all[b0].start;
        while (b0 + cnt < all.size() && cnt < BN && all[b0 + cnt].start + all[b0 + cnt].lit + all[b0 + cnt].ml - OP <= WNEW) cnt++;
        if (cnt == 0) { b0++; continue; }
        const Seq* q = &all[b0];
        std::vector<uint64_t> start(cnt + 1);
        for (size_t i = 0; i < cnt; i++) start[i] = q[i].start;
        start[cnt] = q[cnt - 1].ms + q[cnt - 1].ml;
        auto producer = [&](uint64_t pos) { return (uint32_t)(std::upper_bound(start.begin(), start.begin() + cnt, pos) - start.begin()) - 1; };
        // level with sources R[i]
        auto levels = [&](const std::vector<uint64_t>& R, uint64_t* hist, uint64_t* nfree) {
            std::vector<uint32_t> level(cnt, 0); uint32_t dm = 0;
            for (size_t i = 0; i < cnt; i++) {
                if (!q[i].ml) continue;
                const uint64_t s0 = R[i], s1 = std::min<uint64_t>(s0 + q[i].ml, q[i].ms);
                uint32_t lv = 0;
                if (s1 > OP) {
                    uint32_t l = producer(std::max(s0, OP)), h = producer(s1 - 1);
                    for (uint32_t j = l; j <= h && j < i; j++) {
                        // does the range touch j's match part?
                        const uint64_t a = std::max<uint64_t>(s0, q[j].ms), e = std::min<uint64_t>(s1, q[j].ms + q[j].ml);
                        if (q[j].ml && a < e) lv = std::max(lv, level[j]);
                    }
                }
                level[i] = lv + 1; dm = std::max(dm, level[i]);
                hist[std::min<uint32_t>(level[i] - 1, 7)]++;
                if (level[i] == 1) (*nfree)++;
            }
            return dm;
        };
        std::vector<uint64_t> R(cnt);
        for (size_t i = 0; i < cnt; i++) { R[i] = q[i].ms - q[i].off; nmatch += q[i].ml != 0; }
        uint32_t d1 = levels(R, hist_lv, &free0);
        // relink rounds (all matches in parallel: uses the previous round's R)
        for (uint32_t r = 0; r < ROUNDS; r++) {
            std::vector<uint64_t> Rn = R; bool any = false;
            for (size_t i = 0; i < cnt; i++) {
                if (!q[i].ml || q[i].off < q[i].ml) continue;                 // (self-overlapping: left alone)
                if (R[i] < OP) continue;
                const uint32_t j = producer(R[i]);
                if (j >= i || !q[j].ml) continue;
                if (R[i] >= q[j].ms && R[i] + q[i].ml <= q[j].ms + q[j].ml) { Rn[i] = R[j] + (R[i] - q[j].ms); any = true; relinked_r[r < 8 ? r : 7]++; }
            }
            R = Rn; if (!any) break;
        }
        uint32_t d2 = levels(R, hist_lv2, &free1);
        depth_sum += d1; depth2_sum += d2; dmax = std::max(dmax, d1); d2max = std::max(d2max, d2); batches++;
        b0 += cnt;
    }
    printf("  batches %llu (%.0f seqs), matches %llu; depth mean %.1f max %u -> after relinking mean %.1f max %u; free matches %.1f%% -> %.1f%%; relinked in round 1: %.1f%%\n",
           (unsigned long long)batches, (double)all.size() / batches, (unsigned long long)nmatch, (double)depth_sum / batches, dmax, (double)depth2_sum / batches, d2max,
           100.0 * free0 / nmatch, 100.0 * free1 / nmatch, 100.0 * relinked_r[0] / nmatch);
    printf("  relinked per round (%% of matches):"); for (int i = 0; i < 8; i++) printf(" %.1f", 100.0 * relinked_r[i] / nmatch); printf("\n"); printf("  level histogram before:"); for (int i = 0; i < 8; i++) printf(" %llu", (unsigned long long)hist_lv[i]); printf("\n  after: "); for (int i = 0; i < 8; i++) printf(" %llu", (unsigned long long)hist_lv2[i]); printf("\n");
}
