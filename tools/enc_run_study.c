// study (CPU, no GPU): how much of a head's true match length is already verified by the run of positions behind it that share
// its distance and match their own 4 bytes (half of it -- the compare rounds stay)?  Uses the encoder model's
// index pass.  gcc -O2 -o /tmp/run_study tools/enc_run_study.c; /tmp/run_study plain_input_file
#include <stdio.h>
#include "../tests/sim/wave_encoder_model.c"
int main(int argc, char** argv) {
    FILE* f = fopen(argv[1], "rb"); static uint8_t buf[1 << 24]; uint32_t n = fread(buf, 1, sizeof buf, f); fclose(f);
    if (n > 65536) n = 65536;
    uint16_t* d = calloc(n + 64, 2);
    lz4w_index(buf, n, d, 0);
    // per position: eq
    uint8_t* eq = calloc(n + 64, 1);
    for (uint32_t p = 0; p + 12 <= n; p++) eq[p] = d[p] && p >= d[p] && ld32(buf + p) == ld32(buf + p - d[p]);
    uint32_t heads = 0, exact_known = 0, hist[8] = {0}, rounds_old_sum = 0, supersteps = 0;
    uint32_t needB_steps = 0, needA_steps = 0; double sumL = 0, sumLv = 0;
    uint32_t rB_sum = 0, rOld_sum = 0;
    for (uint32_t b = 0; b < n; b += 256) {
        uint32_t maxOld = 0, maxB = 0, anyA = 0, h = 0;
        for (uint32_t p = b; p < b + 256 && p + 12 <= n; p++) {
            if (!d[p] || (p > 0 && d[p] == d[p - 1]) || !eq[p]) continue;
            heads++; h++;
            uint32_t lim = n - 5 - p; if (lim > 1024) lim = 1024;
            uint32_t L = 0; while (L < lim && buf[p + L] == buf[p - d[p] + L]) L++;
            // run
            uint32_t j = p + 1;
            while (j < b + 256 + 64 && j + 12 <= n && d[j] == d[p] && eq[j]) j++;
            uint32_t Lv = (j - p) + 3; if (Lv > lim) Lv = lim;
            int exact = (j + 12 <= n && d[j] == d[p] && !eq[j]);
            sumL += L; sumLv += Lv;
            uint32_t rem = L - Lv;      // bytes still to be discovered (plus the terminating mismatch)
            if (exact) { exact_known++; continue; }
            hist[rem < 7 ? rem : 7]++;
            anyA = 1;
            // old scheme rounds: 16 then 32...; first round covers 4..20
            uint32_t ro = 1; if (L >= 20) ro += 1 + (L - 20) / 32; if (ro > maxOld) maxOld = ro;
            // new: round A 4 bytes at Lv (covers Lv..Lv+4), then 16, then 32s
            uint32_t rb = 0; if (rem >= 4 && Lv + 4 < lim) { rb = 1; if (rem >= 20) rb += 1 + (rem - 20) / 32; } if (rb > maxB) maxB = rb;
        }
        if (h) { supersteps++; rOld_sum += maxOld; rB_sum += maxB; needA_steps += anyA; }
    }
    printf("%s: heads %u (%.1f per 256), exact-known by run end %u (%.1f%%), mean L %.1f mean verified %.1f\n", argv[1], heads, heads * 256.0 / n, exact_known, 100.0 * exact_known / heads, sumL / heads, sumLv / heads);
    printf("  remaining bytes beyond verified (non-exact heads): "); for (int i = 0; i < 8; i++) printf("%d:%u ", i, hist[i]); printf("\n");
    printf("  per superstep: old rounds (max over heads) %.2f ; new: round A needed in %.0f%%, rounds beyond A %.2f\n", (double)rOld_sum / supersteps, 100.0 * needA_steps / supersteps, (double)rB_sum / supersteps);
    return 0;
}
