/*
 * lz4flex_bench.c -- ORACLE side helper for bench.py's `cpu_baseline` leg only: runs the
 * oracle's block compress / decompress over a batch of independent blocks on `threads`
 * pthreads (static contiguous partition) and returns the best wall time of `reps` passes.
 * Test/measurement infrastructure; never part of the product path.
 */
#define _GNU_SOURCE
#include "lz4flex_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <time.h>

/* C liblz4 1.9.3 entry points (LZ4_compress_default / LZ4_decompress_safe), handed in as pointers by bench.py (the
 * library is loaded with ctypes; no header is needed): the reference's own cross-implementation anchor
 * (tests/tests.rs:25-56), timed beside the port as SURVEY 8(d) prescribes */
typedef int (*lz4_fn)(const char *src, char *dst, int src_size, int dst_cap);

typedef struct {
    int dir;
    lz4_fn fn;
    const uint8_t *in_base; const uint64_t *in_off; const uint32_t *in_len;
    uint8_t *out_base; const uint64_t *out_off; const uint32_t *out_cap; uint32_t *out_len;
    uint32_t begin, end;
} job_t;

static void *worker(void *arg) {
    job_t *j = (job_t *)arg;
    for (uint32_t i = j->begin; i < j->end; i++) {
        int64_t r;
        if (j->fn)
            r = j->fn((const char *)(j->in_base + j->in_off[i]), (char *)(j->out_base + j->out_off[i]), (int)j->in_len[i],
                      (int)j->out_cap[i]);
        else if (j->dir == 0)
            r = lz4o_compress_into(j->in_base + j->in_off[i], j->in_len[i], j->out_base + j->out_off[i], j->out_cap[i]);
        else
            r = lz4o_decompress_into(j->in_base + j->in_off[i], j->in_len[i], j->out_base + j->out_off[i],
                                     j->out_cap[i], NULL);
        j->out_len[i] = r < 0 ? 0xFFFFFFFFu : (uint32_t)r;
    }
    return NULL;
}

static double now_s(void) {
    struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

double lz4o_bench_batch_fn(int dir, void *fn, const uint8_t *in_base, const uint64_t *in_off, const uint32_t *in_len,
                           uint8_t *out_base, const uint64_t *out_off, const uint32_t *out_cap,
                           uint32_t *out_len, uint32_t n_blocks, int threads, int reps) {
    if (threads < 1) threads = 1;
    if ((uint32_t)threads > n_blocks && n_blocks > 0) threads = (int)n_blocks;
    pthread_t *tid = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
    job_t *jobs = (job_t *)malloc(sizeof(job_t) * (size_t)threads);
    double best = 1e30;
    for (int r = 0; r < reps; r++) {
        double t0 = now_s();
        for (int t = 0; t < threads; t++) {
            job_t *j = &jobs[t];
            j->dir = dir; j->fn = (lz4_fn)fn; j->in_base = in_base; j->in_off = in_off; j->in_len = in_len;
            j->out_base = out_base; j->out_off = out_off; j->out_cap = out_cap; j->out_len = out_len;
            j->begin = (uint32_t)((uint64_t)n_blocks * (uint64_t)t / (uint64_t)threads);
            j->end = (uint32_t)((uint64_t)n_blocks * (uint64_t)(t + 1) / (uint64_t)threads);
            if (threads == 1) worker(j); else pthread_create(&tid[t], NULL, worker, j);
        }
        if (threads > 1) for (int t = 0; t < threads; t++) pthread_join(tid[t], NULL);
        double dt = now_s() - t0;
        if (dt < best) best = dt;
    }
    free(tid); free(jobs);
    return best;
}

double lz4o_bench_batch(int dir, const uint8_t *in_base, const uint64_t *in_off, const uint32_t *in_len,
                        uint8_t *out_base, const uint64_t *out_off, const uint32_t *out_cap,
                        uint32_t *out_len, uint32_t n_blocks, int threads, int reps) {
    return lz4o_bench_batch_fn(dir, NULL, in_base, in_off, in_len, out_base, out_off, out_cap, out_len, n_blocks, threads, reps);
}
