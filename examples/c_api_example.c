/* Plain-C caller of the C ABI (include/dann.h): what a cgo / Rust-FFI / JNI binding does, without Python.
 * Builds a small index on the GPU, searches it, and checks the answers against brute force.
 *   gcc -std=c99 -O2 -Iinclude examples/c_api_example.c -Ldiskann_amd -ldann_hip -Wl,-rpath,$PWD/diskann_amd -lm -o c_api_example
 * Exit code 0 = every query's nearest neighbour (and recall@10 >= 0.9) confirmed. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "dann.h"

#define CHECK(call)                                                        \
    do {                                                                   \
        int32_t rc_ = (call);                                              \
        if (rc_ < 0) {                                                     \
            char msg[512];                                                 \
            dann_last_error(msg, sizeof msg);                              \
            fprintf(stderr, "%s failed: %d (%s)\n", #call, (int)rc_, msg); \
            return 1;                                                      \
        }                                                                  \
    } while (0)

static uint32_t rng_state = 12345u;
static float frand(void) { /* xorshift, uniform in [0, 1) */
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 17;
    rng_state ^= rng_state << 5;
    return (float)(rng_state >> 8) / 16777216.0f;
}

int main(void) {
    enum { N = 20000, DIM = 32, NQ = 200, K = 10, R = 32, CLUSTERS = 64 };
    float* centers = malloc(sizeof(float) * CLUSTERS * DIM);
    float* base = malloc(sizeof(float) * (size_t)N * DIM);
    float* queries = malloc(sizeof(float) * (size_t)NQ * DIM);
    for (int i = 0; i < CLUSTERS * DIM; ++i) centers[i] = frand();
    for (int i = 0; i < N + NQ; ++i) {
        float* dst = i < N ? base + (size_t)i * DIM : queries + (size_t)(i - N) * DIM;
        const float* c = centers + (size_t)(rng_state % CLUSTERS) * DIM;
        for (int d = 0; d < DIM; ++d) dst[d] = c[d] + 0.1f * (frand() - 0.5f);
    }
    /* medoid-like start point: the row closest to the mean */
    double mean[DIM] = {0};
    for (int i = 0; i < N; ++i)
        for (int d = 0; d < DIM; ++d) mean[d] += base[(size_t)i * DIM + d];
    int medoid = 0;
    double best = 1e300;
    for (int i = 0; i < N; ++i) {
        double s = 0;
        for (int d = 0; d < DIM; ++d) {
            double t = base[(size_t)i * DIM + d] - mean[d] / N;
            s += t * t;
        }
        if (s < best) best = s, medoid = i;
    }

    dann_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.dtype = DANN_F32;
    cfg.metric = DANN_L2;
    cfg.dim = DIM;
    cfg.capacity = N;
    cfg.max_degree = R;
    cfg.num_start_points = 1;
    cfg.device = -1; /* current device */
    dann_index* idx = NULL;
    CHECK(dann_index_create(&cfg, base + (size_t)medoid * DIM, sizeof(float) * DIM, &idx));
    CHECK(dann_set_elements(idx, 0, N, base, sizeof(float) * (size_t)N * DIM));

    dann_build_config bc;
    memset(&bc, 0, sizeof bc);
    bc.pruned_degree = 28;
    bc.max_degree = R;
    bc.l_build = 64;
    bc.alpha = 1.2f;
    bc.max_occlusion_size = 750;
    bc.max_backedges = 28;
    bc.intra_batch_candidates = 0;
    bc.saturate_after_prune = 0;
    CHECK(dann_build(idx, &bc, 0, N, 0.02f, 4096));

    uint32_t* ids = malloc(sizeof(uint32_t) * NQ * K);
    float* dists = malloc(sizeof(float) * NQ * K);
    dann_search_stats* stats = malloc(sizeof(dann_search_stats) * NQ);
    CHECK(dann_search_batch(idx, queries, NQ, 48, 1, K, ids, dists, stats));

    /* brute force: exact top-K per query */
    int top1_ok = 0;
    double recall = 0;
    for (int q = 0; q < NQ; ++q) {
        uint32_t gt[K];
        float gd[K];
        int n = 0;
        for (int i = 0; i < N; ++i) {
            float s = 0;
            for (int d = 0; d < DIM; ++d) {
                float t = queries[(size_t)q * DIM + d] - base[(size_t)i * DIM + d];
                s += t * t;
            }
            int pos = n < K ? n : K;
            while (pos > 0 && gd[pos - 1] > s) --pos;
            if (pos < K) {
                int last = n < K ? n : K - 1;
                for (int j = last; j > pos; --j) gd[j] = gd[j - 1], gt[j] = gt[j - 1];
                gd[pos] = s, gt[pos] = (uint32_t)i;
                if (n < K) ++n;
            }
        }
        top1_ok += ids[q * K] == gt[0] || fabsf(dists[q * K] - gd[0]) <= 1e-4f * gd[0];
        int hit = 0;
        for (int a = 0; a < K; ++a)
            for (int b = 0; b < K; ++b) hit += ids[q * K + a] == gt[b];
        recall += (double)hit / K;
    }
    recall /= NQ;
    printf("c_api_example: %d points, %d queries, top-1 agreement %d/%d, recall@10 %.4f, cmps/query %.1f\n", N, NQ,
           top1_ok, NQ, recall, (double)stats[0].cmps);
    CHECK(dann_index_destroy(idx));
    return (top1_ok >= NQ * 9 / 10 && recall >= 0.9) ? 0 : 2;
}
