/*
 * oracle/knn.c -- TEST INFRASTRUCTURE (see oracle.h).  CPU restatement of the
 * nearest-neighbour / Chamfer arithmetic the reference reaches through
 * chamferdist._C (utils/chamfer.py:174,206) and knn_cuda.KNN
 * (run_robot.py:65-66, utils/flow_utils.py:158, utils/model_utils.py:42).
 *
 * PARITY UNPINNED for tie / rounding rules: both packages are third-party and
 * not vendored (setup_env.sh:2-7).  Contract used here and by the HIP kernels:
 *   d(i,j) = ((dx*dx) + (dy*dy)) + (dz*dz) in fp32 (general D: left-to-right
 *   running sum starting from the first term), no FMA, strict '<' while
 *   scanning j ascending (=> ties go to the lowest j; K>1 is a stable
 *   ascending order by (d, j)).
 */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

static inline float sqdist(const float *a, const float *b, int D) {
    float dx = a[0] - b[0];
    float s = dx * dx;
    for (int c = 1; c < D; ++c) {
        float dc = a[c] - b[c];
        s = s + dc * dc;
    }
    return s;
}

#define QB 16 /* queries handled together so gcc can vectorise across them */

/* K == 1, D == 3 fast path: identical arithmetic, laid out for SIMD. */
static void knn1_d3(const float *p1, const float *p2, int n1, int n2,
                    float *dists, int64_t *idx) {
    for (int i0 = 0; i0 < n1; i0 += QB) {
        float qx[QB], qy[QB], qz[QB], best[QB];
        int bi[QB];
        int nb = n1 - i0 < QB ? n1 - i0 : QB;
        for (int q = 0; q < QB; ++q) {
            int i = i0 + (q < nb ? q : 0);
            qx[q] = p1[3 * i + 0];
            qy[q] = p1[3 * i + 1];
            qz[q] = p1[3 * i + 2];
            best[q] = INFINITY;
            bi[q] = 0;
        }
        for (int j = 0; j < n2; ++j) {
            const float tx = p2[3 * j + 0], ty = p2[3 * j + 1], tz = p2[3 * j + 2];
            for (int q = 0; q < QB; ++q) {
                float dx = qx[q] - tx, dy = qy[q] - ty, dz = qz[q] - tz;
                float d = (dx * dx + dy * dy) + dz * dz;
                int lt = d < best[q];
                best[q] = lt ? d : best[q];
                bi[q] = lt ? j : bi[q];
            }
        }
        for (int q = 0; q < nb; ++q) {
            dists[i0 + q] = best[q];
            idx[i0 + q] = bi[q];
        }
    }
}

/* utils/chamfer.py:140-193: dists (N,P1,K) squared, idx (N,P1,K) int64, rows
 * beyond lengths1 and slots beyond lengths2 are zero (docstring :163-170). */
void oracle_knn_points(const float *p1, const float *p2,
                       const int64_t *lengths1, const int64_t *lengths2,
                       int N, int P1, int P2, int D, int K,
                       float *dists, int64_t *idx) {
    memset(dists, 0, sizeof(float) * (size_t)N * P1 * K);
    memset(idx, 0, sizeof(int64_t) * (size_t)N * P1 * K);
#pragma omp parallel for schedule(dynamic, 1)
    for (int w = 0; w < N * ((P1 + 255) / 256); ++w) {
        int n = w / ((P1 + 255) / 256);
        int c0 = (w % ((P1 + 255) / 256)) * 256;
        int n1 = lengths1 ? (int)lengths1[n] : P1;
        int n2 = lengths2 ? (int)lengths2[n] : P2;
        int c1 = c0 + 256 < n1 ? c0 + 256 : n1;
        if (c0 >= c1) continue;
        const float *a = p1 + (size_t)n * P1 * D;
        const float *b = p2 + (size_t)n * P2 * D;
        float *dn = dists + (size_t)n * P1 * K;
        int64_t *in = idx + (size_t)n * P1 * K;
        if (K == 1 && D == 3) {
            if (n2 > 0) knn1_d3(a + 3 * c0, b, c1 - c0, n2, dn + c0, in + c0);
            continue;
        }
        float *bd = (float *)malloc(sizeof(float) * K);
        int64_t *bj = (int64_t *)malloc(sizeof(int64_t) * K);
        for (int i = c0; i < c1; ++i) {
            int cnt = 0;
            for (int j = 0; j < n2; ++j) {
                float d = sqdist(a + (size_t)i * D, b + (size_t)j * D, D);
                if (cnt < K) { /* fill, keeping (d, j) ascending, stable */
                    int s = cnt++;
                    while (s > 0 && d < bd[s - 1]) { bd[s] = bd[s - 1]; bj[s] = bj[s - 1]; --s; }
                    bd[s] = d; bj[s] = j;
                } else if (d < bd[K - 1]) {
                    int s = K - 1;
                    while (s > 0 && d < bd[s - 1]) { bd[s] = bd[s - 1]; bj[s] = bj[s - 1]; --s; }
                    bd[s] = d; bj[s] = j;
                }
            }
            for (int k = 0; k < cnt; ++k) {
                dn[(size_t)i * K + k] = bd[k];
                in[(size_t)i * K + k] = bj[k];
            }
        }
        free(bd); free(bj);
    }
}

/* utils/chamfer.py:195-209.  grad_p1[n,i] += 2 g (p1[n,i]-p2[n,idx]),
 * grad_p2[n,idx] -= same; accumulated in (i, k) ascending order. */
void oracle_knn_points_backward(const float *p1, const float *p2,
                                const int64_t *lengths1, const int64_t *lengths2,
                                const int64_t *idx, const float *grad_dists,
                                int N, int P1, int P2, int D, int K,
                                float *grad_p1, float *grad_p2) {
    memset(grad_p1, 0, sizeof(float) * (size_t)N * P1 * D);
    memset(grad_p2, 0, sizeof(float) * (size_t)N * P2 * D);
    for (int n = 0; n < N; ++n) {
        int n1 = lengths1 ? (int)lengths1[n] : P1;
        int n2 = lengths2 ? (int)lengths2[n] : P2;
        int kk = K < n2 ? K : n2;
        for (int i = 0; i < n1; ++i)
            for (int k = 0; k < kk; ++k) {
                int64_t j = idx[((size_t)n * P1 + i) * K + k];
                float g = grad_dists[((size_t)n * P1 + i) * K + k];
                for (int c = 0; c < D; ++c) {
                    float diff = p1[((size_t)n * P1 + i) * D + c] - p2[((size_t)n * P2 + j) * D + c];
                    float v = (2.0f * g) * diff;
                    grad_p1[((size_t)n * P1 + i) * D + c] += v;
                    grad_p2[((size_t)n * P2 + j) * D + c] -= v;
                }
            }
    }
}

/* knn_cuda.KNN(k, transpose_mode=True): ref [B,nr,D], query [B,nq,D] ->
 * dist [B,nq,k] ascending, idx [B,nq,k] int64 (shape comment
 * utils/model_utils.py:42).  `euclidean` != 0 returns sqrt of the squared
 * distance (upstream KNN_CUDA behaviour); ordering is always by (d^2, j). */
void oracle_knn_cuda(const float *ref, const float *query, int B, int nr, int nq,
                     int D, int k, int euclidean, float *dist, int64_t *idx) {
    oracle_knn_points(query, ref, NULL, NULL, B, nq, nr, D, k, dist, idx);
    if (euclidean) {
        size_t tot = (size_t)B * nq * k;
        for (size_t t = 0; t < tot; ++t) dist[t] = sqrtf(dist[t]);
    }
}
