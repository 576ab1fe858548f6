/*
 * oracle/flow.c -- TEST INFRASTRUCTURE (see oracle.h).  CPU restatement of
 *   blend_anchor_motion   utils/flow_utils.py:147-170
 *   flow_loss             networks/loss.py:10-21
 * PINNED by tests/golden/flow.npz (reference Python run with the oracle's KNN as the
 * knn_cuda stand-in; the KNN itself is the UNPINNED contract of oracle/knn.c).
 */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* utils/flow_utils.py:147-170 for one frame pair.  query [nq,3], ref [nr,3],
 * ref_flow [nr,3] -> flow [nq,3], mask [nq] (uint8 0/1).
 *   d,idx = KNN_k(ref, query)  (Euclidean when `euclidean`)          :158
 *   d[d < 1e-10] = 1e-10; w = 1/d; w /= sum_k w                     :160-162
 *   flow = sum_k w_k ref_flow[idx_k]                                 :163
 *   mask = min_k d <= max_k |ref_flow[idx_k]|^2  or  min_k d <= 0.05 :165-167 */
void oracle_blend_anchor_motion(const float *query, const float *ref,
                                const float *ref_flow, int nq, int nr, int k,
                                int euclidean, float *flow, uint8_t *mask) {
    float *dist = (float *)malloc(sizeof(float) * (size_t)nq * k);
    int64_t *idx = (int64_t *)malloc(sizeof(int64_t) * (size_t)nq * k);
    oracle_knn_cuda(ref, query, 1, nr, nq, 3, k, euclidean, dist, idx);
    for (int n = 0; n < nq; ++n) {
        float w[64];
        float wsum = 0.f, dmin = INFINITY, fmax = -INFINITY;
        for (int j = 0; j < k; ++j) {
            float d = dist[(size_t)n * k + j];
            if (d < 1e-10f) d = 1e-10f;
            w[j] = 1.0f / d;
            wsum += w[j];
            dmin = fminf(dmin, d);
            const float *f = ref_flow + 3 * idx[(size_t)n * k + j];
            const float fn = (f[0] * f[0] + f[1] * f[1]) + f[2] * f[2];
            fmax = fmaxf(fmax, fn);
        }
        float acc[3] = {0.f, 0.f, 0.f};
        for (int j = 0; j < k; ++j) {
            const float wn = w[j] / wsum;
            const float *f = ref_flow + 3 * idx[(size_t)n * k + j];
            for (int c = 0; c < 3; ++c) acc[c] += f[c] * wn;
        }
        memcpy(flow + 3 * (size_t)n, acc, 12);
        mask[n] = (dmin <= fmax) || (dmin <= 0.05f);
    }
    free(dist); free(idx);
}

static inline float huber1(float x) { /* F.huber_loss, delta = 1 */
    const float a = fabsf(x);
    return a <= 1.0f ? 0.5f * x * x : (a - 0.5f);
}
static inline float huber1_grad(float x) {
    return fabsf(x) <= 1.0f ? x : (x > 0.f ? 1.0f : -1.0f);
}

/* networks/loss.py:10-21: sum_{b,n} [ m f + smooth (not m) |pred|^2 ],
 * f = sum_c (pred-gt)^2, or Huber(delta=1) when robust.  mask NULL = all ones.
 * Returns the loss (double accumulation); grad_pred (nullable) = d loss / d pred. */
double oracle_flow_loss(const float *gt, const float *pred, const uint8_t *mask,
                        int B, int N, int robust, float smooth_weight,
                        float *grad_pred) {
    double total = 0.0;
    for (size_t e = 0; e < (size_t)B * N; ++e) {
        const int m = mask ? (mask[e] != 0) : 1;
        float f = 0.f, sm = 0.f;
        for (int c = 0; c < 3; ++c) {
            const float p = pred[3 * e + c], d = p - gt[3 * e + c];
            f += robust ? huber1(d) : d * d;
            sm += p * p;
            if (grad_pred) {
                const float gf = robust ? huber1_grad(d) : 2.0f * d;
                grad_pred[3 * e + c] = m ? gf : smooth_weight * (2.0f * p);
            }
        }
        total += m ? (double)f : (double)(smooth_weight * sm);
    }
    return total;
}
