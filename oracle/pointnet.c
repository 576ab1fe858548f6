/*
 * oracle/pointnet.c -- TEST INFRASTRUCTURE (see oracle.h).  CPU restatement of the two live
 * native ops of the reference's PointNet++ path, in both of the reference's semantics:
 *
 *   farthest point sampling
 *     cuda_mode = 0 : CPU fallback  networks/pointnet2_utils.py:88-99   (start index injected,
 *                     arg-max = first maximum, as torch.max(distance,-1)[1])
 *     cuda_mode = 1 : CUDA kernel   networks/pointnet_lib/src/sampling_gpu.cu:93-209
 *                     (start 0 unless injected; block tree arg-max: ties go to the lowest
 *                     thread id = k mod block_size, then the lowest k; block_size =
 *                     opt_n_threads(N), cuda_utils.h:10-14)
 *   ball query
 *     cuda_mode = 0 : CPU fallback  networks/pointnet2_utils.py:102-140 (d2 <= r^2 with
 *                     r^2 = float32(double(r)^2), first nsample in index order, padded with the
 *                     NEAREST point = first minimum) on the reference's matmul-expanded
 *                     square_distance(new_xyz, xyz) with torch's CPU rounding (see the 3-NN section
 *                     below): EVERY row of the golden is bit-equal, boundary rows included
 *                     (`margin` = min |d2-r2|/r2 per row is still reported)
 *     cuda_mode = 1 : CUDA kernel   networks/pointnet_lib/src/ball_query_gpu.cu:9-45 (d2 < r*r in
 *                     fp32, padded with the FIRST hit, zero when there is none)
 *
 * PINNED by tests/golden/pointnet_ops.npz (reference CPU fallbacks run with the start index
 * recorded).  Distance order: ((dx*dx)+(dy*dy))+(dz*dz), no FMA.
 */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

static inline float sqd3(const float *a, const float *b) {
    float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
    return (dx * dx + dy * dy) + dz * dz;
}

static int opt_n_threads(int work) { /* cuda_utils.h:10-14 */
    int p = 1;
    while (p * 2 <= work) p *= 2;
    if (p > 1024) p = 1024;
    if (p < 1) p = 1;
    return p;
}

void oracle_fps(const float *xyz, int B, int N, int npoint, const int32_t *start,
                int cuda_mode, int64_t *idx) {
    float *dm = (float *)malloc(sizeof(float) * N);
    const int bs = opt_n_threads(N);
    for (int b = 0; b < B; ++b) {
        const float *p = xyz + (size_t)b * N * 3;
        for (int k = 0; k < N; ++k) dm[k] = 1e10f;
        int far = start ? start[b] : 0;
        for (int i = 0; i < npoint; ++i) {
            idx[(size_t)b * npoint + i] = far;
            if (i == npoint - 1) break;
            float best = -1.0f;
            int besti = 0;
            for (int k = 0; k < N; ++k) {
                const float d = sqd3(p + 3 * k, p + 3 * far);
                if (d < dm[k]) dm[k] = d;
                const float v = dm[k];
                int take;
                if (!cuda_mode) take = v > best;                     /* first maximum */
                else take = (v > best) || (v == best && (k % bs) < (besti % bs));
                if (take) { best = v; besti = k; }
            }
            far = besti;
        }
    }
    free(dm);
}

float oracle_square_distance_pair(const float *src, const float *dst);

void oracle_ball_query(const float *xyz, const float *new_xyz, int B, int N, int S,
                       double radius, int nsample, int cuda_mode, int64_t *idx, float *margin) {
    const float r2 = cuda_mode ? (float)radius * (float)radius : (float)(radius * radius);
    for (int b = 0; b < B; ++b)
        for (int s = 0; s < S; ++s) {
            const float *q = new_xyz + 3 * ((size_t)b * S + s);
            int64_t *o = idx + ((size_t)b * S + s) * nsample;
            int cnt = 0, nearest = 0;
            float dn = INFINITY, mg = INFINITY;
            for (int l = 0; l < nsample; ++l) o[l] = 0;
            for (int k = 0; k < N; ++k) {
                /* CPU fallback: square_distance(new_xyz, xyz), the matmul expansion (:126), bit for bit;
                 * CUDA kernel: coordinate differences (ball_query_gpu.cu:30-33) */
                const float d = cuda_mode ? sqd3(q, xyz + 3 * ((size_t)b * N + k))
                                          : oracle_square_distance_pair(q, xyz + 3 * ((size_t)b * N + k));
                if (d < dn) { dn = d; nearest = k; }
                const float m = fabsf(d - r2) / r2;
                if (m < mg) mg = m;
                const int hit = cuda_mode ? (d < r2) : (d <= r2);
                if (hit && cnt < nsample) o[cnt++] = k;
            }
            const int pad = cuda_mode ? (cnt > 0 ? (int)o[0] : 0) : nearest;
            for (int l = cnt; l < nsample; ++l) o[l] = pad;
            if (margin) margin[(size_t)b * S + s] = mg;
        }
}

/* ------------------------------------------------------------------------------------------------
 * PointNetFeaturePropagation's 3-NN inverse-distance interpolation, networks/pointnet2_utils.py:326-336,
 * on top of square_distance (:33-55):
 *     dist  = -2 * matmul(src, dst^T);  dist += sum(src**2,-1);  dist += sum(dst**2,-1)
 *     dists, idx = dist.sort(-1)[:3];   w = 1/(dists + 1e-8);  w /= sum(w);  out = sum_k w_k points2[idx_k]
 * Rounding contract, PINNED bit for bit against the reference run on this container's CPU
 * (tests/golden/three_interp.npz, tests/golden/make_golden_parity2.py):
 *     matmul over K = 3  = fma(az, bz, fma(ay, by, ax*bx))      (torch CPU sgemm, probed)
 *     sum(p**2, -1)      = ((x*x) + (y*y)) + (z*z)
 *     d                  = ((-2*mm) + |src|^2) + |dst|^2        (NOT a difference of coordinates: coincident
 *                          points give d = +-1e-7 noise and even negative values, which 1/(d+1e-8) amplifies;
 *                          a direct-difference distance changes the descriptors by ~1e-3 of their scale)
 *     order              = ascending (d, index): stable
 *     norm = (w0 + w1) + w2,  out_c = ((p0_c*w0) + (p1_c*w1)) + (p2_c*w2)
 * ---------------------------------------------------------------------------------------------- */
__attribute__((target("fma"))) static inline float dot3_fma(const float *a, const float *b) {
    return __builtin_fmaf(a[2], b[2], __builtin_fmaf(a[1], b[1], a[0] * b[0]));
}

static inline float sumsq3(const float *a) { return (a[0] * a[0] + a[1] * a[1]) + a[2] * a[2]; }

__attribute__((target("fma"))) float oracle_square_distance_pair(const float *src, const float *dst) {
    return ((-2.0f * dot3_fma(src, dst)) + sumsq3(src)) + sumsq3(dst);
}

__attribute__((target("fma")))
void oracle_three_nn_expanded(const float *xyz1, const float *xyz2, int B, int N, int S,
                              float *dist3, int64_t *idx3) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int n = 0; n < N; ++n) {
            const float *q = xyz1 + 3 * ((size_t)b * N + n);
            const float sq = sumsq3(q);
            float bd[3] = {INFINITY, INFINITY, INFINITY};
            int64_t bi[3] = {-1, -1, -1};
            for (int s = 0; s < S; ++s) {
                const float *t = xyz2 + 3 * ((size_t)b * S + s);
                const float d = ((-2.0f * dot3_fma(q, t)) + sq) + sumsq3(t);
                if (d < bd[2]) {                      /* strict: an equal later index never displaces */
                    int k = 2;
                    while (k > 0 && d < bd[k - 1]) { bd[k] = bd[k - 1]; bi[k] = bi[k - 1]; --k; }
                    bd[k] = d; bi[k] = s;
                }
            }
            for (int k = 0; k < 3; ++k) {
                dist3[((size_t)b * N + n) * 3 + k] = bd[k];
                idx3[((size_t)b * N + n) * 3 + k] = bi[k];
            }
        }
}

void oracle_three_interpolate(const float *xyz1, const float *xyz2, const float *points2, int B, int N,
                              int S, int D, float *out /* [B,N,D] */) {
    float *d3 = (float *)malloc(sizeof(float) * (size_t)B * N * 3);
    int64_t *i3 = (int64_t *)malloc(sizeof(int64_t) * (size_t)B * N * 3);
    oracle_three_nn_expanded(xyz1, xyz2, B, N, S, d3, i3);
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int n = 0; n < N; ++n) {
            const size_t q = (size_t)b * N + n;
            float w[3];
            for (int k = 0; k < 3; ++k) w[k] = 1.0f / (d3[q * 3 + k] + 1e-8f);
            const float norm = (w[0] + w[1]) + w[2];
            for (int k = 0; k < 3; ++k) w[k] = w[k] / norm;
            const float *p0 = points2 + ((size_t)b * S + i3[q * 3 + 0]) * D;
            const float *p1 = points2 + ((size_t)b * S + i3[q * 3 + 1]) * D;
            const float *p2 = points2 + ((size_t)b * S + i3[q * 3 + 2]) * D;
            for (int c = 0; c < D; ++c) out[q * D + c] = (p0[c] * w[0] + p1[c] * w[1]) + p2[c] * w[2];
        }
    free(d3);
    free(i3);
}
