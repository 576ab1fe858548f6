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
 *                     NEAREST point) -- evaluated with the direct-difference distance instead of
 *                     the reference's matmul expansion (BLAS rounding is not
 *                     reproducible on a GPU; rows whose boundary margin is below 1e-5 r^2 are
 *                     reported through `margin` and excluded from bit-exact comparison)
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
                const float d = sqd3(q, xyz + 3 * ((size_t)b * N + k));
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
