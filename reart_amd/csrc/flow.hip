// reart_amd/csrc/flow.hip -- flow-loss kernels for gfx950.
//
// Replaces blend_anchor_motion (reference utils/flow_utils.py:147-170, including its
// knn_cuda.KNN(k=3) call at :158) and flow_loss (networks/loss.py:10-21) with its gradient.
#include "common.h"
#include "internal.h"
#include <math.h>

// ---------------------------------------------------------------------------------------
// blend: inverse-distance blending of the k nearest anchors' flow + validity mask
//   d[d < 1e-10] = 1e-10; w = 1/d; w /= sum w; flow = sum_k w_k f[idx_k]
//   mask = min d <= max_k |f[idx_k]|^2  or  min d <= 0.05            (flow_utils.py:160-167)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void blend_kernel(const float *__restrict__ dist,
                                                    const int64_t *__restrict__ idx,
                                                    const float *__restrict__ ref_flow, int nq,
                                                    int k, float *__restrict__ flow,
                                                    uint8_t *__restrict__ mask) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= nq) return;
    float wsum = 0.f, dmin = INFINITY, fmx = -INFINITY;
    for (int j = 0; j < k; ++j) {
        float d = dist[(size_t)n * k + j];
        if (d < 1e-10f) d = 1e-10f;
        wsum += 1.0f / d;
        dmin = fminf(dmin, d);
        const float *f = ref_flow + 3 * idx[(size_t)n * k + j];
        fmx = fmaxf(fmx, (f[0] * f[0] + f[1] * f[1]) + f[2] * f[2]);
    }
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int j = 0; j < k; ++j) {
        float d = dist[(size_t)n * k + j];
        if (d < 1e-10f) d = 1e-10f;
        const float wn = (1.0f / d) / wsum;
        const float *f = ref_flow + 3 * idx[(size_t)n * k + j];
        a0 += f[0] * wn; a1 += f[1] * wn; a2 += f[2] * wn;
    }
    flow[3 * (size_t)n] = a0; flow[3 * (size_t)n + 1] = a1; flow[3 * (size_t)n + 2] = a2;
    mask[n] = (dmin <= fmx) || (dmin <= 0.05f);
}

static size_t blend_ws_layout(int nq, int nr, int k, size_t *o_d, size_t *o_i, size_t *o_knn) {
    size_t off = 0;
    *o_d = off; off += reart_align_up(sizeof(float) * (size_t)nq * k, 256);
    *o_i = off; off += reart_align_up(sizeof(int64_t) * (size_t)nq * k, 256);
    *o_knn = off; off += reart_knn_points_workspace_bytes(1, nq, nr, k);
    return off;
}

extern "C" size_t reart_blend_anchor_motion_workspace_bytes(int nq, int nr, int k) {
    if (nq <= 0 || nr <= 0 || k <= 0) return 0;
    size_t a, b, c;
    return blend_ws_layout(nq, nr, k, &a, &b, &c);
}

extern "C" int reart_blend_anchor_motion(const float *query, const float *ref, const float *ref_flow,
                                         int nq, int nr, int k, int euclidean, float *flow,
                                         uint8_t *mask, void *workspace, size_t workspace_bytes,
                                         void *stream) {
    if (nq < 0 || nr < 0 || k < 1) return REART_ERR_INVALID_ARG;
    if (k > REART_MAX_K) return REART_ERR_UNSUPPORTED;
    if (k > nr) return REART_ERR_INVALID_ARG;
    if (nq == 0) return REART_OK;
    if (!query || !ref || !ref_flow || !flow || !mask || !workspace) return REART_ERR_INVALID_ARG;
    size_t o_d, o_i, o_knn;
    if (workspace_bytes < blend_ws_layout(nq, nr, k, &o_d, &o_i, &o_knn)) return REART_ERR_INVALID_ARG;
    char *ws = (char *)workspace;
    float *dist = (float *)(ws + o_d);
    int64_t *idx = (int64_t *)(ws + o_i);
    hipStream_t st = (hipStream_t)stream;
    int rc = reart_knn_run(1, &query, &ref, nullptr, nullptr, 1, &nq, &nr, k, euclidean ? 1 : 0, &dist,
                           &idx, ws + o_knn, workspace_bytes - o_knn, st);
    if (rc != REART_OK) return rc;
    hipLaunchKernelGGL(blend_kernel, dim3(reart_div_up(nq, 256)), dim3(256), 0, st, dist, idx, ref_flow,
                       nq, k, flow, mask);
    REART_CHECK_LAUNCH();
    return REART_OK;
}

// The T-1 blends of one iteration (run_robot.py:194-201 calls blend_anchor_motion once per frame pair) as ONE call: the queries
// of all frames [B,nq,3], the frames' reference sets padded to a common length [B,nr_max,3] with their true lengths ref_len [B]
// (NULL: all nr_max) -- one search over the batch (the same exact brute-force kernels with per-batch target lengths, so the same
// neighbours, distances and tie rule as B separate calls) and one blend launch.  In the kinematic projection loop the nine
// per-frame calls were 36 launches of 20 us each on side streams: 0.33 ms of an iteration between two re-solves.
__global__ __launch_bounds__(256) void blend_batch_kernel(const float *__restrict__ dist, const int64_t *__restrict__ idx,
                                                          const float *__restrict__ ref_flow, int nq, int nr_max, int k,
                                                          float *__restrict__ flow, uint8_t *__restrict__ mask) {
    const int n = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (n >= nq) return;
    const size_t q = (size_t)b * nq + n;
    const float *rf = ref_flow + (size_t)b * nr_max * 3;
    float wsum = 0.f, dmin = INFINITY, fmx = -INFINITY;
    for (int j = 0; j < k; ++j) {
        float d = dist[q * k + j];
        if (d < 1e-10f) d = 1e-10f;
        wsum += 1.0f / d;
        dmin = fminf(dmin, d);
        const float *f = rf + 3 * idx[q * k + j];
        fmx = fmaxf(fmx, (f[0] * f[0] + f[1] * f[1]) + f[2] * f[2]);
    }
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int j = 0; j < k; ++j) {
        float d = dist[q * k + j];
        if (d < 1e-10f) d = 1e-10f;
        const float wn = (1.0f / d) / wsum;
        const float *f = rf + 3 * idx[q * k + j];
        a0 += f[0] * wn; a1 += f[1] * wn; a2 += f[2] * wn;
    }
    flow[3 * q] = a0; flow[3 * q + 1] = a1; flow[3 * q + 2] = a2;
    mask[q] = (dmin <= fmx) || (dmin <= 0.05f);
}

static size_t blend_batch_ws_layout(int B, int nq, int nr, int k, size_t *o_d, size_t *o_i, size_t *o_knn) {
    size_t off = 0;
    *o_d = off; off += reart_align_up(sizeof(float) * (size_t)B * nq * k, 256);
    *o_i = off; off += reart_align_up(sizeof(int64_t) * (size_t)B * nq * k, 256);
    *o_knn = off; off += reart_knn_points_workspace_bytes(B, nq, nr, k);
    return off;
}

extern "C" size_t reart_blend_anchor_motion_batch_workspace_bytes(int B, int nq, int nr_max, int k) {
    if (B <= 0 || nq <= 0 || nr_max <= 0 || k <= 0) return 0;
    size_t a, b, c;
    return blend_batch_ws_layout(B, nq, nr_max, k, &a, &b, &c);
}

extern "C" int reart_blend_anchor_motion_batch(const float *query, const float *ref, const float *ref_flow, const int64_t *ref_len,
                                               int B, int nq, int nr_max, int k, int euclidean, float *flow, uint8_t *mask,
                                               void *workspace, size_t workspace_bytes, void *stream) {
    if (B < 0 || nq < 0 || nr_max < 0 || k < 1) return REART_ERR_INVALID_ARG;
    if (k > REART_MAX_K) return REART_ERR_UNSUPPORTED;
    if (k > nr_max) return REART_ERR_INVALID_ARG;             // (a frame whose ref_len is below k: the caller's error, like nr < k above)
    if (B == 0 || nq == 0) return REART_OK;
    if (!query || !ref || !ref_flow || !flow || !mask || !workspace) return REART_ERR_INVALID_ARG;
    size_t o_d, o_i, o_knn;
    if (workspace_bytes < blend_batch_ws_layout(B, nq, nr_max, k, &o_d, &o_i, &o_knn)) return REART_ERR_INVALID_ARG;
    char *ws = (char *)workspace;
    float *dist = (float *)(ws + o_d);
    int64_t *idx = (int64_t *)(ws + o_i);
    hipStream_t st = (hipStream_t)stream;
    const int64_t *lent = ref_len;
    int rc = reart_knn_run(1, &query, &ref, nullptr, ref_len ? &lent : nullptr, B, &nq, &nr_max, k, euclidean ? 1 : 0, &dist, &idx,
                           ws + o_knn, workspace_bytes - o_knn, st);
    if (rc != REART_OK) return rc;
    hipLaunchKernelGGL(blend_batch_kernel, dim3(reart_div_up(nq, 256), B), dim3(256), 0, st, dist, idx, ref_flow, nq, nr_max, k, flow, mask);
    REART_CHECK_LAUNCH();
    return REART_OK;
}

// ---------------------------------------------------------------------------------------
// flow_loss (networks/loss.py:10-21) and d loss / d pred
//   L = sum_{b,n} [ m f + smooth (not m) |pred|^2 ],  f = |pred-gt|^2 or Huber(delta=1)
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float huber1(float x) {
    const float a = fabsf(x);
    return a <= 1.0f ? 0.5f * x * x : (a - 0.5f);
}
__device__ __forceinline__ float huber1_grad(float x) {
    return fabsf(x) <= 1.0f ? x : (x > 0.f ? 1.0f : -1.0f);
}

#define FL_BS 256
__global__ __launch_bounds__(FL_BS) void flow_loss_kernel(const float *__restrict__ gt,
                                                          const float *__restrict__ pred,
                                                          const uint8_t *__restrict__ mask, size_t total,
                                                          int robust, float smooth,
                                                          float *__restrict__ grad,
                                                          double *__restrict__ partial) {
    __shared__ double s_red[FL_BS / REART_WAVE];
    double acc = 0.0;
    for (size_t e = (size_t)blockIdx.x * FL_BS + threadIdx.x; e < total; e += (size_t)gridDim.x * FL_BS) {
        const bool m = mask ? (mask[e] != 0) : true;
        float f = 0.f, sm = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float p = pred[3 * e + c], d = p - gt[3 * e + c];
            f += robust ? huber1(d) : d * d;
            sm += p * p;
            if (grad) {
                const float gf = robust ? huber1_grad(d) : 2.0f * d;
                grad[3 * e + c] = m ? gf : smooth * (2.0f * p);
            }
        }
        acc += m ? (double)f : (double)(smooth * sm);
    }
    acc = reart_wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < FL_BS / REART_WAVE; ++w) t += s_red[w];
        partial[blockIdx.x] = t;
    }
}
__global__ void flow_loss_final_kernel(const double *__restrict__ partial, int n, float *__restrict__ loss) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < n; ++i) t += partial[i];
        loss[0] = (float)t;
    }
}

#define FL_MAXBLOCKS 256
extern "C" size_t reart_flow_loss_workspace_bytes(void) { return sizeof(double) * FL_MAXBLOCKS; }

extern "C" int reart_flow_loss(const float *gt, const float *pred, const uint8_t *mask, int B, int N,
                               int robust, float smooth_weight, float *loss, float *grad_pred,
                               void *workspace, size_t workspace_bytes, void *stream) {
    if (B < 0 || N < 0) return REART_ERR_INVALID_ARG;
    if (!loss || !workspace || workspace_bytes < reart_flow_loss_workspace_bytes()) return REART_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    const size_t total = (size_t)B * N;
    if (total && (!gt || !pred)) return REART_ERR_INVALID_ARG;
    int blocks = (int)((total + FL_BS - 1) / FL_BS);
    blocks = blocks > FL_MAXBLOCKS ? FL_MAXBLOCKS : (blocks < 1 ? 1 : blocks);
    hipLaunchKernelGGL(flow_loss_kernel, dim3(blocks), dim3(FL_BS), 0, st, gt, pred, mask, total, robust,
                       smooth_weight, grad_pred, (double *)workspace);
    hipLaunchKernelGGL(flow_loss_final_kernel, dim3(1), dim3(64), 0, st, (const double *)workspace, blocks, loss);
    REART_CHECK_LAUNCH();
    return REART_OK;
}
