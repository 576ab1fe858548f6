/*
 * reart_hip.h -- C ABI of libreart_hip.so: the MI355X (gfx950) implementation of
 * reart's per-iteration point-cloud hot path.
 *
 * Conventions:
 *   - plain pointers + sizes, no torch / pybind types; every pointer is a DEVICE
 *     pointer unless the parameter name starts with `h_`;
 *   - the library never allocates, frees or retains device memory: outputs and
 *     scratch are caller-owned (sizes from the *_workspace_bytes queries);
 *   - every entry point is asynchronous on the hipStream_t passed as `stream`
 *     (void* so that this header needs no HIP headers), re-entrant, does no host
 *     read of device data and is therefore hipGraph-capture safe;
 *   - return value: REART_OK (0) or a negative reart_status; never exit(), never
 *     throws.  (The reference's wrappers return 1 and exit(-1) on launch failure,
 *     networks/pointnet_lib/src/ball_query.cpp:25, ball_query_gpu.cu:62-66.)
 *   - fp32 data, int64 indices at the chamferdist / knn_cuda boundaries, int32 at
 *     the pointnet2_cuda boundary, exactly as the reference interfaces.
 *
 * Each entry point names the reference interface (file:line under the upstream
 * stevenlsw/reart tree) it replaces.
 */
#ifndef REART_HIP_H
#define REART_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    REART_OK = 0,
    REART_ERR_INVALID_ARG = -1,   /* null pointer, negative size, bad enum      */
    REART_ERR_UNSUPPORTED = -2,   /* e.g. D != 3, K > REART_MAX_K                */
    REART_ERR_LAUNCH = -3,        /* hipGetLastError() after a launch            */
    REART_ERR_NO_DEVICE = -4      /* no HIP device visible                       */
} reart_status;

#define REART_MAX_K 16            /* neighbours kept in registers per query      */

/* Library / device probes (host side, no device work). */
int reart_version(void);                       /* 100*major + minor              */
int reart_device_count(void);                  /* 0 when no GPU is visible       */
const char *reart_status_string(int status);

/* ------------------------------------------------------------------------ */
/* K-nearest neighbours / Chamfer                                            */
/* ------------------------------------------------------------------------ */

/* Replaces chamferdist._C.knn_points_idx(p1,p2,lengths1,lengths2,K,version)
 * (utils/chamfer.py:174; contract in the docstring :145-171).
 *   p1 [N,P1,3], p2 [N,P2,3] f32 contiguous; lengths1/2 [N] i64 or NULL (= full);
 *   dists [N,P1,K] f32 squared L2, idx [N,P1,K] i64 into p2, ascending by
 *   (distance, index); rows >= lengths1[n] and slots >= lengths2[n] are zero.
 * Distance contract: ((dx*dx)+(dy*dy))+(dz*dz) in fp32, no FMA, ties -> lowest j.
 * workspace (SoA target image + per-slice partial results):
 *   reart_knn_points_workspace_bytes(N,P1,P2,K) bytes. */
size_t reart_knn_points_workspace_bytes(int N, int P1, int P2, int K);
int reart_knn_points_idx(const float *p1, const float *p2,
                         const int64_t *lengths1, const int64_t *lengths2,
                         int N, int P1, int P2, int D, int K,
                         float *dists, int64_t *idx,
                         void *workspace, size_t workspace_bytes, void *stream);

/* Replaces chamferdist._C.knn_points_backward(p1,p2,lengths1,lengths2,idx,grad_dists)
 * (utils/chamfer.py:206-208).  grad_p1 [N,P1,3], grad_p2 [N,P2,3] are fully
 * written (zero where nothing contributes).  Deterministic: the scatter into
 * grad_p2 is a per-target gather over a counting sort of idx, summed in
 * ascending (i,k) order -- no float atomics.
 *   workspace: reart_knn_points_backward_workspace_bytes(N,P1,P2,K) bytes. */
size_t reart_knn_points_backward_workspace_bytes(int N, int P1, int P2, int K);
int reart_knn_points_backward(const float *p1, const float *p2,
                              const int64_t *lengths1, const int64_t *lengths2,
                              const int64_t *idx, const float *grad_dists,
                              int N, int P1, int P2, int D, int K,
                              float *grad_p1, float *grad_p2,
                              void *workspace, size_t workspace_bytes, void *stream);

/* Both directions of ChamferDistance.forward(bidirectional=True)
 * (utils/chamfer.py:78-123) in one launch: x,y [N,P,3];
 *   d_xy/i_xy: NN of each x point in y; d_yx/i_yx: NN of each y point in x. */
size_t reart_chamfer_bidir_workspace_bytes(int N, int P);
int reart_chamfer_bidir(const float *x, const float *y, int N, int P,
                        float *d_xy, int64_t *i_xy, float *d_yx, int64_t *i_yx,
                        void *workspace, size_t workspace_bytes, void *stream);

/* Replaces knn_cuda.KNN(k, transpose_mode=True).forward(ref, query)
 * (run_robot.py:65-66,122; shape contract utils/model_utils.py:42):
 *   ref [B,nr,3], query [B,nq,3] -> dist [B,nq,k] ascending, idx [B,nq,k] i64.
 * euclidean != 0: dist = sqrt(squared distance) (upstream KNN_CUDA 0.2).
 * workspace: reart_knn_points_workspace_bytes(B, nq, nr, k). */
int reart_knn_cuda(const float *ref, const float *query, int B, int nr, int nq,
                   int D, int k, int euclidean, float *dist, int64_t *idx,
                   void *workspace, size_t workspace_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* REART_HIP_H */
