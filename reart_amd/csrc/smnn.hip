// reart_amd/csrc/smnn.hip -- second-nearest-neighbour ratio test + mutual filter on 64-d descriptors.
//
// Replaces match_smnn / match_snn of the reference (utils/flow_utils.py:7-100, called from
// compute_corr_list_filter :116-143 once before the loop): the reference materialises the full
// [N,N] descriptor distance matrix with torch.cdist, takes topk(2) per row and per column,
// applies the ratio test d1/d2 <= th and keeps mutual pairs.  Here each query row keeps its two
// smallest distances while streaming over the other cloud's descriptors (wave-uniform -> scalar
// loads); nothing of size N^2 is stored.  Distances by direct differences in fp32.
#include "common.h"
#include "internal.h"
#include <math.h>

#define SM_D 64   // descriptor width of PointNet2Msg2 (feature_extractor.py:63)

// desc [E][N][64] row-major (point-major).  For query set A against target set B:
// best[0..1] squared distances ascending (ties -> lowest index), idx of the best.
__global__ __launch_bounds__(64) void top2_kernel(const float *__restrict__ A, const float *__restrict__ Bm,
                                                  int NA, int NB, float *__restrict__ d01,
                                                  int *__restrict__ i0) {
    const int e = blockIdx.y;
    const int i = blockIdx.x * 64 + threadIdx.x;
    const int ic = i < NA ? i : NA - 1;
    float q[SM_D];
    const float *qa = A + ((size_t)e * NA + ic) * SM_D;
#pragma unroll
    for (int k = 0; k < SM_D; ++k) q[k] = qa[k];
    const float *tb = Bm + (size_t)e * NB * SM_D;
    float b0 = INFINITY, b1 = INFINITY;
    int j0 = 0;
    for (int j = 0; j < NB; ++j) {
        const float *t = tb + (size_t)j * SM_D;   // wave-uniform address: scalar loads
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < SM_D; ++k) {
            const float df = q[k] - t[k];
            s = fmaf(df, df, s);
        }
        if (s < b0) { b1 = b0; b0 = s; j0 = j; }
        else if (s < b1) b1 = s;
    }
    if (i < NA) {
        d01[2 * ((size_t)e * NA + i)] = b0;
        d01[2 * ((size_t)e * NA + i) + 1] = b1;
        i0[(size_t)e * NA + i] = j0;
    }
}

// mutual second-nearest-neighbour filter: keep[e][i] = 1 and tgt[e][i] = j iff
//   j = nn_B(a_i), ratio_A(i) <= th, nn_A(b_j) == i, ratio_B(j) <= th      (flow_utils.py:76-96)
__global__ __launch_bounds__(256) void smnn_kernel(const float *__restrict__ dA, const int *__restrict__ iA,
                                                   const float *__restrict__ dB, const int *__restrict__ iB,
                                                   int NA, int NB, float th, uint8_t *__restrict__ keep,
                                                   int64_t *__restrict__ tgt) {
    const int e = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= NA) return;
    const size_t a = (size_t)e * NA + i;
    const int j = iA[a];
    const size_t b = (size_t)e * NB + j;
    const float ra = sqrtf(dA[2 * a]) / sqrtf(dA[2 * a + 1]);   // vals[:,0] / vals[:,1] of Euclidean distances
    const float rb = sqrtf(dB[2 * b]) / sqrtf(dB[2 * b + 1]);
    // th < 0: no ratio test -- plain mutual nearest neighbours (matching="mnn", flow_utils.py:102-113, 126-137)
    const bool ok = (th < 0.f || ((ra <= th) && (rb <= th))) && (iB[b] == i);
    keep[a] = ok ? 1 : 0;
    tgt[a] = j;
}

extern "C" size_t reart_match_smnn_workspace_bytes(int E, int NA, int NB) {
    if (E <= 0 || NA <= 0 || NB <= 0) return 0;
    return reart_align_up(sizeof(float) * 2 * (size_t)E * NA, 256) + reart_align_up(sizeof(int) * (size_t)E * NA, 256) +
           reart_align_up(sizeof(float) * 2 * (size_t)E * NB, 256) + reart_align_up(sizeof(int) * (size_t)E * NB, 256);
}

extern "C" int reart_match_smnn(const float *desc1, const float *desc2, int E, int N1, int N2, int D, float th,
                                uint8_t *keep, int64_t *tgt, void *workspace, size_t workspace_bytes,
                                void *stream) {
    if (E < 0 || N1 < 1 || N2 < 1) return REART_ERR_INVALID_ARG;
    if (th >= 0.f && (N1 < 2 || N2 < 2)) return REART_ERR_INVALID_ARG;   // the ratio test needs a second neighbour (the reference raises)
    if (D != SM_D) return REART_ERR_UNSUPPORTED;
    if (E == 0) return REART_OK;
    if (!desc1 || !desc2 || !keep || !tgt || !workspace) return REART_ERR_INVALID_ARG;
    if (workspace_bytes < reart_match_smnn_workspace_bytes(E, N1, N2)) return REART_ERR_INVALID_ARG;
    char *ws = (char *)workspace;
    float *dA = (float *)ws; ws += reart_align_up(sizeof(float) * 2 * (size_t)E * N1, 256);
    int *iA = (int *)ws; ws += reart_align_up(sizeof(int) * (size_t)E * N1, 256);
    float *dB = (float *)ws; ws += reart_align_up(sizeof(float) * 2 * (size_t)E * N2, 256);
    int *iB = (int *)ws;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(top2_kernel, dim3(reart_div_up(N1, 64), E), dim3(64), 0, st, desc1, desc2, N1, N2, dA, iA);
    hipLaunchKernelGGL(top2_kernel, dim3(reart_div_up(N2, 64), E), dim3(64), 0, st, desc2, desc1, N2, N1, dB, iB);
    hipLaunchKernelGGL(smnn_kernel, dim3(reart_div_up(N1, 256), E), dim3(256), 0, st, dA, iA, dB, iB, N1, N2, th, keep, tgt);
    REART_CHECK_LAUNCH();
    return REART_OK;
}
