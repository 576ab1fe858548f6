// reart_amd/csrc/gemm.hip -- the dense part of the PointNet++ correspondence extractor on the
// gfx950 matrix cores: Y = relu(X W + b) [+ max-pool over K consecutive rows], exact fp32.
//
// Replaces the 1x1 Conv2d/Conv1d + BatchNorm(eval) + ReLU stacks and the max over nsample of
// the reference's PointNetSetAbstractionMsg / PointNetSetAbstraction / PointNetFeaturePropagation
// (networks/pointnet2_utils.py:209-235, 257-295, 309-348; model networks/feature_extractor.py:10-49).
// BatchNorm is folded into (W, b) on the host (eval mode: an affine map).
//
// MFMA: v_mfma_f32_32x32x2_f32 -- f32 in, f32 accumulate, bit-identical to a k-ordered fmaf
// chain (cdna guide section 3), so results stay within fp32 round-off of the reference's conv.
// Tile: 128 rows x 64 cols per workgroup (4 waves, each 32 x 64 = two accumulators sharing
// the A fragment), K step 16 through LDS, the global loads of step i+1 in flight (registers) while step i
// runs on the matrix cores; 16-byte loads for contiguous row segments.  The A tile can be GATHERED on the fly from the ball
// query indices (grouped features | relative xyz), so the grouped tensor [B,S,K,C] of the
// reference (up to 400 MB per scale at T=20) is never materialised.
#include "common.h"
#include "internal.h"
#include <math.h>

#define GM_BM 128
#define GM_BN 64
#define GM_BK 16   // measured: 32 is slower (8.3 vs 6.7 ms for the extractor: the Cin = 6 layers pad twice as far, fewer resident waves)
#define GM_LDA (GM_BK + 1)   // +1: column reads by 32 lanes hit 32 different banks
#define GM_LDB (GM_BN + 4)

typedef float f16v __attribute__((ext_vector_type(16)));

struct GemmArgs {
    const float *X; int ldx;          // plain input [rows, ldx] (used when idx == NULL)
    // gathered input: row r -> point gidx[r] of cloud b = r / (S*K), centre g = r / K
    const int64_t *idx; int K, S, Npts;
    const float *F; int D;            // features [B*Npts, D] (may be NULL, D = 0)
    const float *Q;                   // xyz [B*Npts, 3]
    const float *C;                   // centres [B*S, 3] (NULL: absolute xyz)
    int xyz_first;                    // 1: [xyz | F] (sample_and_group_all), 0: [F | xyz - centre] (MSG)
    const float *Wt;                  // [Cin, Cout]
    const float *bias;                // [Cout]
    int rows, Cin, Cout, relu, pool_k;
    float *Y; int ldy, ycol0;         // [rows, ldy] or [rows / pool_k, ldy], written at columns ycol0..
};

// Per-thread description of the A row this thread stages (fixed for the whole K loop): the
// gather index arithmetic (two integer divisions) is done once, not once per element.
struct ARow {
    const float *x;    // plain row, or NULL
    const float *f;    // gathered feature row (D floats), or NULL
    const float *q;    // gathered xyz row
    float c0, c1, c2;  // centre to subtract (0 when absolute)
    bool valid;
};

__device__ __forceinline__ ARow gemm_row(const GemmArgs &a, int r) {
    ARow w = {nullptr, nullptr, nullptr, 0.f, 0.f, 0.f, r < a.rows};
    if (!w.valid) return w;
    if (!a.idx) { w.x = a.X + (size_t)r * a.ldx; return w; }
    const int b = r / (a.S * a.K);
    const size_t prow = (size_t)b * a.Npts + (size_t)a.idx[r];
    w.q = a.Q + prow * 3;
    w.f = a.F ? a.F + prow * a.D : nullptr;
    if (a.C) {
        const float *c = a.C + (size_t)(r / a.K) * 3;
        w.c0 = c[0]; w.c1 = c[1]; w.c2 = c[2];
    }
    return w;
}

__device__ __forceinline__ float gemm_load_a(const GemmArgs &a, const ARow &w, int k) {
    if (!w.valid || k >= a.Cin) return 0.f;
    if (w.x) return w.x[k];
    const int kx = a.xyz_first ? k : k - a.D;      // index into the xyz part, valid when 0 <= kx < 3
    if (kx >= 0 && kx < 3) return w.q[kx] - (kx == 0 ? w.c0 : (kx == 1 ? w.c1 : w.c2));
    return w.f[a.xyz_first ? k - 3 : k];
}

__global__ __launch_bounds__(256) void mlp_gemm_kernel(GemmArgs a) {
    __shared__ float As[GM_BM * GM_LDA];
    __shared__ float Bs[GM_BK * GM_LDB];
    __shared__ float Pm[4][GM_BN];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int row0 = blockIdx.x * GM_BM, col0 = blockIdx.y * GM_BN;
    f16v c0 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f16v c1 = c0;
    constexpr int AK = GM_BK / 2;                     // k per thread of the A tile
    const int ar = tid >> 1, ak = (tid & 1) * AK;     // A tile: 128 rows x 2 half-rows of AK
    const int bk = tid >> 4, bc = (tid & 15) * 4;     // B tile: GM_BK k x 16 float4, rows bk, bk + 16, ...
    const ARow arow = gemm_row(a, row0 + ar);
    // 16-byte path for the A tile: the thread's 8 consecutive k lie inside one contiguous, aligned row
    // segment (plain rows, or the feature part of a gathered row); anything else goes element by element
    const float *aseg = arow.x ? arow.x : ((!a.xyz_first && arow.f) ? arow.f : nullptr);
    const int aseg_len = arow.x ? a.Cin : ((!a.xyz_first && arow.f) ? a.D : 0);
    const bool avec = arow.valid && aseg && ((((size_t)aseg) & 15) == 0);
    auto load_a = [&](int k0, float (&v)[AK]) {
        const int k = k0 + ak;
        if (avec && k + AK <= aseg_len) {
#pragma unroll
            for (int u = 0; u < AK; u += 4) {
                const float4 p = *(const float4 *)(aseg + k + u);
                v[u] = p.x; v[u + 1] = p.y; v[u + 2] = p.z; v[u + 3] = p.w;
            }
        } else {
#pragma unroll
            for (int u = 0; u < AK; ++u) v[u] = gemm_load_a(a, arow, k + u);
        }
    };
    auto load_b = [&](int kq) {
        const int k = kq + bk, c = col0 + bc;
        float4 w = {0.f, 0.f, 0.f, 0.f};
        if (k < a.Cin) {
            if (c + 3 < a.Cout && (a.Cout & 3) == 0) {
                w = *(const float4 *)(a.Wt + (size_t)k * a.Cout + c);
            } else {
                if (c < a.Cout) w.x = a.Wt[(size_t)k * a.Cout + c];
                if (c + 1 < a.Cout) w.y = a.Wt[(size_t)k * a.Cout + c + 1];
                if (c + 2 < a.Cout) w.z = a.Wt[(size_t)k * a.Cout + c + 2];
                if (c + 3 < a.Cout) w.w = a.Wt[(size_t)k * a.Cout + c + 3];
            }
        }
        return w;
    };
    // register double buffering: the loads of K-step i+1 are in flight while step i runs on the matrix cores
    float av8[AK];
    float4 bw[GM_BK / 16];
    load_a(0, av8);
#pragma unroll
    for (int h = 0; h < GM_BK / 16; ++h) bw[h] = load_b(16 * h);
    for (int k0 = 0; k0 < a.Cin; k0 += GM_BK) {
#pragma unroll
        for (int u = 0; u < AK; ++u) As[ar * GM_LDA + ak + u] = av8[u];
#pragma unroll
        for (int h = 0; h < GM_BK / 16; ++h) *(float4 *)(Bs + (bk + 16 * h) * GM_LDB + bc) = bw[h];
        __syncthreads();
        if (k0 + GM_BK < a.Cin) {
            load_a(k0 + GM_BK, av8);
#pragma unroll
            for (int h = 0; h < GM_BK / 16; ++h) bw[h] = load_b(k0 + GM_BK + 16 * h);
        }
#pragma unroll
        for (int kk = 0; kk < GM_BK; kk += 2) {
            const int kl = kk + (lane >> 5);
            const float av = As[(wv * 32 + (lane & 31)) * GM_LDA + kl];
            const float b0 = Bs[kl * GM_LDB + (lane & 31)];
            const float b1 = Bs[kl * GM_LDB + 32 + (lane & 31)];
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b0, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b1, c1, 0, 0, 0);
        }
        __syncthreads();
    }
    // epilogue: C/D layout row = (reg&3) + 8*(reg>>2) + 4*(lane>>5), col = lane & 31
    const int cA = col0 + (lane & 31), cB = cA + 32;
    const float biasA = (a.bias && cA < a.Cout) ? a.bias[cA] : 0.f;
    const float biasB = (a.bias && cB < a.Cout) ? a.bias[cB] : 0.f;
    float mA = -INFINITY, mB = -INFINITY;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int r = row0 + wv * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
        float vA = c0[reg] + biasA, vB = c1[reg] + biasB;
        if (a.relu) { vA = vA > 0.f ? vA : 0.f; vB = vB > 0.f ? vB : 0.f; }
        if (a.pool_k) {
            if (r < a.rows) { mA = fmaxf(mA, vA); mB = fmaxf(mB, vB); }
        } else if (r < a.rows) {
            if (cA < a.Cout) a.Y[(size_t)r * a.ldy + a.ycol0 + cA] = vA;
            if (cB < a.Cout) a.Y[(size_t)r * a.ldy + a.ycol0 + cB] = vB;
        }
    }
    if (!a.pool_k) return;
    // max over the wave's 32 rows, then over pool_k / 32 waves
    mA = fmaxf(mA, __shfl_xor(mA, 32, 64));
    mB = fmaxf(mB, __shfl_xor(mB, 32, 64));
    if (lane < 32) { Pm[wv][lane] = mA; Pm[wv][32 + lane] = mB; }
    __syncthreads();
    const int wpg = a.pool_k / 32;                 // waves per pooled group: 1, 2 or 4
    const int groups = 4 / wpg;
    for (int e = tid; e < groups * GM_BN; e += 256) {
        const int g = e / GM_BN, c = e % GM_BN;
        float m = -INFINITY;
        for (int w = 0; w < wpg; ++w) m = fmaxf(m, Pm[g * wpg + w][c]);
        const int prow = (row0 + g * a.pool_k) / a.pool_k;
        if (row0 + g * a.pool_k < a.rows && col0 + c < a.Cout) a.Y[(size_t)prow * a.ldy + a.ycol0 + col0 + c] = m;
    }
}

extern "C" int reart_mlp_layer(const float *X, int ldx, const int64_t *gather_idx, int K, int S, int Npts,
                               const float *F, int D, const float *Q, const float *C, int xyz_first,
                               const float *Wt, const float *bias, int rows, int Cin, int Cout, int relu,
                               int pool_k, float *Y, int ldy, int ycol0, void *stream) {
    if (rows < 0 || Cin < 1 || Cout < 1) return REART_ERR_INVALID_ARG;
    if (rows == 0) return REART_OK;
    if (!Wt || !Y || ycol0 < 0 || ldy < ycol0 + Cout) return REART_ERR_INVALID_ARG;
    if (gather_idx) {
        if (!Q || K < 1 || S < 1 || Npts < 1 || (D > 0 && !F) || Cin != D + 3) return REART_ERR_INVALID_ARG;
    } else if (!X || ldx < Cin) {
        return REART_ERR_INVALID_ARG;
    }
    if (pool_k && (pool_k != 32 && pool_k != 64 && pool_k != 128)) return REART_ERR_UNSUPPORTED;
    if (pool_k && rows % pool_k != 0) return REART_ERR_INVALID_ARG;
    GemmArgs a = {};
    a.X = X; a.ldx = ldx; a.idx = gather_idx; a.K = K; a.S = S; a.Npts = Npts; a.F = F; a.D = D; a.Q = Q; a.C = C;
    a.xyz_first = xyz_first; a.Wt = Wt; a.bias = bias; a.rows = rows; a.Cin = Cin; a.Cout = Cout; a.relu = relu;
    a.pool_k = pool_k; a.Y = Y; a.ldy = ldy; a.ycol0 = ycol0;
    const dim3 grid(reart_div_up(rows, GM_BM), reart_div_up(Cout, GM_BN));
    hipLaunchKernelGGL(mlp_gemm_kernel, grid, dim3(256), 0, (hipStream_t)stream, a);
    REART_CHECK_LAUNCH();
    return REART_OK;
}

// ---------------------------------------------------------------------------------------
// 3-NN inverse-distance feature interpolation of PointNetFeaturePropagation
// (networks/pointnet2_utils.py:326-336): idx/dist = 3 nearest of xyz2 for each xyz1 point
// (squared distance), w = 1/(d+1e-8) normalised, out = sum_k w_k points2[idx_k].
// out is written into columns [col0, col0+D) of a [B*N, ldo] matrix so that the concatenation
// with the skip features (:338-342) needs no extra pass.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void interp3_kernel(const float *__restrict__ dist,
                                                      const int64_t *__restrict__ idx,
                                                      const float *__restrict__ P2, int N, int S2, int D,
                                                      float *__restrict__ out, int ldo, int col0) {
    const int r = blockIdx.x, b = blockIdx.y;      // one workgroup per query point
    const size_t q = (size_t)b * N + r;
    float w[3];
    int id[3];
    float ws = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        w[k] = 1.0f / (dist[q * 3 + k] + 1e-8f);
        id[k] = (int)idx[q * 3 + k];
        ws += w[k];
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) w[k] = w[k] / ws;
    const float *p0 = P2 + ((size_t)b * S2 + id[0]) * D;
    const float *p1 = P2 + ((size_t)b * S2 + id[1]) * D;
    const float *p2 = P2 + ((size_t)b * S2 + id[2]) * D;
    for (int c = threadIdx.x; c < D; c += 256)
        out[q * ldo + col0 + c] = (p0[c] * w[0] + p1[c] * w[1]) + p2[c] * w[2];
}

extern "C" size_t reart_three_interpolate_workspace_bytes(int B, int N, int S2) {
    if (B <= 0 || N <= 0 || S2 <= 0) return 0;
    return reart_align_up(sizeof(float) * (size_t)B * N * 3, 256) + reart_align_up(sizeof(int64_t) * (size_t)B * N * 3, 256) +
           reart_knn_points_workspace_bytes(B, N, S2, 3);
}

extern "C" int reart_three_interpolate(const float *xyz1, const float *xyz2, const float *points2, int B,
                                       int N, int S2, int D, float *out, int ldo, int col0,
                                       void *workspace, size_t workspace_bytes, void *stream) {
    if (B < 0 || N < 0 || S2 < 3 || D < 1 || ldo < col0 + D) return REART_ERR_INVALID_ARG;
    if (B == 0 || N == 0) return REART_OK;
    if (!xyz1 || !xyz2 || !points2 || !out || !workspace) return REART_ERR_INVALID_ARG;
    if (workspace_bytes < reart_three_interpolate_workspace_bytes(B, N, S2)) return REART_ERR_INVALID_ARG;
    char *ws = (char *)workspace;
    float *dist = (float *)ws;
    size_t off = reart_align_up(sizeof(float) * (size_t)B * N * 3, 256);
    int64_t *idx = (int64_t *)(ws + off);
    off += reart_align_up(sizeof(int64_t) * (size_t)B * N * 3, 256);
    hipStream_t st = (hipStream_t)stream;
    int rc = reart_knn_run(1, &xyz1, &xyz2, nullptr, nullptr, B, &N, &S2, 3, 0, &dist, &idx, ws + off,
                           workspace_bytes - off, st);
    if (rc != REART_OK) return rc;
    hipLaunchKernelGGL(interp3_kernel, dim3(N, B), dim3(256), 0, st, dist, idx, points2, N, S2, D, out, ldo, col0);
    REART_CHECK_LAUNCH();
    return REART_OK;
}
