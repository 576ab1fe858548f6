// reart_amd/csrc/structure.hip -- end-of-run structure extraction for gfx950.
//
// The reference finishes every optimisation instance with ~25 Python functions that issue hundreds of tiny
// tensor ops and one Python loop per part / per part pair (run_robot.py:224-330).  Their numerical content is
// four small computations; each is one launch here:
//   screw_fit_kernel       utils/graph_utils.py  compute_relative_trans :170-186, compute_geo_cost :131-167,
//                          compute_mean_screw_param :207-232, compute_screw_trans :235-283, frobenius_cost :189-196,
//                          merge_graph's identity cost :338-342; utils/kinematic_utils.py build_graph :84-99
//                          (with screw_se3/dq_utils.py transform_to_dq :129-134, dq_to_screw :137-182,
//                           screw_se3/geo_utils.py matrix_to_quaternion :536-587, inverse_transformation :9-53)
//   part_fps_kernel        utils/graph_utils.py  fps_sample_cano :37-52 (per-part farthest point sampling)
//   part_pair_cost_kernel  utils/graph_utils.py  compute_spatial_cost :70-84 + compute_joint_cost :87-100
//   group_err_kernel       utils/model_utils.py  compute_group_temporal_err :107-118
// All of it is a few thousand rigid transforms / a few hundred 20-point sets: latency-bound, so the design goal is
// "one launch, no host round trip per part", not bandwidth.
#include "common.h"
#include "internal.h"
#include "screw_dev.h"
#include <math.h>

// ------------------------------------------------------------------------------------------------------------
// relative transform inv(A) * B of two [R|t] (4x4 row-major in memory) -> 3x4
__device__ __forceinline__ void rel_transform(const float *A, const float *B, float *M) {
    float tinv[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)  // -R_a^T t_a
        tinv[i] = (-A[i] * A[3] + -A[4 + i] * A[7]) + -A[8 + i] * A[11];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) M[4 * i + j] = (A[i] * B[j] + A[4 + i] * B[4 + j]) + A[8 + i] * B[8 + j];
        M[4 * i + 3] = ((A[i] * B[3] + A[4 + i] * B[7]) + A[8 + i] * B[11]) + tinv[i];
    }
}

__device__ __forceinline__ void q_mul(const float *q1, const float *q2, float *o) {  // dq_utils.py:63-83
    o[0] = ((q2[0] * q1[0] - q2[1] * q1[1]) - q2[2] * q1[2]) - q2[3] * q1[3];
    o[1] = ((q2[0] * q1[1] + q2[1] * q1[0]) - q2[2] * q1[3]) + q2[3] * q1[2];
    o[2] = ((q2[0] * q1[2] + q2[1] * q1[3]) + q2[2] * q1[0]) - q2[3] * q1[1];
    o[3] = ((q2[0] * q1[3] - q2[1] * q1[2]) + q2[2] * q1[1]) + q2[3] * q1[0];
}

// rigid transform M (3x4) -> screw parameters: axis l, moment m, angle theta, displacement d
__device__ __forceinline__ void transform_to_screw(const float *M, float *l, float *m, float &theta, float &d) {
    const float eps = 1e-6f;
    const float m00 = M[0], m01 = M[1], m02 = M[2], m10 = M[4], m11 = M[5], m12 = M[6], m20 = M[8], m21 = M[9], m22 = M[10];
    const float a[4] = {((1.0f + m00) + m11) + m22, ((1.0f + m00) - m11) - m22, ((1.0f - m00) + m11) - m22,
                        ((1.0f - m00) - m11) + m22};
    float qa[4];
    int best = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        qa[i] = a[i] > 0.f ? sqrtf(a[i]) : 0.f;
        if (qa[i] > qa[best]) best = i;  // first maximum
    }
    float c[4];
    if (best == 0) { c[0] = qa[0] * qa[0]; c[1] = m21 - m12; c[2] = m02 - m20; c[3] = m10 - m01; }
    else if (best == 1) { c[0] = m21 - m12; c[1] = qa[1] * qa[1]; c[2] = m10 + m01; c[3] = m02 + m20; }
    else if (best == 2) { c[0] = m02 - m20; c[1] = m10 + m01; c[2] = qa[2] * qa[2]; c[3] = m12 + m21; }
    else { c[0] = m10 - m01; c[1] = m20 + m02; c[2] = m21 + m12; c[3] = qa[3] * qa[3]; }
    const float den = 2.0f * fmaxf(qa[best], 0.1f);
    float qr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) qr[i] = c[i] / den;
    const float tq[4] = {0.f, M[3], M[7], M[11]};
    float qd[4];
    q_mul(tq, qr, qd);
#pragma unroll
    for (int i = 0; i < 4; ++i) qd[i] *= 0.5f;
    const float nq = sqrtf(((qr[0] * qr[0] + qr[1] * qr[1]) + qr[2] * qr[2]) + qr[3] * qr[3]);
    const float qn[4] = {qr[0] / nq, qr[1] / nq, qr[2] / nq, qr[3] / nq};
    const float nim = sqrtf((qn[1] * qn[1] + qn[2] * qn[2]) + qn[3] * qn[3]);
    float th = 2.0f * atan2f(nim, qn[0]);
    const bool no_rot = (fabsf(th) < eps) || (fabsf(th - PI_F) < eps);
    const float conj[4] = {qr[0], -qr[1], -qr[2], -qr[3]};
    const float qd2[4] = {2.0f * qd[0], 2.0f * qd[1], 2.0f * qd[2], 2.0f * qd[3]};
    float tt[4];
    q_mul(qd2, conj, tt);
    const float t[3] = {tt[1], tt[2], tt[3]};
    float dd;
    if (!no_rot) {
        const float s = sinf(th / 2.0f);
        l[0] = qr[1] / s; l[1] = qr[2] / s; l[2] = qr[3] / s;
        dd = 0.f;
    } else {
        dd = sqrtf((t[0] * t[0] + t[1] * t[1]) + t[2] * t[2]);
        const float dn = dd + 1e-10f;
        l[0] = t[0] / dn; l[1] = t[1] / dn; l[2] = t[2] / dn;
    }
    const float cs = (l[0] + l[1]) + l[2];
    if (!(cs >= 0.f)) { th = -th; l[0] = -l[0]; l[1] = -l[1]; l[2] = -l[2]; if (no_rot) dd = -dd; }
    if (!no_rot) dd = (t[0] * l[0] + t[1] * l[1]) + t[2] * l[2];
    // torch.isclose(d, 0): |d| <= 1e-8
    if (no_rot && fabsf(dd) <= 1e-8f) l[0] = 1.0f;
    if (no_rot) th = eps;
    const float tl[3] = {t[1] * l[2] - t[2] * l[1], t[2] * l[0] - t[0] * l[2], t[0] * l[1] - t[1] * l[0]};
    const float tn = tanf(th / 2.0f);
    const float u[3] = {tl[0] / tn, tl[1] / tn, tl[2] / tn};
    m[0] = 0.5f * (tl[0] + (l[1] * u[2] - l[2] * u[1]));
    m[1] = 0.5f * (tl[1] + (l[2] * u[0] - l[0] * u[2]));
    m[2] = 0.5f * (tl[2] + (l[0] * u[1] - l[1] * u[0]));
    theta = th;
    d = dd;
}

// sum of squares of (P * inv(G) - I) for two 3x4 rigid transforms (frobenius_cost; the 4th row contributes 0)
__device__ __forceinline__ float frob_cost(const float *Pm, const float *Gm) {
    // inv(G) = [G_R^T | -G_R^T g_t]
    float gi[12];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) gi[4 * i + j] = Gm[4 * j + i];
        gi[4 * i + 3] = (-Gm[i] * Gm[3] + -Gm[4 + i] * Gm[7]) + -Gm[8 + i] * Gm[11];
    }
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float e = ((Pm[4 * i] * gi[j] + Pm[4 * i + 1] * gi[4 + j]) + Pm[4 * i + 2] * gi[8 + j]) - (i == j ? 1.f : 0.f);
            acc += e * e;
        }
        const float e = ((Pm[4 * i] * gi[3] + Pm[4 * i + 1] * gi[7]) + Pm[4 * i + 2] * gi[11]) + Pm[4 * i + 3];
        acc += e * e;
    }
    return acc;
}

#define SF_BS 256

struct ScrewFitArgs {
    const float *trans;    // [T,P,4,4]; with pairs == nullptr: the relative transforms themselves [T,E,4,4]
    const int *pairs;      // [E,2] (src, tgt): rel = inv(trans[:,src]) * trans[:,tgt]
    int T, P, E, plain_mean;
    float *screw;          // [T,E,8]  axis, moment, theta, distance       (nullable)
    float *rel;            // [T,E,4,4]                                     (nullable)
    float *mean;           // [E,6]    mean axis, mean moment               (workspace when the caller passes none)
    float *recon;          // [T,E,4,4] reconstruction by the cheaper joint type (nullable)
    float *cost;           // [E,4]    revolute, prismatic, min, identity (mean over frames)
    float *mean_cost;      // scalar   mean_e(min) / T                      (nullable)
};

__device__ __forceinline__ void load_rel(const ScrewFitArgs &a, int t, int e, float *M) {
    if (a.pairs) {
        const float *A = a.trans + ((size_t)t * a.P + a.pairs[2 * e]) * 16;
        const float *B = a.trans + ((size_t)t * a.P + a.pairs[2 * e + 1]) * 16;
        float Al[12], Bl[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) { Al[i] = A[i]; Bl[i] = B[i]; }
        rel_transform(Al, Bl, M);
    } else {
        const float *R = a.trans + ((size_t)t * a.E + e) * 16;
#pragma unroll
        for (int i = 0; i < 12; ++i) M[i] = R[i];
    }
}

// One workgroup; thread <- edge (strided), sequential over the T frames of that edge.
__global__ __launch_bounds__(SF_BS) void screw_fit_kernel(ScrewFitArgs a) {
    __shared__ double s_red[SF_BS / 64];
    __shared__ double s_tot[2];
    const int tid = threadIdx.x;
    // ---- pass 1: screw parameters per frame, masked means (compute_mean_screw_param)
    for (int e = tid; e < a.E; e += SF_BS) {
        float sa[6] = {0, 0, 0, 0, 0, 0}, su[6] = {0, 0, 0, 0, 0, 0};
        int nall = 0, nkeep = 0;
        for (int t = 0; t < a.T; ++t) {
            float M[12], l[3], m[3], th, d;
            load_rel(a, t, e, M);
            transform_to_screw(M, l, m, th, d);
            if (a.screw) {
                float *o = a.screw + ((size_t)t * a.E + e) * 8;
                o[0] = l[0]; o[1] = l[1]; o[2] = l[2]; o[3] = m[0]; o[4] = m[1]; o[5] = m[2]; o[6] = th; o[7] = d;
            }
            if (a.rel) {
                float *o = a.rel + ((size_t)t * a.E + e) * 16;
#pragma unroll
                for (int i = 0; i < 12; ++i) o[i] = M[i];
                o[12] = 0.f; o[13] = 0.f; o[14] = 0.f; o[15] = 1.f;
            }
            const bool unit = ((fabsf(th) <= 1e-5f) || (fabsf(th - PI_F) <= 1e-5f)) && (d <= 1e-5f);
#pragma unroll
            for (int c = 0; c < 3; ++c) { sa[c] += l[c]; sa[3 + c] += m[c]; }
            ++nall;
            if (!unit) {
#pragma unroll
                for (int c = 0; c < 3; ++c) { su[c] += l[c]; su[3 + c] += m[c]; }
                ++nkeep;
            }
        }
        const bool plain = a.plain_mean || a.E <= 1 || nkeep == 0;
#pragma unroll
        for (int c = 0; c < 6; ++c) a.mean[(size_t)e * 6 + c] = plain ? sa[c] / (float)nall : su[c] / (float)nkeep;
    }
    // ---- pass 2: revolute / prismatic reconstructions and their costs
    double sum2 = 0.0;
    for (int e = tid; e < a.E; e += SF_BS) {
        float ml[3], mm[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) { ml[c] = a.mean[(size_t)e * 6 + c]; mm[c] = a.mean[(size_t)e * 6 + 3 + c]; }
        float cr = 0.f, c1 = 0.f, ci = 0.f;
        for (int t = 0; t < a.T; ++t) {
            float M[12], l[3], m[3], th, d, Rr[12], Rp[12];
            load_rel(a, t, e, M);
            transform_to_screw(M, l, m, th, d);
            screw_fwd(ml, mm, th, 1e-6f, Rr);
            screw_fwd(ml, mm, 1e-6f, d, Rp);
            cr += frob_cost(Rr, M);
            const float Mp[12] = {1.f, 0.f, 0.f, M[3], 0.f, 1.f, 0.f, M[7], 0.f, 0.f, 1.f, M[11]};
            c1 += frob_cost(Rp, Mp);
            float idc = 0.f;
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float ei = M[4 * i + j] - (i == j ? 1.f : 0.f);
                    idc += ei * ei;
                    if (j < 3) { const float e2 = Rp[4 * i + j] - M[4 * i + j]; sum2 += (double)e2 * (double)e2; }
                }
            ci += idc;
        }
        a.cost[(size_t)e * 4 + 0] = cr;
        a.cost[(size_t)e * 4 + 1] = c1;
        a.cost[(size_t)e * 4 + 3] = ci / (float)a.T;
    }
    // geo_cost_2: F.mse_loss over every rotation element of every frame and edge (one scalar)
    for (int o = 32; o >= 1; o >>= 1) sum2 += __shfl_xor(sum2, o, 64);
    if ((tid & 63) == 0) s_red[tid >> 6] = sum2;
    __syncthreads();
    if (tid == 0) {
        double tot = 0.0;
        for (int w = 0; w < SF_BS / 64; ++w) tot += s_red[w];
        s_tot[0] = tot / ((double)a.T * (double)a.E * 9.0);
    }
    __syncthreads();
    const float cost2 = (float)s_tot[0];
    double gsum = 0.0;
    for (int e = tid; e < a.E; e += SF_BS) {
        const float cr = a.cost[(size_t)e * 4 + 0];
        const float cp = a.cost[(size_t)e * 4 + 1] + cost2;
        a.cost[(size_t)e * 4 + 1] = cp;
        const float mn = cp < cr ? cp : cr;   // torch.min
        a.cost[(size_t)e * 4 + 2] = mn;
        gsum += (double)mn;
        if (a.recon) {
            const bool pris = cp <= cr;
            float ml[3], mm[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) { ml[c] = a.mean[(size_t)e * 6 + c]; mm[c] = a.mean[(size_t)e * 6 + 3 + c]; }
            for (int t = 0; t < a.T; ++t) {
                float M[12], l[3], m[3], th, d, R[12];
                load_rel(a, t, e, M);
                transform_to_screw(M, l, m, th, d);
                screw_fwd(ml, mm, pris ? 1e-6f : th, pris ? d : 1e-6f, R);
                float *o = a.recon + ((size_t)t * a.E + e) * 16;
#pragma unroll
                for (int i = 0; i < 12; ++i) o[i] = R[i];
                o[12] = 0.f; o[13] = 0.f; o[14] = 0.f; o[15] = 1.f;
            }
        }
    }
    if (a.mean_cost) {
        __syncthreads();
        for (int o = 32; o >= 1; o >>= 1) gsum += __shfl_xor(gsum, o, 64);
        if ((tid & 63) == 0) s_red[tid >> 6] = gsum;
        __syncthreads();
        if (tid == 0) {
            double tot = 0.0;
            for (int w = 0; w < SF_BS / 64; ++w) tot += s_red[w];
            *a.mean_cost = (float)(tot / (double)a.E) / (float)a.T;
        }
    }
}

extern "C" size_t reart_screw_fit_workspace_bytes(int T, int E) {
    (void)T;
    return E > 0 ? (size_t)E * 6 * sizeof(float) : 0;
}

extern "C" int reart_screw_fit(const float *trans, int T, int P, const int32_t *pairs, int E, int plain_mean,
                               float *screw, float *rel, float *mean, float *recon, float *cost, float *mean_cost,
                               void *workspace, size_t workspace_bytes, void *stream) {
    if (!trans || !cost || T <= 0 || P <= 0 || E < 0) return REART_ERR_INVALID_ARG;
    if (E == 0) return REART_OK;
    if (!mean) {
        if (!workspace || workspace_bytes < reart_screw_fit_workspace_bytes(T, E)) return REART_ERR_INVALID_ARG;
        mean = (float *)workspace;
    }
    ScrewFitArgs a{trans, pairs, T, P, E, plain_mean, screw, rel, mean, recon, cost, mean_cost};
    hipLaunchKernelGGL(screw_fit_kernel, dim3(1), dim3(SF_BS), 0, (hipStream_t)stream, a);
    REART_CHECK_LAUNCH();
    return REART_OK;
}

// ------------------------------------------------------------------------------------------------------------
// Per-part farthest point sampling: one workgroup per part compacts the part's member indices (ascending) into
// LDS and runs num_fps rounds on them.  Start = the part's first member (the reference's CUDA FPS starts at 0);
// ties: first maximum (torch.max) or, cuda_mode, the CUDA kernel's tree order for a block of opt_n_threads(n).
#define PF_BS 256

__device__ __forceinline__ bool pf_better(float v2, int i2, float v, int i, int cuda_mode, int bsmask) {
    if (v2 > v) return true;
    if (v2 < v) return false;
    if (cuda_mode) {
        const int t2 = i2 & bsmask, t = i & bsmask;
        return (t2 < t) || (t2 == t && i2 < i);
    }
    return i2 < i;
}

__global__ __launch_bounds__(PF_BS) void part_fps_kernel(const float *__restrict__ cano, const int64_t *__restrict__ seg,
                                                         int N, const int64_t *__restrict__ labels, int F, int cuda_mode,
                                                         int64_t *__restrict__ idx, int *__restrict__ count) {
    extern __shared__ __attribute__((aligned(16))) int s_dyn[];
    int *s_mem = s_dyn;                    // [N] member indices
    float *s_dm = (float *)(s_dyn + N);    // [N] running min distance
    __shared__ int s_cnt[PF_BS + 1];
    __shared__ float s_v[2][PF_BS / 64];
    __shared__ int s_i[2][PF_BS / 64];
    const int p = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t lab = labels[p];
    const int chunk = (N + PF_BS - 1) / PF_BS, k0 = tid * chunk, k1 = min(N, k0 + chunk);
    int c = 0;
    for (int k = k0; k < k1; ++k) c += seg[k] == lab;
    s_cnt[tid + 1] = c;
    if (tid == 0) s_cnt[0] = 0;
    __syncthreads();
    if (tid == 0)
        for (int i = 1; i <= PF_BS; ++i) s_cnt[i] += s_cnt[i - 1];
    __syncthreads();
    int w = s_cnt[tid];
    for (int k = k0; k < k1; ++k)
        if (seg[k] == lab) { s_mem[w] = k; s_dm[w] = 1e10f; ++w; }
    const int n = s_cnt[PF_BS];
    if (tid == 0) count[p] = n;
    __syncthreads();
    if (n < F) {   // fps_sample_cano raises; the host reads count[] and raises the same error
        for (int it = tid; it < F; it += PF_BS) idx[(size_t)p * F + it] = -1;
        return;
    }
    int bsmask = 1;
    while (bsmask * 2 <= n) bsmask *= 2;
    bsmask = min(bsmask, 1024) - 1;
    int far = 0;
    for (int it = 0; it < F; ++it) {
        if (tid == 0) idx[(size_t)p * F + it] = s_mem[far];
        if (it == F - 1) break;
        const float *f = cano + 3 * (size_t)s_mem[far];
        const float fx = f[0], fy = f[1], fz = f[2];
        float bv = -INFINITY;
        int bi = 0x7fffffff;
        for (int k = tid; k < n; k += PF_BS) {
            const float *q = cano + 3 * (size_t)s_mem[k];
            const float d = reart_sqdist3(q[0], q[1], q[2], fx, fy, fz);
            const float dm = d < s_dm[k] ? d : s_dm[k];
            s_dm[k] = dm;
            if (pf_better(dm, k, bv, bi, cuda_mode, bsmask)) { bv = dm; bi = k; }
        }
        for (int o = 32; o >= 1; o >>= 1) {
            const float v2 = __shfl_xor(bv, o, 64);
            const int i2 = __shfl_xor(bi, o, 64);
            if (pf_better(v2, i2, bv, bi, cuda_mode, bsmask)) { bv = v2; bi = i2; }
        }
        const int buf = it & 1;
        if (lane == 0) { s_v[buf][wv] = bv; s_i[buf][wv] = bi; }
        __syncthreads();
        bv = s_v[buf][0]; bi = s_i[buf][0];
#pragma unroll
        for (int q = 1; q < PF_BS / 64; ++q)
            if (pf_better(s_v[buf][q], s_i[buf][q], bv, bi, cuda_mode, bsmask)) { bv = s_v[buf][q]; bi = s_i[buf][q]; }
        far = bi;
    }
}

extern "C" int reart_part_fps(const float *cano, const int64_t *seg, int N, const int64_t *labels, int Ps, int num_fps,
                              int cuda_mode, int64_t *idx, int32_t *count, void *stream) {
    if (!cano || !seg || !labels || !idx || !count || N <= 0 || Ps < 0 || num_fps <= 0) return REART_ERR_INVALID_ARG;
    if (Ps == 0) return REART_OK;
    const size_t lds = (size_t)N * 8;
    if (lds > 150 * 1024) return REART_ERR_UNSUPPORTED;
    if (lds > REART_LDS_DEFAULT_CAP &&
        hipFuncSetAttribute((const void *)part_fps_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024) != hipSuccess)
        return REART_ERR_LAUNCH;
    hipLaunchKernelGGL(part_fps_kernel, dim3(Ps), dim3(PF_BS), lds, (hipStream_t)stream, cano, seg, N, labels, num_fps,
                       cuda_mode, idx, count);
    REART_CHECK_LAUNCH();
    return REART_OK;
}

// ------------------------------------------------------------------------------------------------------------
// All ordered part pairs (i, j): closest pair of the two parts' FPS sets in the canonical frame (first minimum:
// per source point the nearest target with the lowest index, then the first source point attaining the minimum)
// and that pair's squared distance summed over the predicted frames.
__global__ __launch_bounds__(256) void part_pair_cost_kernel(const float *__restrict__ cano_fps, const float *__restrict__ frame_fps,
                                                             int T, int Ps, int F, float *__restrict__ cano_dist,
                                                             int64_t *__restrict__ pair, float *__restrict__ joint) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= Ps * Ps) return;
    const int i = e / Ps, j = e % Ps;
    const float *a = cano_fps + (size_t)i * F * 3, *b = cano_fps + (size_t)j * F * 3;
    float best = INFINITY;
    int bs = 0, bt = 0;
    for (int s = 0; s < F; ++s) {
        const float ax = a[3 * s], ay = a[3 * s + 1], az = a[3 * s + 2];
        float dm = INFINITY;
        int tm = 0;
        for (int t = 0; t < F; ++t) {
            const float d = reart_sqdist3(ax, ay, az, b[3 * t], b[3 * t + 1], b[3 * t + 2]);
            if (d < dm) { dm = d; tm = t; }
        }
        if (dm < best) { best = dm; bs = s; bt = tm; }
    }
    cano_dist[e] = best;
    if (pair) { pair[2 * (size_t)e] = bs; pair[2 * (size_t)e + 1] = bt; }
    if (joint && frame_fps) {
        float acc = 0.f;
        for (int t = 0; t < T; ++t) {
            const float *pa = frame_fps + (((size_t)t * Ps + i) * F + bs) * 3;
            const float *pb = frame_fps + (((size_t)t * Ps + j) * F + bt) * 3;
            acc += reart_sqdist3(pa[0], pa[1], pa[2], pb[0], pb[1], pb[2]);
        }
        joint[e] = acc;
    }
}

extern "C" int reart_part_pair_cost(const float *cano_fps, const float *frame_fps, int T, int Ps, int F, float *cano_dist,
                                    int64_t *pair, float *joint, void *stream) {
    if (!cano_fps || !cano_dist || Ps < 0 || F <= 0 || T < 0) return REART_ERR_INVALID_ARG;
    if (Ps == 0) return REART_OK;
    hipLaunchKernelGGL(part_pair_cost_kernel, dim3((Ps * Ps + 255) / 256), dim3(256), 0, (hipStream_t)stream, cano_fps,
                       frame_fps, T, Ps, F, cano_dist, pair, joint);
    REART_CHECK_LAUNCH();
    return REART_OK;
}

// ------------------------------------------------------------------------------------------------------------
// compute_group_temporal_err: per part, mean over frames and member points of the squared distance to the part's
// per-frame centroid; the energy term is the maximum over parts.
__global__ __launch_bounds__(256) void group_err_kernel(const float *__restrict__ pcs, int T, int N, const int64_t *__restrict__ seg,
                                                        const int64_t *__restrict__ labels, float *__restrict__ per_part,
                                                        unsigned *__restrict__ worst_bits) {
    __shared__ double s_r[4][4];
    __shared__ float s_c[3];
    const int p = blockIdx.x, tid = threadIdx.x;
    const int64_t lab = labels[p];
    double tot = 0.0;
    int cnt = 0;
    for (int t = 0; t < T; ++t) {
        const float *f = pcs + (size_t)t * N * 3;
        double sx = 0, sy = 0, sz = 0, c = 0;
        for (int k = tid; k < N; k += 256)
            if (seg[k] == lab) { sx += f[3 * k]; sy += f[3 * k + 1]; sz += f[3 * k + 2]; c += 1; }
        for (int o = 32; o >= 1; o >>= 1) {
            sx += __shfl_xor(sx, o, 64); sy += __shfl_xor(sy, o, 64); sz += __shfl_xor(sz, o, 64); c += __shfl_xor(c, o, 64);
        }
        if ((tid & 63) == 0) { s_r[tid >> 6][0] = sx; s_r[tid >> 6][1] = sy; s_r[tid >> 6][2] = sz; s_r[tid >> 6][3] = c; }
        __syncthreads();
        if (tid == 0) {
            double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
            for (int w = 0; w < 4; ++w) { a0 += s_r[w][0]; a1 += s_r[w][1]; a2 += s_r[w][2]; a3 += s_r[w][3]; }
            s_c[0] = (float)(a0 / a3); s_c[1] = (float)(a1 / a3); s_c[2] = (float)(a2 / a3);
            s_r[0][3] = a3;
        }
        __syncthreads();
        const float cx = s_c[0], cy = s_c[1], cz = s_c[2];
        cnt = (int)s_r[0][3];
        double acc = 0;
        for (int k = tid; k < N; k += 256)
            if (seg[k] == lab) acc += (double)reart_sqdist3(f[3 * k], f[3 * k + 1], f[3 * k + 2], cx, cy, cz);
        tot += acc;
        __syncthreads();
    }
    for (int o = 32; o >= 1; o >>= 1) tot += __shfl_xor(tot, o, 64);
    if ((tid & 63) == 0) s_r[tid >> 6][0] = tot;
    __syncthreads();
    if (tid == 0) {
        const double all = ((s_r[0][0] + s_r[1][0]) + s_r[2][0]) + s_r[3][0];
        const float v = cnt > 0 ? (float)(all / ((double)T * (double)cnt)) : 0.f;
        per_part[p] = v;
        atomicMax(worst_bits, __float_as_uint(v));   // v >= 0: the bit pattern orders like the value
    }
}

extern "C" int reart_group_temporal_err(const float *pcs, int T, int N, const int64_t *seg, const int64_t *labels, int Ps,
                                        float *per_part, float *worst, void *stream) {
    if (!pcs || !seg || !labels || !per_part || !worst || T <= 0 || N <= 0 || Ps <= 0) return REART_ERR_INVALID_ARG;
    if (hipMemsetAsync(worst, 0, sizeof(float), (hipStream_t)stream) != hipSuccess) return REART_ERR_LAUNCH;
    hipLaunchKernelGGL(group_err_kernel, dim3(Ps), dim3(256), 0, (hipStream_t)stream, pcs, T, N, seg, labels, per_part,
                       (unsigned *)worst);
    REART_CHECK_LAUNCH();
    return REART_OK;
}
