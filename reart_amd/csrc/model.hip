// reart_amd/csrc/model.hip -- relaxation-model kernels for gfx950.
//
// Replaces, fused, the per-iteration PyTorch graph of the reference's
//   BaseModel.forward            networks/model.py:39-70   (seg head networks/blocks.py:99-118,
//                                F.gumbel_softmax(hard=True), rotation_6d_to_matrix
//                                screw_se3/geo_utils.py:632-651, bmm + weighted sum :63-69)
// and its autograd backward, plus compute_pc_transform (utils/model_utils.py:54-67) and the
// Adam update (run_robot.py:145-151,219-221).
//
// The reference materialises [(T-1)*P, N, 3] three times per forward (18.7 MB each at
// T=20); here the forward is one pass: per point, logits -> Gumbel-softmax -> selected part
// -> B rigid transforms, ~1.3 MB of HBM traffic.  All of this is launch/latency bound
// (a few MFLOP); the layout goal is few launches and deterministic reductions.
//
// Internal layouts (caller-owned "saved" buffers): yT [P][N] soft assignment, hT [H][N]
// hidden activations, hard_idx [N] sampled part.
#include "common.h"
#include "internal.h"
#include <math.h>

#define MB_BS 64        // threads per workgroup in the per-point kernels
#define RED_CHUNK 64    // points per partial-reduction chunk
#define RED_BS 256

// ------------------------------------------------------------------------------- helpers
__device__ __forceinline__ float dot3f(const float *a, const float *b) {
    return fmaf(a[2], b[2], fmaf(a[1], b[1], a[0] * b[0]));
}
__device__ __forceinline__ void cross3f(const float *a, const float *b, float *c) {
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}
__device__ __forceinline__ float norm3f(const float *a) {
    return sqrtf((a[0] * a[0] + a[1] * a[1]) + a[2] * a[2]);
}

// screw_se3/geo_utils.py:632-651 (rows b1,b2,b3); same operation order as oracle/model.c
__device__ __forceinline__ void r6d_to_matrix(const float *d6, float *R) {
    const float *a1 = d6, *a2 = d6 + 3;
    const float n1 = fmaxf(norm3f(a1), 1e-12f);
    float b1[3] = {a1[0] / n1, a1[1] / n1, a1[2] / n1};
    const float d = (b1[0] * a2[0] + b1[1] * a2[1]) + b1[2] * a2[2];
    float u[3] = {a2[0] - d * b1[0], a2[1] - d * b1[1], a2[2] - d * b1[2]};
    const float n2 = fmaxf(norm3f(u), 1e-12f);
    float b2[3] = {u[0] / n2, u[1] / n2, u[2] / n2};
    float b3[3];
    cross3f(b1, b2, b3);
#pragma unroll
    for (int c = 0; c < 3; ++c) { R[c] = b1[c]; R[3 + c] = b2[c]; R[6 + c] = b3[c]; }
}

__device__ __forceinline__ void r6d_backward(const float *d6, const float *gR, float *g6) {
    const float *a1 = d6, *a2 = d6 + 3;
    const float n1r = norm3f(a1), n1 = fmaxf(n1r, 1e-12f);
    float b1[3] = {a1[0] / n1, a1[1] / n1, a1[2] / n1};
    const float d = (b1[0] * a2[0] + b1[1] * a2[1]) + b1[2] * a2[2];
    float u[3] = {a2[0] - d * b1[0], a2[1] - d * b1[1], a2[2] - d * b1[2]};
    const float n2r = norm3f(u), n2 = fmaxf(n2r, 1e-12f);
    float b2[3] = {u[0] / n2, u[1] / n2, u[2] / n2};
    float gb1[3] = {gR[0], gR[1], gR[2]}, gb2[3] = {gR[3], gR[4], gR[5]};
    const float gb3[3] = {gR[6], gR[7], gR[8]};
    float t[3];
    cross3f(b2, gb3, t);
    gb1[0] += t[0]; gb1[1] += t[1]; gb1[2] += t[2];
    cross3f(gb3, b1, t);
    gb2[0] += t[0]; gb2[1] += t[1]; gb2[2] += t[2];
    float gu[3];
    if (n2r > 1e-12f) {
        const float s = dot3f(b2, gb2);
#pragma unroll
        for (int c = 0; c < 3; ++c) gu[c] = (gb2[c] - b2[c] * s) / n2;
    } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) gu[c] = gb2[c] / n2;
    }
    float ga2[3] = {gu[0], gu[1], gu[2]};
    const float gd = -dot3f(gu, b1);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        gb1[c] += -d * gu[c] + gd * a2[c];
        ga2[c] += gd * b1[c];
    }
    if (n1r > 1e-12f) {
        const float s = dot3f(b1, gb1);
#pragma unroll
        for (int c = 0; c < 3; ++c) g6[c] = (gb1[c] - b1[c] * s) / n1;
    } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) g6[c] = gb1[c] / n1;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) g6[3 + c] = ga2[c];
}

// v = R x + t with R row-major 3x3 (fmaf chain in ascending column order)
__device__ __forceinline__ void apply_rt(const float *Rt /*[12]: R(9) t(3)*/, float x0, float x1,
                                         float x2, float *v) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float acc = x0 * Rt[3 * c];
        acc = fmaf(x1, Rt[3 * c + 1], acc);
        acc = fmaf(x2, Rt[3 * c + 2], acc);
        v[c] = acc + Rt[9 + c];
    }
}

// Philox4x32-10 counter-based generator (Salmon et al. 2011) for the in-kernel Gumbel noise
__device__ __forceinline__ void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                           uint32_t k0, uint32_t k1, uint32_t *out) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
// -log(Exp(1)) sample: u in (0,1) -> e = -log u -> g = -log e   (F.gumbel_softmax recipe).
// 23 random bits + 0.5: every value (k + 0.5) * 2^-23 is exactly representable, so u never
// rounds to 1.0 (which would give e = 0, g = +inf and a NaN softmax once in 2^24 draws).
__device__ __forceinline__ float gumbel_from_bits(uint32_t bits) {
    const float u = ((float)(bits >> 9) + 0.5f) * (1.0f / 8388608.0f);
    return -logf(-logf(u));
}

// ------------------------------------------------------------------------------- forward

template <int PP>
__global__ __launch_bounds__(MB_BS) void base_fwd_kernel(BaseFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float s_rt[];  // [B*P][12]
    const int tid = threadIdx.x;
    const int P = (PP > 0) ? PP : a.P;
    for (int e = tid; e < a.B * a.P; e += MB_BS) {
        float R[9];
        r6d_to_matrix(a.p6d + 6 * (size_t)e, R);
#pragma unroll
        for (int c = 0; c < 9; ++c) s_rt[12 * e + c] = R[c];
#pragma unroll
        for (int c = 0; c < 3; ++c) s_rt[12 * e + 9 + c] = a.pt[3 * (size_t)e + c];
        if (blockIdx.x == 0 && a.trans_list) {
            float *T = a.trans_list + 16 * (size_t)e;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
#pragma unroll
                for (int c = 0; c < 3; ++c) T[4 * r + c] = R[3 * r + c];
                T[4 * r + 3] = a.pt[3 * (size_t)e + r];
            }
            T[12] = 0.f; T[13] = 0.f; T[14] = 0.f; T[15] = 1.f;
        }
    }
    __syncthreads();

    const int n = blockIdx.x * MB_BS + tid;
    const bool live = n < a.N;
    const int nc = live ? n : a.N - 1;
    const float x0 = a.cano[3 * (size_t)nc], x1 = a.cano[3 * (size_t)nc + 1], x2 = a.cano[3 * (size_t)nc + 2];

    constexpr int PMAX = (PP > 0) ? PP : 32;
    float s[PMAX];
#pragma unroll
    for (int p = 0; p < PMAX; ++p) s[p] = 0.f;
    for (int j = 0; j < a.H; ++j) {
        float acc = a.W1[3 * j] * x0;
        acc = fmaf(a.W1[3 * j + 1], x1, acc);
        acc = fmaf(a.W1[3 * j + 2], x2, acc);
        acc = acc + a.b1[j];
        const float h = acc > 0.f ? acc : 0.f;
        if (a.hT && live) a.hT[(size_t)j * a.N + n] = h;
#pragma unroll
        for (int p = 0; p < PMAX; ++p)
            if (PP > 0 || p < P) s[p] = fmaf(a.W2[(size_t)p * a.H + j], h, s[p]);
    }
    // noise-free arg-max (networks/model.py:70) -- first maximum
    int am = 0;
    float sm = s[0];
#pragma unroll
    for (int p = 1; p < PMAX; ++p)
        if ((PP > 0 || p < P) && s[p] > sm) { sm = s[p]; am = p; }

    const float tau = a.tau_ptr ? a.tau_ptr[0] : a.tau;
    float g[PMAX];
    if (a.gumbel) {
#pragma unroll
        for (int p = 0; p < PMAX; ++p)
            g[p] = (PP > 0 || p < P) ? a.gumbel[(size_t)nc * P + p] : 0.f;
    } else {
        const uint64_t it = a.iter_ptr ? (uint64_t)a.iter_ptr[0] : 0ull;
#pragma unroll
        for (int q = 0; q < PMAX; q += 4) {
            uint32_t r[4];
            philox4x32((uint32_t)n, (uint32_t)(q >> 2), (uint32_t)it, (uint32_t)(it >> 32),
                       (uint32_t)a.seed, (uint32_t)(a.seed >> 32), r);
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (q + u < PMAX) g[q + u] = gumbel_from_bits(r[u]);
        }
    }
    float z[PMAX];
    float m = -INFINITY;
#pragma unroll
    for (int p = 0; p < PMAX; ++p)
        if (PP > 0 || p < P) { z[p] = (s[p] + g[p]) / tau; m = fmaxf(m, z[p]); }
    float sum = 0.f;
#pragma unroll
    for (int p = 0; p < PMAX; ++p)
        if (PP > 0 || p < P) { z[p] = expf(z[p] - m); sum += z[p]; }
    int k = 0;
    float yk = -1.f;
#pragma unroll
    for (int p = 0; p < PMAX; ++p)
        if (PP > 0 || p < P) {
            z[p] = z[p] / sum;
            if (z[p] > yk) { yk = z[p]; k = p; }
            if (a.yT && live) a.yT[(size_t)p * a.N + n] = z[p];
        }
    const float w = (1.0f - yk) + yk;  // y_hard - y_soft.detach() + y_soft
    if (live) {
        if (a.seg_part) a.seg_part[n] = am;
        if (a.hard_idx) a.hard_idx[n] = k;
    }
    for (int t = 0; t < a.B; ++t) {
        float v[3];
        apply_rt(s_rt + 12 * (t * a.P + k), x0, x1, x2, v);
        v[0] = w * v[0]; v[1] = w * v[1]; v[2] = w * v[2];
        if (live) {
            float *o = a.out + 3 * ((size_t)t * a.N + n);
            o[0] = v[0]; o[1] = v[1]; o[2] = v[2];
        }
        if (a.out_soa && n < a.Npad) {
            float *o = a.out_soa + (size_t)t * 3 * a.Npad;
            o[n] = live ? v[0] : INFINITY;
            o[a.Npad + n] = live ? v[1] : INFINITY;
            o[2 * (size_t)a.Npad + n] = live ? v[2] : INFINITY;
        }
    }
}

template <int PP>
static void launch_base_fwd(const BaseFwdArgs &a, hipStream_t st) {
    const int cover = a.out_soa ? (a.Npad > a.N ? a.Npad : a.N) : a.N;
    const size_t lds = sizeof(float) * 12 * (size_t)a.B * a.P;
    hipLaunchKernelGGL((base_fwd_kernel<PP>), dim3(reart_div_up(cover, MB_BS)), dim3(MB_BS), lds, st, a);
}

static int dispatch_base_fwd(const BaseFwdArgs &a, hipStream_t st) {
    if (a.P < 1 || a.P > 32) return REART_ERR_UNSUPPORTED;
    if ((size_t)a.B * a.P * 12 * sizeof(float) > 64 * 1024) return REART_ERR_UNSUPPORTED;
    switch (a.P) {
        case 20: launch_base_fwd<20>(a, st); break;
        case 10: launch_base_fwd<10>(a, st); break;
        case 8: launch_base_fwd<8>(a, st); break;
        default: launch_base_fwd<0>(a, st); break;
    }
    REART_CHECK_LAUNCH();
    return REART_OK;
}

extern "C" int reart_base_forward(const float *cano, int N, int P, int B, const float *W1,
                                  const float *b1, const float *W2, int H, const float *prop6d,
                                  const float *propt, const float *gumbel, float tau, float *out,
                                  int64_t *seg_part, float *trans_list, float *yT, float *hT,
                                  int32_t *hard_idx, void *stream) {
    if (N < 0 || P < 1 || B < 0 || H < 1) return REART_ERR_INVALID_ARG;
    if (N == 0 || B == 0) return REART_OK;
    if (!cano || !W1 || !b1 || !W2 || !prop6d || !propt || !gumbel || !out) return REART_ERR_INVALID_ARG;
    if (!(tau > 0.f)) return REART_ERR_INVALID_ARG;
    BaseFwdArgs a = {};
    a.cano = cano; a.W1 = W1; a.b1 = b1; a.W2 = W2; a.p6d = prop6d; a.pt = propt;
    a.gumbel = gumbel; a.tau = tau; a.N = N; a.P = P; a.B = B; a.H = H; a.Npad = 0;
    a.out = out; a.seg_part = seg_part; a.trans_list = trans_list; a.yT = yT; a.hT = hT;
    a.hard_idx = hard_idx;
    return dispatch_base_fwd(a, (hipStream_t)stream);
}

// entry used by the fused step (step.hip)
int reart_base_forward_ex(const BaseFwdArgs &a, hipStream_t st) { return dispatch_base_fwd(a, st); }

// ------------------------------------------------------------------------------- backward
// (1) per point: dL/dw (dense in p) -> softmax backward -> ds; hidden-layer gradient.
// layout of one partial row / of the reduced gradient vector
__host__ __device__ static inline int off_gW2() { return 0; }
__host__ __device__ static inline int off_gW1(int P, int H) { return P * H; }
__host__ __device__ static inline int off_gb1(int P, int H) { return P * H + 3 * H; }
__host__ __device__ static inline int off_gRt(int P, int H) { return P * H + 4 * H; }
__host__ __device__ static inline int n_out(int P, int H, int B) { return P * H + 4 * H + 12 * B * P; }

template <int PP>
__global__ __launch_bounds__(MB_BS) void base_bwd_point_kernel(BaseBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float s_rt[];
    const int tid = threadIdx.x;
    const int P = (PP > 0) ? PP : a.P;
    for (int e = tid; e < a.B * a.P; e += MB_BS) {
        float R[9];
        r6d_to_matrix(a.p6d + 6 * (size_t)e, R);
#pragma unroll
        for (int c = 0; c < 9; ++c) s_rt[12 * e + c] = R[c];
#pragma unroll
        for (int c = 0; c < 3; ++c) s_rt[12 * e + 9 + c] = a.pt[3 * (size_t)e + c];
    }
    __syncthreads();
    const int n = blockIdx.x * MB_BS + tid;
    if (n >= a.N) return;
    const float x0 = a.cano[3 * (size_t)n], x1 = a.cano[3 * (size_t)n + 1], x2 = a.cano[3 * (size_t)n + 2];
    constexpr int PMAX = (PP > 0) ? PP : 32;
    float dw[PMAX];
#pragma unroll
    for (int p = 0; p < PMAX; ++p) dw[p] = 0.f;
    for (int t = 0; t < a.B; ++t) {
        const float *g = a.G + 3 * ((size_t)t * a.N + n);
        const float gv[3] = {g[0], g[1], g[2]};
#pragma unroll
        for (int p = 0; p < PMAX; ++p)
            if (PP > 0 || p < P) {
                float v[3];
                apply_rt(s_rt + 12 * (t * a.P + p), x0, x1, x2, v);
                dw[p] += dot3f(gv, v);
            }
    }
    const float tau = a.tau_ptr ? a.tau_ptr[0] : a.tau;
    float y[PMAX];
    float dot = 0.f;
#pragma unroll
    for (int p = 0; p < PMAX; ++p)
        if (PP > 0 || p < P) { y[p] = a.yT[(size_t)p * a.N + n]; dot = fmaf(y[p], dw[p], dot); }
#pragma unroll
    for (int p = 0; p < PMAX; ++p)
        if (PP > 0 || p < P) {
            dw[p] = (y[p] * (dw[p] - dot)) / tau;  // now ds[p]
            a.dsT[(size_t)p * a.N + n] = dw[p];
        }
    for (int j = 0; j < a.H; ++j) {
        float dh = 0.f;
#pragma unroll
        for (int p = 0; p < PMAX; ++p)
            if (PP > 0 || p < P) dh = fmaf(a.W2[(size_t)p * a.H + j], dw[p], dh);
        const float h = a.hT[(size_t)j * a.N + n];
        a.dpT[(size_t)j * a.N + n] = h > 0.f ? dh : 0.f;
    }
}

// (2) per chunk of RED_CHUNK points: partial sums of every parameter gradient, each output
// accumulated sequentially over the chunk's points (ascending n) -> deterministic.
//   gW2[p,j] = sum_n ds[n,p] h[n,j];  gW1[j,c] = sum_n dp[n,j] x[n,c];  gb1[j] = sum_n dp[n,j]
//   gR[t,p]  = sum_{n:k_n=p} w_n G[t,n] x_n^T;  gt[t,p] = sum_{n:k_n=p} w_n G[t,n]
__global__ __launch_bounds__(RED_BS) void base_bwd_reduce_kernel(BaseBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int LD = RED_CHUNK + 1;  // +1 pad: rows indexed by lane -> conflict-free columns
    float *s_h = smem;                         // [H][LD]   (hT tile, then dpT tile)
    float *s_ds = s_h + (size_t)a.H * LD;      // [P][LD]
    float *s_x = s_ds + (size_t)a.P * LD;      // [RED_CHUNK][3]
    float *s_w = s_x + RED_CHUNK * 3;          // [RED_CHUNK]
    int *s_k = (int *)(s_w + RED_CHUNK);       // [RED_CHUNK]
    float *s_acc = (float *)(s_k + RED_CHUNK); // [B*P*12]
    const int tid = threadIdx.x, chunk = blockIdx.x;
    const int n0 = chunk * RED_CHUNK;
    const int cn = (a.N - n0) < RED_CHUNK ? (a.N - n0) : RED_CHUNK;
    float *prow = a.partial + (size_t)chunk * n_out(a.P, a.H, a.B);

    for (int e = tid; e < a.H * RED_CHUNK; e += RED_BS) {
        const int j = e / RED_CHUNK, i = e % RED_CHUNK;
        s_h[j * LD + i] = (i < cn) ? a.hT[(size_t)j * a.N + n0 + i] : 0.f;
    }
    for (int e = tid; e < a.P * RED_CHUNK; e += RED_BS) {
        const int p = e / RED_CHUNK, i = e % RED_CHUNK;
        s_ds[p * LD + i] = (i < cn) ? a.dsT[(size_t)p * a.N + n0 + i] : 0.f;
    }
    for (int i = tid; i < RED_CHUNK; i += RED_BS) {
        const bool ok = i < cn;
        const int n = ok ? n0 + i : n0;
        const int k = a.hard_idx[n];
        const float yk = a.yT[(size_t)k * a.N + n];
        s_k[i] = ok ? k : -1;
        s_w[i] = (1.0f - yk) + yk;
        s_x[3 * i] = ok ? a.cano[3 * (size_t)n] : 0.f;
        s_x[3 * i + 1] = ok ? a.cano[3 * (size_t)n + 1] : 0.f;
        s_x[3 * i + 2] = ok ? a.cano[3 * (size_t)n + 2] : 0.f;
    }
    for (int e = tid; e < a.B * a.P * 12; e += RED_BS) s_acc[e] = 0.f;
    __syncthreads();
    // gW2
    for (int o = tid; o < a.P * a.H; o += RED_BS) {
        const int p = o / a.H, j = o % a.H;
        float acc = 0.f;
        for (int i = 0; i < RED_CHUNK; ++i) acc = fmaf(s_ds[p * LD + i], s_h[j * LD + i], acc);
        prow[off_gW2() + o] = acc;
    }
    // gR | gt : thread (t, c) walks the chunk, scattering into its own LDS column
    for (int o = tid; o < a.B * 12; o += RED_BS) {
        const int t = o / 12, c = o % 12;
        for (int i = 0; i < cn; ++i) {
            const int k = s_k[i];
            const float *g = a.G + 3 * ((size_t)t * a.N + n0 + i);
            float v;
            if (c < 9) v = (s_w[i] * g[c / 3]) * s_x[3 * i + c % 3];
            else v = s_w[i] * g[c - 9];
            s_acc[(t * a.P + k) * 12 + c] += v;
        }
    }
    __syncthreads();
    for (int e = tid; e < a.B * a.P * 12; e += RED_BS) prow[off_gRt(a.P, a.H) + e] = s_acc[e];
    // reload the tile with dpT for gW1 / gb1
    __syncthreads();
    for (int e = tid; e < a.H * RED_CHUNK; e += RED_BS) {
        const int j = e / RED_CHUNK, i = e % RED_CHUNK;
        s_h[j * LD + i] = (i < cn) ? a.dpT[(size_t)j * a.N + n0 + i] : 0.f;
    }
    __syncthreads();
    for (int o = tid; o < 4 * a.H; o += RED_BS) {
        const int j = o / 4, c = o % 4;
        float acc = 0.f;
        if (c < 3) {
            for (int i = 0; i < RED_CHUNK; ++i) acc = fmaf(s_h[j * LD + i], s_x[3 * i + c], acc);
            prow[off_gW1(a.P, a.H) + 3 * j + c] = acc;
        } else {
            for (int i = 0; i < RED_CHUNK; ++i) acc += s_h[j * LD + i];
            prow[off_gb1(a.P, a.H) + j] = acc;
        }
    }
}

// (3) sum the chunk partials in ascending chunk order; Gram-Schmidt backward for the 6-vectors
__global__ __launch_bounds__(256) void base_bwd_finalize_kernel(BaseBwdArgs a) {
    const int o = blockIdx.x * 256 + threadIdx.x;
    const int nW = a.P * a.H + 4 * a.H;
    const int no = n_out(a.P, a.H, a.B);
    if (o < nW) {
        float acc = 0.f;
        for (int c = 0; c < a.nchunk; ++c) acc += a.partial[(size_t)c * no + o];
        if (o < a.P * a.H) a.gW2[o] = acc;
        else if (o < a.P * a.H + 3 * a.H) a.gW1[o - a.P * a.H] = acc;
        else a.gb1[o - a.P * a.H - 3 * a.H] = acc;
    } else if (o < nW + a.B * a.P) {
        const int e = o - nW;
        float gRt[12];
#pragma unroll
        for (int c = 0; c < 12; ++c) gRt[c] = 0.f;
        for (int ch = 0; ch < a.nchunk; ++ch) {
            const float *pr = a.partial + (size_t)ch * no + off_gRt(a.P, a.H) + 12 * e;
#pragma unroll
            for (int c = 0; c < 12; ++c) gRt[c] += pr[c];
        }
        float g6[6];
        r6d_backward(a.p6d + 6 * (size_t)e, gRt, g6);
#pragma unroll
        for (int c = 0; c < 6; ++c) a.g6d[6 * (size_t)e + c] = g6[c];
#pragma unroll
        for (int c = 0; c < 3; ++c) a.gt[3 * (size_t)e + c] = gRt[9 + c];
    }
}

static size_t base_bwd_ws_layout(int N, int P, int B, int H, size_t *o_ds, size_t *o_dp, size_t *o_part) {
    const int nchunk = reart_div_up(N, RED_CHUNK);
    size_t off = 0;
    *o_ds = off; off += reart_align_up(sizeof(float) * (size_t)P * N, 256);
    *o_dp = off; off += reart_align_up(sizeof(float) * (size_t)H * N, 256);
    *o_part = off; off += reart_align_up(sizeof(float) * (size_t)nchunk * n_out(P, H, B), 256);
    return off;
}

extern "C" size_t reart_base_backward_workspace_bytes(int N, int P, int B, int H) {
    if (N <= 0 || P <= 0 || B <= 0 || H <= 0) return 0;
    size_t a, b, c;
    return base_bwd_ws_layout(N, P, B, H, &a, &b, &c);
}

int reart_base_backward_ex(BaseBwdArgs a, void *workspace, size_t workspace_bytes, hipStream_t st) {
    if (a.P > 32) return REART_ERR_UNSUPPORTED;
    size_t o_ds, o_dp, o_part;
    const size_t need = base_bwd_ws_layout(a.N, a.P, a.B, a.H, &o_ds, &o_dp, &o_part);
    if (!workspace || workspace_bytes < need) return REART_ERR_INVALID_ARG;
    char *ws = (char *)workspace;
    a.dsT = (float *)(ws + o_ds); a.dpT = (float *)(ws + o_dp); a.partial = (float *)(ws + o_part);
    a.nchunk = reart_div_up(a.N, RED_CHUNK);
    const size_t lds1 = sizeof(float) * 12 * (size_t)a.B * a.P;
    const size_t lds2 = sizeof(float) * ((size_t)(a.H + a.P) * (RED_CHUNK + 1) + RED_CHUNK * 5 + (size_t)a.B * a.P * 12);
    if (lds1 > 64 * 1024 || lds2 > 64 * 1024) return REART_ERR_UNSUPPORTED;
    const dim3 g1(reart_div_up(a.N, MB_BS));
    switch (a.P) {
        case 20: hipLaunchKernelGGL((base_bwd_point_kernel<20>), g1, dim3(MB_BS), lds1, st, a); break;
        case 10: hipLaunchKernelGGL((base_bwd_point_kernel<10>), g1, dim3(MB_BS), lds1, st, a); break;
        case 8: hipLaunchKernelGGL((base_bwd_point_kernel<8>), g1, dim3(MB_BS), lds1, st, a); break;
        default: hipLaunchKernelGGL((base_bwd_point_kernel<0>), g1, dim3(MB_BS), lds1, st, a); break;
    }
    hipLaunchKernelGGL(base_bwd_reduce_kernel, dim3(a.nchunk), dim3(RED_BS), lds2, st, a);
    const int nfin = a.P * a.H + 4 * a.H + a.B * a.P;
    hipLaunchKernelGGL(base_bwd_finalize_kernel, dim3(reart_div_up(nfin, 256)), dim3(256), 0, st, a);
    REART_CHECK_LAUNCH();
    return REART_OK;
}

extern "C" int reart_base_backward(const float *cano, int N, int P, int B, const float *W1,
                                   const float *b1, const float *W2, int H, const float *prop6d,
                                   const float *propt, const float *yT, const float *hT,
                                   const int32_t *hard_idx, float tau, const float *G, float *gW1,
                                   float *gb1, float *gW2, float *g6d, float *gt, void *workspace,
                                   size_t workspace_bytes, void *stream) {
    (void)W1; (void)b1;
    if (N <= 0 || P < 1 || B <= 0 || H < 1) return REART_ERR_INVALID_ARG;
    if (!cano || !W2 || !prop6d || !propt || !yT || !hT || !hard_idx || !G || !gW1 || !gb1 || !gW2 ||
        !g6d || !gt)
        return REART_ERR_INVALID_ARG;
    BaseBwdArgs a = {};
    a.cano = cano; a.W2 = W2; a.p6d = prop6d; a.pt = propt; a.yT = yT; a.hT = hT;
    a.hard_idx = hard_idx; a.tau = tau; a.G = G; a.N = N; a.P = P; a.B = B; a.H = H;
    a.gW1 = gW1; a.gb1 = gb1; a.gW2 = gW2; a.g6d = g6d; a.gt = gt;
    return reart_base_backward_ex(a, workspace, workspace_bytes, (hipStream_t)stream);
}

// --------------------------------------------------------------- hard-label rigid apply
// utils/model_utils.py:54-67 (compute_pc_transform) and the apply in KinematicModel.forward
// (networks/model.py:161-165): out[t,n] = R[t,part_n] x_n + t[t,part_n], pose [B,P,4,4].
__global__ __launch_bounds__(256) void pc_transform_kernel(const float *__restrict__ cano,
                                                           const float *__restrict__ pose,
                                                           const int64_t *__restrict__ part, int N,
                                                           int P, int B, float *__restrict__ out) {
    const int n = blockIdx.x * 256 + threadIdx.x, t = blockIdx.y;
    if (n >= N) return;
    const float *T = pose + 16 * ((size_t)t * P + part[n]);
    const float x0 = cano[3 * (size_t)n], x1 = cano[3 * (size_t)n + 1], x2 = cano[3 * (size_t)n + 2];
    float *o = out + 3 * ((size_t)t * N + n);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float acc = x0 * T[4 * c];
        acc = fmaf(x1, T[4 * c + 1], acc);
        acc = fmaf(x2, T[4 * c + 2], acc);
        o[c] = acc + T[4 * c + 3];
    }
}

extern "C" int reart_compute_pc_transform(const float *cano, const float *pose, const int64_t *part,
                                          int N, int P, int B, float *out, void *stream) {
    if (N < 0 || P < 1 || B < 0) return REART_ERR_INVALID_ARG;
    if (N == 0 || B == 0) return REART_OK;
    if (!cano || !pose || !part || !out) return REART_ERR_INVALID_ARG;
    hipLaunchKernelGGL(pc_transform_kernel, dim3(reart_div_up(N, 256), B), dim3(256), 0,
                       (hipStream_t)stream, cano, pose, part, N, P, B, out);
    REART_CHECK_LAUNCH();
    return REART_OK;
}

extern "C" int reart_rotation_6d_to_matrix(const float *d6, int n, float *R, void *stream);
__global__ __launch_bounds__(256) void r6d_kernel(const float *__restrict__ d6, int n, float *__restrict__ R) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float r[9];
    r6d_to_matrix(d6 + 6 * (size_t)i, r);
#pragma unroll
    for (int c = 0; c < 9; ++c) R[9 * (size_t)i + c] = r[c];
}
extern "C" int reart_rotation_6d_to_matrix(const float *d6, int n, float *R, void *stream) {
    if (n < 0) return REART_ERR_INVALID_ARG;
    if (n == 0) return REART_OK;
    if (!d6 || !R) return REART_ERR_INVALID_ARG;
    hipLaunchKernelGGL(r6d_kernel, dim3(reart_div_up(n, 256)), dim3(256), 0, (hipStream_t)stream, d6, n, R);
    REART_CHECK_LAUNCH();
    return REART_OK;
}

// ------------------------------------------------------------------------------- Adam
// torch.optim.Adam single-tensor step (amsgrad off, weight_decay 0), up to 8 tensors per
// launch.  `step_ptr` (device int64, nullable) holds the number of steps ALREADY taken, so
// a captured graph needs no per-iteration host argument.

__global__ __launch_bounds__(256) void adam_kernel(AdamArgs a) {
    const AdamSeg s = a.seg[blockIdx.y];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= s.n) return;
    const int step = a.step_ptr ? (int)a.step_ptr[0] + 1 : a.step;
    const double bc1 = 1.0 - pow((double)a.beta1, (double)step);
    const double bc2 = 1.0 - pow((double)a.beta2, (double)step);
    const float step_size = (float)((double)s.lr / bc1);
    const float bc2s = (float)sqrt(bc2);
    const float g = s.g[i];
    float m = s.m[i], v = s.v[i];
    m = m + (g - m) * (1.0f - a.beta1);
    v = v * a.beta2 + ((1.0f - a.beta2) * g) * g;
    const float denom = sqrtf(v) / bc2s + a.eps;
    s.m[i] = m; s.v[i] = v;
    s.p[i] = s.p[i] - step_size * (m / denom);
}

int reart_adam_ex(const AdamArgs &a, hipStream_t st) {
    int maxn = 0;
    for (int k = 0; k < a.nseg; ++k) maxn = a.seg[k].n > maxn ? a.seg[k].n : maxn;
    if (maxn == 0) return REART_OK;
    hipLaunchKernelGGL(adam_kernel, dim3(reart_div_up(maxn, 256), a.nseg), dim3(256), 0, st, a);
    REART_CHECK_LAUNCH();
    return REART_OK;
}

extern "C" int reart_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq,
                               int n, int step, float lr, float beta1, float beta2, float eps,
                               void *stream) {
    if (n < 0 || step < 1) return REART_ERR_INVALID_ARG;
    if (n == 0) return REART_OK;
    if (!param || !grad || !exp_avg || !exp_avg_sq) return REART_ERR_INVALID_ARG;
    AdamArgs a = {};
    a.seg[0] = {param, grad, exp_avg, exp_avg_sq, n, lr};
    a.nseg = 1; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.step = step;
    return reart_adam_ex(a, (hipStream_t)stream);
}
