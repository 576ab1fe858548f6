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

#define RED_CHUNK 64    // points per partial-reduction chunk

// ------------------------------------------------------------------------------- helpers
#ifdef REART_PHASE_CLOCK   // diagnostic build only: shader-clock stamps of workgroup 0 per phase
__device__ unsigned long long g_phase_ts[2][16];
extern "C" int reart_debug_phase_clock(unsigned long long *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase_ts), sizeof(g_phase_ts)) == hipSuccess ? REART_OK : REART_ERR_LAUNCH;
}
// (slots 12 / 13 of a row: the constant-rate 100 MHz wall clock next to the first / latest stamp -- the rate of s_memtime)
#define PHASE_TS(which, k) do { if (blockIdx.x == 1 && threadIdx.x == 0) { g_phase_ts[which][k] = __builtin_amdgcn_s_memtime(); \
    g_phase_ts[which][(k) == 0 ? 12 : 13] = wall_clock64(); } } while (0)
#else
#define PHASE_TS(which, k) do { } while (0)
#endif
#ifdef REART_PHASE_CLOCK
#define PHASE_SYNC_TS(which, k) do { __syncthreads(); PHASE_TS(which, k); } while (0)   // serialises: attribution only
#else
#define PHASE_SYNC_TS(which, k) do { } while (0)
#endif
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
// Workgroup = W waves x the same 64 points, W = ceil(P/2): wave g owns parts {2g, 2g+1}.
//   * logits: each wave runs the full ascending-j fmaf chain of ITS parts (the oracle's
//     rounding order), weights broadcast from LDS;
//   * Gumbel noise, (s+g)/tau, exp and the division are evaluated only for the wave's own
//     parts; max / sum / arg-max meet in LDS (the sum is re-done by every lane in ascending
//     part order, again the oracle's order);
//   * the rigid apply of frame t is done by wave t mod W.
// The kernel is latency bound (a few thousand dependent instructions per wave at one wave per
// SIMD), so the design goal is the shortest per-wave instruction stream, not occupancy.
#define FW_PTS 64
#define FW_PG 2

// HALF: a workgroup owns 32 points and the two halves of a wave work on different part pairs (lane = half * 32 +
// point), so that the same chains run in W/2 waves per workgroup on twice as many workgroups (the kernel is bound by the
// serial work of one workgroup; 256 CUs take the extra workgroups for free).  Same arithmetic per (point, part).
template <int PP, bool HALF, bool BATCH>
__global__ __launch_bounds__(64 * (HALF ? (((PP > 0 ? PP : 32) + 2 * FW_PG - 1) / (2 * FW_PG)) : (((PP > 0 ? PP : 32) + FW_PG - 1) / FW_PG)))
void base_fwd_kernel(Batched<BaseFwdArgs> ab) {
    const BaseFwdArgs &a = ab.a[BATCH ? blockIdx.y : 0];      // a single instance reads its block at a fixed offset
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int PMAX = (PP > 0) ? PP : 32;
    constexpr int W = HALF ? (PMAX + 2 * FW_PG - 1) / (2 * FW_PG) : (PMAX + FW_PG - 1) / FW_PG;   // waves
    constexpr int NS = HALF ? 2 * W : W;                  // part-pair slices (one per wave, or per half wave)
    constexpr int PTS = HALF ? FW_PTS / 2 : FW_PTS;       // points per workgroup
    constexpr int BS = 64 * W;
    float *s_rt = smem;                                   // [B*P][12]
    float *s_wb = s_rt + 12 * (size_t)a.B * a.P;          // [H][4]  W1 row | b1
    float *s_w2T = s_wb + 4 * (size_t)a.H;                // [W][H][2]  W2 of wave g's two parts, j-major: 16-byte reads give two j
    float *s_a = s_w2T + (size_t)a.H * PMAX;              // [PMAX][64]  logits, later y
    float *s_e = s_a + PMAX * PTS;                        // [PMAX][PTS]  z, later exp(z - max)
    float *s_hh = s_e + PMAX * PTS;                       // [64][H + 4]  hidden activations, point-major: a lane reads four j at once
    const int HS = a.H + 4;                               // row stride (16-byte aligned rows, lanes spread over the banks)
    const int tid = threadIdx.x, lane = tid & 63, grp = tid >> 6;
    const int pl = HALF ? (lane & 31) : lane;             // point of this lane inside the workgroup
    const int sid = HALF ? 2 * grp + (lane >> 5) : grp;   // part-pair slice of this lane
    PHASE_TS(0, 0);
    const int P = (PP > 0) ? PP : a.P;
    // Prologue loads: the first batch of every group (pose parameters, W1|b1, one W2 column per thread) is
    // issued before anything is stored to LDS, so the prologue is one memory round trip deep
    const int nRT = a.B * a.P;
    auto put_rt = [&](int e, const float (&d6)[6], const float (&tv)[3]) {
        float R[9];
        r6d_to_matrix(d6, R);
#pragma unroll
        for (int c = 0; c < 9; ++c) s_rt[12 * e + c] = R[c];
#pragma unroll
        for (int c = 0; c < 3; ++c) s_rt[12 * e + 9 + c] = tv[c];
        if (blockIdx.x == 0) {
            if (a.trans_list) {
                float *T = a.trans_list + 16 * (size_t)e;
#pragma unroll
                for (int r = 0; r < 3; ++r) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) T[4 * r + c] = R[3 * r + c];
                    T[4 * r + 3] = tv[r];
                }
                T[12] = 0.f; T[13] = 0.f; T[14] = 0.f; T[15] = 1.f;
            }
            if (a.rt_table) {
#pragma unroll
                for (int c = 0; c < 9; ++c) a.rt_table[12 * (size_t)e + c] = R[c];
#pragma unroll
                for (int c = 0; c < 3; ++c) a.rt_table[12 * (size_t)e + 9 + c] = tv[c];
            }
        }
    };
    {
        float d6[6], tv[3], vw[PMAX];
        const int e0 = tid < nRT ? tid : 0;
#pragma unroll
        for (int c = 0; c < 6; ++c) d6[c] = a.p6d[6 * (size_t)e0 + c];
#pragma unroll
        for (int c = 0; c < 3; ++c) tv[c] = a.pt[3 * (size_t)e0 + c];
        const int ew = tid < 4 * a.H ? tid : 0;
        const float wbv = (ew & 3) < 3 ? a.W1[3 * (ew >> 2) + (ew & 3)] : a.b1[ew >> 2];
        const int jw = tid < a.H ? tid : 0;
#pragma unroll
        for (int p = 0; p < PMAX; ++p) vw[p] = a.W2[(size_t)(p < P ? p : 0) * a.H + jw];
        if (tid < nRT) put_rt(tid, d6, tv);
        if (tid < 4 * a.H) s_wb[tid] = wbv;
        if (tid < a.H) {
#pragma unroll
            for (int p = 0; p < PMAX; ++p) s_w2T[((p >> 1) * a.H + tid) * 2 + (p & 1)] = p < P ? vw[p] : 0.f;
        }
    }
    for (int e = tid + BS; e < nRT; e += BS) {
        float d6[6], tv[3];
#pragma unroll
        for (int c = 0; c < 6; ++c) d6[c] = a.p6d[6 * (size_t)e + c];
#pragma unroll
        for (int c = 0; c < 3; ++c) tv[c] = a.pt[3 * (size_t)e + c];
        put_rt(e, d6, tv);
    }
    for (int e = tid + BS; e < 4 * a.H; e += BS) {
        const int j = e >> 2, c = e & 3;
        s_wb[e] = c < 3 ? a.W1[3 * j + c] : a.b1[j];
    }
    for (int j = tid + BS; j < a.H; j += BS)          // one hidden unit per thread: no integer division
        for (int p = 0; p < PMAX; ++p) s_w2T[((p >> 1) * a.H + j) * 2 + (p & 1)] = p < P ? a.W2[(size_t)p * a.H + j] : 0.f;

    const int n = blockIdx.x * PTS + pl;
    const bool live = n < a.N;
    const int nc = live ? n : a.N - 1;
    const float x0 = a.cano[3 * (size_t)nc], x1 = a.cano[3 * (size_t)nc + 1], x2 = a.cano[3 * (size_t)nc + 2];
    const int p0 = sid * FW_PG;
    const bool has1 = p0 + 1 < P;
    const float tau = a.tau_ptr ? a.tau_ptr[0] : a.tau;
    // Gumbel noise of this wave's parts (issued early: independent of the logits)
    float g0, g1 = 0.f;
    if (a.gumbel) {
        g0 = a.gumbel[(size_t)nc * P + (p0 < P ? p0 : 0)];
        if (has1) g1 = a.gumbel[(size_t)nc * P + p0 + 1];
    } else {
        const uint64_t it = a.iter_ptr ? (uint64_t)a.iter_ptr[0] : 0ull;
        uint32_t r[4];
        philox4x32((uint32_t)n, (uint32_t)(p0 >> 2), (uint32_t)it, (uint32_t)(it >> 32), (uint32_t)a.seed,
                   (uint32_t)(a.seed >> 32), r);
        g0 = gumbel_from_bits(r[p0 & 3]);
        g1 = gumbel_from_bits(r[(p0 & 3) + 1]);   // p0 is even: p0 & 3 in {0, 2}
    }
    __syncthreads();
    PHASE_TS(0, 1);
    // hidden layer once per point: wave g evaluates its slice of the H units for the 64 points
    {
        const int jq = (a.H + NS - 1) / NS, j0 = sid * jq, j1 = (j0 + jq < a.H) ? j0 + jq : a.H;
        for (int j = j0; j < j1; ++j) {
            const float4 wb = *(const float4 *)(s_wb + 4 * j);
            float acc = wb.x * x0;
            acc = fmaf(wb.y, x1, acc);
            acc = fmaf(wb.z, x2, acc);
            acc = acc + wb.w;
            const float h = acc > 0.f ? acc : 0.f;
            s_hh[pl * HS + j] = h;
            if (a.hT && live) a.hT[(size_t)j * a.N + n] = h;
        }
    }
    __syncthreads();
    // logits of this wave's two parts: the full ascending-j fmaf chain (the oracle's rounding order)
    float sp0 = 0.f, sp1 = 0.f;
    {
        const float *hrow = s_hh + pl * HS, *wrow = s_w2T + (size_t)(sid < PMAX / 2 ? sid : PMAX / 2 - 1) * a.H * 2;
        int j = 0;
        const int H4 = (a.H & 3) == 0 ? a.H : 0;   // rows are 16-byte aligned only when H is a multiple of 4
#pragma unroll 2
        for (; j + 4 <= H4; j += 4) {   // 16-byte LDS reads: four h of this point, two (w0, w1) pairs per read
            const float4 h4 = *(const float4 *)(hrow + j);
            const float4 wa = *(const float4 *)(wrow + 2 * j), wb2 = *(const float4 *)(wrow + 2 * j + 4);
            sp0 = fmaf(wa.x, h4.x, sp0); sp1 = fmaf(wa.y, h4.x, sp1);   // sp1: ignored when !has1 (weights are 0)
            sp0 = fmaf(wa.z, h4.y, sp0); sp1 = fmaf(wa.w, h4.y, sp1);
            sp0 = fmaf(wb2.x, h4.z, sp0); sp1 = fmaf(wb2.y, h4.z, sp1);
            sp0 = fmaf(wb2.z, h4.w, sp0); sp1 = fmaf(wb2.w, h4.w, sp1);
        }
        for (; j < a.H; ++j) {
            const float h = hrow[j];
            sp0 = fmaf(wrow[2 * j], h, sp0);
            sp1 = fmaf(wrow[2 * j + 1], h, sp1);
        }
    }
    PHASE_TS(0, 2);
    const float z0 = (sp0 + g0) / tau, z1 = has1 ? (sp1 + g1) / tau : -INFINITY;
    if (p0 < P) { s_a[p0 * PTS + pl] = sp0; s_e[p0 * PTS + pl] = z0; }
    if (has1) { s_a[(p0 + 1) * PTS + pl] = sp1; s_e[(p0 + 1) * PTS + pl] = z1; }
    __syncthreads();
    // noise-free arg-max (networks/model.py:70, first maximum) and the softmax max
    float m = -INFINITY;
    int am = 0;
    float sm = -INFINITY;
#pragma unroll
    for (int p = 0; p < PMAX; ++p)
        if (PP > 0 || p < P) {
            m = fmaxf(m, s_e[p * PTS + pl]);
            if (sid == 0) {
                const float sv = s_a[p * PTS + pl];
                if (sv > sm) { sm = sv; am = p; }
            }
        }
    __syncthreads();
    const float e0 = expf(z0 - m), e1 = has1 ? expf(z1 - m) : 0.f;
    if (p0 < P) s_e[p0 * PTS + pl] = e0;
    if (has1) s_e[(p0 + 1) * PTS + pl] = e1;
    __syncthreads();
    float sum = 0.f;
#pragma unroll
    for (int p = 0; p < PMAX; ++p)
        if (PP > 0 || p < P) sum += s_e[p * PTS + pl];   // ascending part order
    const float y0 = e0 / sum, y1 = e1 / sum;
    if (p0 < P) { s_a[p0 * PTS + pl] = y0; if (a.yT && live) a.yT[(size_t)p0 * a.N + n] = y0; }
    if (has1) { s_a[(p0 + 1) * PTS + pl] = y1; if (a.yT && live) a.yT[(size_t)(p0 + 1) * a.N + n] = y1; }
    __syncthreads();
    PHASE_TS(0, 3);
    int k = 0;
    float yk = -1.f;
#pragma unroll
    for (int p = 0; p < PMAX; ++p)
        if (PP > 0 || p < P) {
            const float yv = s_a[p * PTS + pl];
            if (yv > yk) { yk = yv; k = p; }
        }
    const float w = (1.0f - yk) + yk;  // y_hard - y_soft.detach() + y_soft
    if (live && sid == 0) {
        if (a.seg_part) a.seg_part[n] = am;
        if (a.hard_idx) a.hard_idx[n] = k;
    }
    PHASE_TS(0, 4);
    for (int t = sid; t < a.B; t += NS) {
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
        if (a.boxes && blockIdx.x * PTS < a.Npad) {
            // AABBs of this workgroup's output points of frame t, one per NN_BOX consecutive points
            // (block-skip test of the K-NN kernels)
            float lo[3], hi[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                static_assert(NN_BOX == 16, "a box is one DPP row");
                lo[c] = reart_row16_min(live ? v[c] : INFINITY);        // four DPP steps each, no LDS crossbar; a row lies inside
                hi[c] = reart_row16_max(live ? v[c] : -INFINITY);       // one half wave, so its lanes are active together
                if (hi[c] == -INFINITY) hi[c] = INFINITY;
            }
            const int pos = blockIdx.x * PTS + pl;
            if ((pl & (NN_BOX - 1)) == 0 && pos < a.Npad) {
                float *o = a.boxes + ((size_t)t * (a.Npad / NN_BOX) + pos / NN_BOX) * 8;
                o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = hi[0]; o[4] = hi[1]; o[5] = hi[2];
            }
        }
    }
    PHASE_TS(0, 5);
}

template <int PP, bool HALF>
static void launch_base_fwd_t(const BaseFwdArgs *ak, int K, hipStream_t st) {
    const BaseFwdArgs &a = ak[0];
    constexpr int PMAX = (PP > 0) ? PP : 32;
    constexpr int W = HALF ? (PMAX + 2 * FW_PG - 1) / (2 * FW_PG) : (PMAX + FW_PG - 1) / FW_PG;
    constexpr int PTS = HALF ? FW_PTS / 2 : FW_PTS;
    const int cover = a.out_soa ? (a.Npad > a.N ? a.Npad : a.N) : a.N;
    const size_t lds = sizeof(float) * (12 * (size_t)a.B * a.P + (size_t)a.H * (4 + PMAX) + 2 * (size_t)PTS * PMAX +
                                        (size_t)(a.H + 4) * PTS);
    if (lds > REART_LDS_DEFAULT_CAP) {  // stateless: raise the dynamic-LDS cap whenever the launch needs it (160 KiB per CU on gfx950)
        (void)hipFuncSetAttribute((const void *)base_fwd_kernel<PP, HALF, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
        (void)hipFuncSetAttribute((const void *)base_fwd_kernel<PP, HALF, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
    }
    if (K == 1) hipLaunchKernelGGL((base_fwd_kernel<PP, HALF, false>), dim3(reart_div_up(cover, PTS)), dim3(64 * W), lds, st, reart_batched(ak, 1));
    else hipLaunchKernelGGL((base_fwd_kernel<PP, HALF, true>), dim3(reart_div_up(cover, PTS), K), dim3(64 * W), lds, st, reart_batched(ak, K));
}
// a.pts = 64 | 32: points per forward workgroup (32, the default: half waves on different part pairs)
template <int PP>
static void launch_base_fwd(const BaseFwdArgs *a, int K, hipStream_t st) {
    if (a[0].pts == 64) launch_base_fwd_t<PP, false>(a, K, st);
    else launch_base_fwd_t<PP, true>(a, K, st);
}

static int dispatch_base_fwd(const BaseFwdArgs *ak, int K, hipStream_t st) {
    if (K < 1 || K > REART_BATCH_MAX) return REART_ERR_INVALID_ARG;
    const BaseFwdArgs &a = ak[0];
    for (int k = 1; k < K; ++k)      // one launch geometry for all
        if (ak[k].N != a.N || ak[k].P != a.P || ak[k].B != a.B || ak[k].H != a.H || ak[k].Npad != a.Npad || ak[k].pts != a.pts ||
            !ak[k].out_soa != !a.out_soa)
            return REART_ERR_INVALID_ARG;
    if (a.P < 1 || a.P > 32) return REART_ERR_UNSUPPORTED;
    if (((size_t)a.B * a.P * 12 + 2 * FW_PTS * 32 + (size_t)a.H * (36 + FW_PTS) + 4 * FW_PTS) * sizeof(float) > 152 * 1024) return REART_ERR_UNSUPPORTED;
    switch (a.P) {
        case 20: launch_base_fwd<20>(ak, K, st); break;
        case 10: launch_base_fwd<10>(ak, K, st); break;
        case 8: launch_base_fwd<8>(ak, K, st); break;
        default: launch_base_fwd<0>(ak, K, st); break;
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
    return dispatch_base_fwd(&a, 1, (hipStream_t)stream);
}

// The production noise stream, exported: out[n][p] = the Gumbel sample the forward kernel draws for (point n, part p) in
// iteration `iter` of an engine seeded with `seed` -- the same philox4x32 call (counter = n, p / 4, iter; key = seed; word
// p % 4) and the same gumbel_from_bits, so injecting `out` into the forward reproduces the in-kernel path bit for bit.
__global__ __launch_bounds__(256) void gumbel_noise_kernel(uint64_t seed, uint64_t it, int N, int P, float *__restrict__ out) {
    const int Q = (P + 3) >> 2;
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (size_t)N * Q) return;
    const int n = (int)(e / Q), q = (int)(e % Q);
    uint32_t r[4];
    philox4x32((uint32_t)n, (uint32_t)q, (uint32_t)it, (uint32_t)(it >> 32), (uint32_t)seed, (uint32_t)(seed >> 32), r);
#pragma unroll
    for (int w = 0; w < 4; ++w)
        if (4 * q + w < P) out[(size_t)n * P + 4 * q + w] = gumbel_from_bits(r[w]);
}

extern "C" int reart_gumbel_noise(uint64_t seed, int64_t iter, int N, int P, float *out, void *stream) {
    if (N < 0 || P < 1 || iter < 0) return REART_ERR_INVALID_ARG;
    if (N == 0) return REART_OK;
    if (!out) return REART_ERR_INVALID_ARG;
    const size_t total = (size_t)N * ((P + 3) >> 2);
    gumbel_noise_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(seed, (uint64_t)iter, N, P, out);
    REART_CHECK_LAUNCH();
    return REART_OK;
}

// entry used by the fused step (step.hip)
int reart_base_forward_ex(const BaseFwdArgs &a, hipStream_t st) { return dispatch_base_fwd(&a, 1, st); }
int reart_base_forward_batch(const BaseFwdArgs *a, int K, hipStream_t st) { return dispatch_base_fwd(a, K, st); }

// ------------------------------------------------------------------------------- backward
// layout of one partial row / of the reduced gradient vector
__host__ __device__ static inline int off_gW2() { return 0; }
__host__ __device__ static inline int off_gW1(int P, int H) { return P * H; }
__host__ __device__ static inline int off_gb1(int P, int H) { return P * H + 3 * H; }
__host__ __device__ static inline int off_gRt(int P, int H) { return P * H + 4 * H; }
__host__ __device__ static inline int n_out(int P, int H, int B) { return P * H + 4 * H + 12 * B * P; }

// [R|t] table [B*P][12] in global memory, so that the backward can read it with scalar loads
__global__ __launch_bounds__(256) void rt_table_kernel(const float *__restrict__ p6d,
                                                       const float *__restrict__ pt, int n,
                                                       float *__restrict__ table) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    float R[9];
    r6d_to_matrix(p6d + 6 * (size_t)e, R);
#pragma unroll
    for (int c = 0; c < 9; ++c) table[12 * (size_t)e + c] = R[c];
#pragma unroll
    for (int c = 0; c < 3; ++c) table[12 * (size_t)e + 9 + c] = pt[3 * (size_t)e + c];
}

// (1) One workgroup per chunk of 64 points, W = ceil(P/2) waves x the same points:
//   a. wave g: dL/dw[n,p] = sum_t G[t,n].(R[t,p] x_n + t[t,p]) for parts {2g, 2g+1}; the [R|t]
//      table is wave-uniform and read with scalar loads
//   b. softmax backward -> ds[n,:] (LDS); wave g: hidden gradient dp[n,j] for its slice of j
//   c. partial sums over the chunk of every parameter gradient, each output accumulated
//      sequentially over the chunk's points (ascending n) -> deterministic:
//        gW2[p,j] = sum_n ds[n,p] h[n,j];  gW1[j,c] = sum_n dp[n,j] x[n,c];  gb1[j] = sum_n dp[n,j]
//        gR[t,p]  = sum_{n:k_n=p} w_n G[t,n] x_n^T;  gt[t,p] = sum_{n:k_n=p} w_n G[t,n]
//      For gR/gt the chunk's points are first ordered by part with a stable in-wave counting
//      sort (ballot + popcount), so thread (t, entry) walks each part's points in ascending n
//      with a register accumulator -- no LDS read-modify-write chain, no atomics.
#define BW_LD (RED_CHUNK + 1)

template <int PP, bool BATCH>
__global__ __launch_bounds__(64 * (((PP > 0 ? PP : 32) + FW_PG - 1) / FW_PG)) void base_bwd_block_kernel(Batched<BaseBwdArgs> ab) {
    const BaseBwdArgs &a = ab.a[BATCH ? blockIdx.y : 0];
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int PMAX = (PP > 0) ? PP : 32;
    constexpr int W = (PMAX + FW_PG - 1) / FW_PG;
    constexpr int BS = 64 * W;
    const int P = (PP > 0) ? PP : a.P;
    float *s_h = smem;                               // [H][BW_LD]  hT tile, later dp tile
    float *s_ds = s_h + (size_t)a.H * BW_LD;         // [PMAX][BW_LD]  dw, later ds
    float *s_x = s_ds + (size_t)PMAX * BW_LD;        // [RED_CHUNK][3]
    float *s_w = s_x + RED_CHUNK * 3;                // [RED_CHUNK]
    int *s_kn = (int *)(s_w + RED_CHUNK);            // [RED_CHUNK] hard part of each point (-1: padding)
    float *s_G = (float *)(s_kn + RED_CHUNK + PMAX + 4);  // [B][RED_CHUNK*3]  upstream gradient tile
    float *s_w2T = s_G + (size_t)a.B * RED_CHUNK * 3;// [H][PMAX]
    float *s_rt = s_w2T + (size_t)a.H * PMAX;        // [B*P][12]  [R|t] rows
    const int tid = threadIdx.x, lane = tid & 63, grp = tid >> 6, chunk = blockIdx.x;
    // a workgroup owns a.cpts (64 or 32) points; the tiles keep their 64-point layout (columns >= cn are zero), the
    // matrix-core loops stop at cn: with 32 points twice as many workgroups each run half the reductions
    const int n0 = chunk * a.cpts;
    const int cn = (a.N - n0) < a.cpts ? (a.N - n0) : a.cpts;
    const int cn16 = (cn + 15) & ~15;   // tile columns beyond cn are zero: whole blocks of 16 points keep the loops unrolled
    float *prow = a.partial + (size_t)chunk * n_out(a.P, a.H, a.B);
    PHASE_TS(1, 0);

    // Tile loads.  Every group issues a batch of unconditional loads (clamped addresses, masked
    // afterwards), and the first batch of EVERY group is in flight before anything is stored to LDS:
    // written as plain guarded loops the compiler serialises the loads and the prologue becomes a
    // chain of a dozen L2 / HBM round trips (measured 25 k cycles, now one round trip deep).
    constexpr int UH = 13, UG = 6, UR = 8;   // H = 128, B = 19, P = 20: one batch each
    const int ilast = cn - 1, nH = a.H * RED_CHUNK, nG = a.B * RED_CHUNK * 3, rlast = 3 * cn - 1, nR = a.B * a.P * 12;
    const bool live = lane < cn;
    const int n = live ? n0 + lane : n0;
    auto load_h = [&](int e0, float (&v)[UH]) {
#pragma unroll
        for (int u = 0; u < UH; ++u) {
            const int e = e0 + u * BS;
            const int ec = e < nH ? e : tid;
            const int j = ec >> 6, i = ec & 63;            // RED_CHUNK == 64
            v[u] = a.hT[(size_t)j * a.N + n0 + (i < cn ? i : ilast)];
        }
    };
    auto store_h = [&](int e0, const float (&v)[UH]) {
#pragma unroll
        for (int u = 0; u < UH; ++u) {
            const int e = e0 + u * BS;
            if (e < nH) s_h[(e >> 6) * BW_LD + (e & 63)] = ((e & 63) < cn) ? v[u] : 0.f;
        }
    };
    // upstream gradient tile; the fused step adds the flow-loss terms of the two adjacent pairs
    // (complete frame fc = t or t + 1: + d/d pred_flow of pair fc - 1, - d/d pred_flow of pair fc)
    auto load_g = [&](int e0, float (&g)[UG], float (&gh)[UG], float (&gl)[UG]) {
#pragma unroll
        for (int u = 0; u < UG; ++u) {
            const int e = e0 + u * BS;
            const int ec = e < nG ? e : tid;
            const int t = ec / (RED_CHUNK * 3), r0_ = ec - t * (RED_CHUNK * 3);
            const int r = r0_ < 3 * cn ? r0_ : rlast;
            g[u] = a.G[3 * ((size_t)t * a.N + n0) + r];
            gh[u] = 0.f; gl[u] = 0.f;
            if (a.gpf) {   // uniform
                const int fc = t < a.cano_idx ? t : t + 1;   // complete-sequence index of frame t
                const int fh = fc - 1 >= 0 ? fc - 1 : 0, fl = fc <= a.B - 1 ? fc : a.B - 1;
                gh[u] = a.gpf[3 * ((size_t)fh * a.N + n0) + r];
                gl[u] = a.gpf[3 * ((size_t)fl * a.N + n0) + r];
            }
        }
    };
    auto store_g = [&](int e0, const float (&g)[UG], const float (&gh)[UG], const float (&gl)[UG]) {
#pragma unroll
        for (int u = 0; u < UG; ++u) {
            const int e = e0 + u * BS;
            if (e < nG) {
                const int t = e / (RED_CHUNK * 3), r0_ = e - t * (RED_CHUNK * 3);
                float v = g[u];
                if (a.gpf) {
                    const int fc = t < a.cano_idx ? t : t + 1;
                    if (fc - 1 >= 0) v += gh[u];       // operation order: (G + gh) - gl
                    if (fc <= a.B - 1) v -= gl[u];
                }
                s_G[e] = r0_ < 3 * cn ? v : 0.f;
            }
        }
    };
    auto load_r = [&](int e0, float (&v)[UR]) {
#pragma unroll
        for (int u = 0; u < UR; ++u) v[u] = a.rt_table[e0 + u * BS < nR ? e0 + u * BS : tid];
    };
    auto store_r = [&](int e0, const float (&v)[UR]) {
#pragma unroll
        for (int u = 0; u < UR; ++u)
            if (e0 + u * BS < nR) s_rt[e0 + u * BS] = v[u];
    };
    float vh[UH], vg[UG], vgh[UG], vgl[UG], vr[UR], vw[PMAX];
    load_g(tid, vg, vgh, vgl);     // produced by the previous kernels on other XCDs: the longest latency first
    load_h(tid, vh);
    load_r(tid, vr);
    const int jw = tid < a.H ? tid : 0;
#pragma unroll
    for (int p = 0; p < PMAX; ++p) vw[p] = a.W2[(size_t)(p < P ? p : 0) * a.H + jw];
    const float x0 = a.cano[3 * (size_t)n], x1 = a.cano[3 * (size_t)n + 1], x2 = a.cano[3 * (size_t)n + 2];
    int kn = -1;
    float yk = 0.f;
    if (grp == 0) {
        kn = live ? a.hard_idx[n] : -1;
        yk = a.yT[(size_t)(kn < 0 ? 0 : kn) * a.N + n];
    }
    store_g(tid, vg, vgh, vgl);
    store_h(tid, vh);
    store_r(tid, vr);
    if (tid < a.H) {
#pragma unroll
        for (int p = 0; p < PMAX; ++p) s_w2T[tid * PMAX + p] = vw[p];
    }
    for (int e0 = tid + UG * BS; e0 < nG; e0 += UG * BS) { load_g(e0, vg, vgh, vgl); store_g(e0, vg, vgh, vgl); }
    for (int e0 = tid + UH * BS; e0 < nH; e0 += UH * BS) { load_h(e0, vh); store_h(e0, vh); }
    for (int e0 = tid + UR * BS; e0 < nR; e0 += UR * BS) { load_r(e0, vr); store_r(e0, vr); }
    for (int j = tid + BS; j < a.H; j += BS)
        for (int p = 0; p < P; ++p) s_w2T[j * PMAX + p] = a.W2[(size_t)p * a.H + j];
    if (grp == 0) {
        s_w[lane] = (1.0f - yk) + yk;
        s_x[3 * lane] = live ? x0 : 0.f; s_x[3 * lane + 1] = live ? x1 : 0.f; s_x[3 * lane + 2] = live ? x2 : 0.f;
        s_kn[lane] = kn;
    }
    // a. dw[n,p] = sum_t G[t,n] . (R[t,p] x_n + t[t,p]) as ONE matrix product on the matrix cores:
    //      dw = A Bm,   A[n][(t,r,c)] = G[t,n,r] * (c < 3 ? x_n[c] : 1),   Bm[(t,r,c)][p] = [R|t][t,p][r][c]
    //    (K = 12 B, ascending (t, r, c): a fixed summation order, deterministic).  v_mfma_f32_16x16x4_f32:
    //    one 16 (points) x 16 (parts) tile per wave, the four k of an instruction are c = 0..3 of one (t, r).
    //    The per-lane VALU form of this sum (apply, then dot) was LDS-bound on the broadcast [R|t] rows.
    const int p0 = grp * FW_PG;
    const bool has0 = p0 < P, has1 = p0 + 1 < P;
    __syncthreads();
    PHASE_TS(1, 1);
    {
        typedef float f4v __attribute__((ext_vector_type(4)));
        const int ntn = (P + 15) / 16;                       // part tiles
        const int mtn = (cn + 15) >> 4;                      // point tiles
        for (int tile = grp; tile < mtn * ntn; tile += W) {
            const int nt = tile / mtn, mt = tile - nt * mtn;
            const int nl = 16 * mt + (lane & 15), kq = lane >> 4;          // A row (point), k within the instruction
            const int pc = 16 * nt + (lane & 15);                          // B column (part)
            const bool pok = pc < P;
            const float xt = kq < 3 ? s_x[3 * nl + kq] : 1.0f;
            const float *gp = s_G + 3 * nl;                                // + t * 192 + r
            const float *bp = s_rt + 12 * (pok ? pc : 0);                  // + t * 12 P + (kq < 3 ? 3 r + kq : 9 + r)
            f4v c = {0.f, 0.f, 0.f, 0.f};
            const int bo = kq < 3 ? kq : 9;                     // offset of (r = 0, c = kq) inside a [R|t] row
            const int bs = kq < 3 ? 3 : 1;                      // stride over r
            constexpr int TU = 4;                               // frames per batch: 24 LDS reads in flight, then 12 MFMAs
            for (int t0 = 0; t0 < a.B; t0 += TU) {
                float av[TU][3], bv[TU][3];
#pragma unroll
                for (int u = 0; u < TU; ++u) {
                    const int t = t0 + u < a.B ? t0 + u : a.B - 1;
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        av[u][r] = gp[t * (RED_CHUNK * 3) + r];
                        bv[u][r] = bp[t * 12 * a.P + bo + bs * r];
                    }
                }
#pragma unroll
                for (int u = 0; u < TU; ++u) {
                    const bool tok = t0 + u < a.B;
#pragma unroll
                    for (int r = 0; r < 3; ++r)
                        c = __builtin_amdgcn_mfma_f32_16x16x4f32(tok ? av[u][r] * xt : 0.f, (pok && tok) ? bv[u][r] : 0.f, c, 0, 0, 0);
                }
            }
            // C/D layout: row = 4 (lane >> 4) + reg, col = lane & 15
#pragma unroll
            for (int reg = 0; reg < 4; ++reg)
                if (pok) s_ds[pc * BW_LD + 16 * mt + 4 * kq + reg] = c[reg];
        }
    }
    __syncthreads();
    PHASE_TS(1, 2);
    const float dw0 = has0 ? s_ds[p0 * BW_LD + lane] : 0.f;
    const float dw1 = has1 ? s_ds[(p0 + 1) * BW_LD + lane] : 0.f;
    // b. softmax backward: dot over all parts in ascending order, ds for this wave's parts
    const float tau = a.tau_ptr ? a.tau_ptr[0] : a.tau;
    float dot = 0.f;
#pragma unroll
    for (int p = 0; p < PMAX; ++p)
        if (PP > 0 || p < P) dot = fmaf(a.yT[(size_t)p * a.N + n], s_ds[p * BW_LD + lane], dot);
    float ds0 = 0.f, ds1 = 0.f;
    if (has0) ds0 = (a.yT[(size_t)p0 * a.N + n] * (dw0 - dot)) / tau;
    if (has1) ds1 = (a.yT[(size_t)(p0 + 1) * a.N + n] * (dw1 - dot)) / tau;
    __syncthreads();  // every wave has read dw before it is overwritten by ds
    if (has0) s_ds[p0 * BW_LD + lane] = live ? ds0 : 0.f;
    if (has1) s_ds[(p0 + 1) * BW_LD + lane] = live ? ds1 : 0.f;
    __syncthreads();
    PHASE_TS(1, 3);
    // c1. gW2[p,j] = sum_i ds[p,i] h[j,i] on the matrix cores: v_mfma_f32_32x32x2_f32 is bit for bit the
    // ascending-i fmaf chain (cdna guide section 3).  One 32 (parts) x 32 (hidden) tile per wave.
    for (int tile = grp; tile * 32 < a.H; tile += W) {
        typedef float f16v __attribute__((ext_vector_type(16)));
        f16v c = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const int pr = lane & 31, jc = tile * 32 + (lane & 31), kh = lane >> 5;
        const float *ap = s_ds + (pr < P ? pr : 0) * BW_LD + kh;
        const float *bp = s_h + (jc < a.H ? jc : 0) * BW_LD + kh;
        const bool aok = pr < P, bok = jc < a.H;
        for (int kb = 0; kb < cn16; kb += 16) {
#pragma unroll
            for (int kk = kb; kk < kb + 16; kk += 2) {
                const float av = aok ? ap[kk] : 0.f;
                const float bv = bok ? bp[kk] : 0.f;
                c = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, c, 0, 0, 0);
            }
        }
        // C/D layout: row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5), col = lane & 31
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int prow_p = (reg & 3) + 8 * (reg >> 2) + 4 * kh;
            if (prow_p < P && bok) prow[off_gW2() + prow_p * a.H + jc] = c[reg];
        }
    }
    PHASE_TS(1, 4);
    // c2. gR | gt on the matrix cores:  out[p][(t, e)] = sum_n onehot[p][n] * v[n][(t, e)]  with
    //   v = (w_n G[t,n,r]) x_n[c]  (e = 3 r + c < 9)   or   w_n G[t,n,e-9]  (translation),
    // the operands built on the fly from the LDS tiles.  A one-hot left factor makes every product
    // exact (1 * v = v, 0 * v = 0), so each output is the ascending-n running sum of its part's
    // points: deterministic, no sorting, no atomics.
    for (int tile = grp; tile * 32 < a.B * 12; tile += W) {
        typedef float f16v __attribute__((ext_vector_type(16)));
        f16v c = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const int pr = lane & 31, col = tile * 32 + (lane & 31), kh = lane >> 5;
        const bool cok = col < a.B * 12;
        const int t = cok ? col / 12 : 0, e = cok ? col - t * 12 : 0;
        const int gr = e < 9 ? e / 3 : e - 9, xc = e < 9 ? e - 3 * (e / 3) : 0;
        const float *gcol = s_G + t * (RED_CHUNK * 3) + gr;
        for (int kb = 0; kb < cn16; kb += 16) {
#pragma unroll
            for (int kk = kb; kk < kb + 16; kk += 2) {
                const int n = kk + kh;
                const float av = (s_kn[n] == pr) ? 1.f : 0.f;
                float bv = s_w[n] * gcol[3 * n];
                if (e < 9) bv = bv * s_x[3 * n + xc];
                c = __builtin_amdgcn_mfma_f32_32x32x2f32(av, cok ? bv : 0.f, c, 0, 0, 0);
            }
        }
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int prow_p = (reg & 3) + 8 * (reg >> 2) + 4 * kh;
            if (prow_p < P && cok) prow[off_gRt(a.P, a.H) + (t * a.P + prow_p) * 12 + e] = c[reg];
        }
    }
    PHASE_TS(1, 5);
    // every lane needs all ds of its point for the hidden gradient
    float dsr[PMAX];
#pragma unroll
    for (int p = 0; p < PMAX; ++p) dsr[p] = (PP > 0 || p < P) ? s_ds[p * BW_LD + lane] : 0.f;
    __syncthreads();  // the h tile and ds have been consumed by c1
    PHASE_TS(1, 6);
    // b'. dp[n,j] = relu'(h) * sum_p W2[p,j] ds[p], written over the h tile
    const int jq = (a.H + W - 1) / W, j0 = grp * jq, j1 = (j0 + jq < a.H) ? j0 + jq : a.H;
    for (int j = j0; j < j1; ++j) {
        float dh = 0.f;
        if (PMAX % 4 == 0) {
#pragma unroll
            for (int p4 = 0; p4 < PMAX / 4; ++p4) {
                const float4 wv = *(const float4 *)(s_w2T + j * PMAX + 4 * p4);
                const float w4[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (PP > 0 || 4 * p4 + u < P) dh = fmaf(w4[u], dsr[4 * p4 + u], dh);
            }
        } else {
#pragma unroll
            for (int p = 0; p < PMAX; ++p)
                if (PP > 0 || p < P) dh = fmaf(s_w2T[j * PMAX + p], dsr[p], dh);
        }
        const float h = s_h[j * BW_LD + lane];
        s_h[j * BW_LD + lane] = (live && h > 0.f) ? dh : 0.f;
    }
    __syncthreads();
    PHASE_TS(1, 7);
    // c3. gW1 / gb1 partial from (dp, x)
    for (int o = tid; o < 4 * a.H; o += BS) {
        const int j = o >> 2, c = o & 3;
        float acc = 0.f;
        if (c < 3) {
            for (int ib = 0; ib < cn16; ib += 16) {
#pragma unroll
                for (int i = ib; i < ib + 16; ++i) acc = fmaf(s_h[j * BW_LD + i], s_x[3 * i + c], acc);
            }
            prow[off_gW1(a.P, a.H) + 3 * j + c] = acc;
        } else {
            for (int ib = 0; ib < cn16; ib += 16) {
#pragma unroll
                for (int i = ib; i < ib + 16; ++i) acc += s_h[j * BW_LD + i];
            }
            prow[off_gb1(a.P, a.H) + j] = acc;
        }
    }
    PHASE_TS(1, 8);
}

// (2) sum the chunk partials in ascending chunk order; Gram-Schmidt backward for the
// 6-vectors; optionally the Adam update of the very parameter this thread reduced.

__device__ __forceinline__ void adam_update(float *p, float g, float *m, float *v, float lr,
                                            float step_size_base, float bc2s, float beta1, float beta2,
                                            float eps, float weight_decay = 0.f) {
    (void)lr;
    if (weight_decay != 0.f) g = g + weight_decay * (*p);   // torch.optim.Adam: grad.add(param, alpha=weight_decay)
    float mm = *m, vv = *v;
    mm = mm + (g - mm) * (1.0f - beta1);
    vv = vv * beta2 + ((1.0f - beta2) * g) * g;
    const float denom = sqrtf(vv) / bc2s + eps;
    *m = mm; *v = vv;
    *p = *p - step_size_base * (mm / denom);
}

__device__ __forceinline__ void base_bwd_finalize_body(const BaseBwdArgs &a, const FinalizeAdam &ad) {
    // Every output is summed over the partial rows by FOUR lanes, a quarter of the rows each (one batch of loads in
    // flight per lane at 128 rows instead of four dependent batches), combined in a fixed order by two xor-shuffles.
    const int og = blockIdx.x * 256 + threadIdx.x;
    const int nWr = a.P * a.H + 4 * a.H;                 // real weight entries
    const int nW = (4 * nWr + 63) & ~63;                 // four lanes per weight; pose groups start wave-aligned
    const int RQ = (a.nchunk + 3) >> 2;                  // rows per quarter
    const int o = og >> 2;                               // weight entry of this lane (first branch)
    const int no = n_out(a.P, a.H, a.B);
    float ss_seg = 0.f, ss_tr = 0.f, bc2s = 1.f;
    if (ad.enabled) {
        // bias corrections of THIS step, written by the previous bookkeeping kernel / prepare
        const double bc1 = ad.bias_corr[0];
        ss_seg = (float)((double)ad.seg_lr / bc1);
        ss_tr = (float)((double)ad.trans_lr / bc1);
        bc2s = (float)ad.bias_corr[1];
    }
    if (og < nW) {
        if (o >= nWr) return;                            // whole quads leave together
        float acc = 0.f;
        int c = (og & 3) * RQ;
        const int cend = c + RQ < a.nchunk ? c + RQ : a.nchunk;
        for (; c + 32 <= cend; c += 32) {  // 32 independent loads in flight, fixed add order
            float v[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) v[u] = a.partial[(size_t)(c + u) * no + o];
#pragma unroll
            for (int u = 0; u < 32; ++u) acc += v[u];
        }
        for (; c + 8 <= cend; c += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = a.partial[(size_t)(c + u) * no + o];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
        }
        for (; c < cend; ++c) acc += a.partial[(size_t)c * no + o];
        acc = acc + __shfl_xor(acc, 1, 64);              // (q0 + q1), (q2 + q3): commutative, every lane of the quad agrees
        acc = acc + __shfl_xor(acc, 2, 64);
        if (og & 3) return;
        // partial-row order is W2 | W1 | b1; moment order is W1 | b1 | W2
        const int nW2 = a.P * a.H, nW1 = 3 * a.H;
        if (o < nW2) {
            a.gW2[o] = acc;
            if (ad.enabled)
                adam_update(ad.W2 + o, acc, ad.m + nW1 + a.H + o, ad.v + nW1 + a.H + o, ad.seg_lr, ss_seg, bc2s,
                            ad.beta1, ad.beta2, ad.eps, ad.weight_decay);
        } else if (o < nW2 + nW1) {
            const int q = o - nW2;
            a.gW1[q] = acc;
            if (ad.enabled)
                adam_update(ad.W1 + q, acc, ad.m + q, ad.v + q, ad.seg_lr, ss_seg, bc2s, ad.beta1, ad.beta2, ad.eps, ad.weight_decay);
        } else {
            const int q = o - nW2 - nW1;
            a.gb1[q] = acc;
            if (ad.enabled)
                adam_update(ad.b1 + q, acc, ad.m + nW1 + q, ad.v + nW1 + q, ad.seg_lr, ss_seg, bc2s, ad.beta1,
                            ad.beta2, ad.eps, ad.weight_decay);
        }
    } else if (og < nW + 64 * a.B * a.P) {
        // one wave per (frame, part): 4 row quarters x 16 lanes; lane c < 12 of a quarter sums one entry of dL/d[R|t]
        // over its rows (all its loads in flight at once), the quarters meet through two xor-shuffles, then every
        // 16-lane group holds the 12 sums and runs the Gram-Schmidt backward; the first group updates the parameters
        const int q = og - nW, e = q >> 6, tq = (q >> 4) & 3, c = q & 15;
        float acc = 0.f;
        if (c < 12) {
            const float *pr = a.partial + off_gRt(a.P, a.H) + 12 * (size_t)e + c;
            int ch = tq * RQ;
            const int chend = ch + RQ < a.nchunk ? ch + RQ : a.nchunk;
            for (; ch + 32 <= chend; ch += 32) {
                float v[32];
#pragma unroll
                for (int u = 0; u < 32; ++u) v[u] = pr[(size_t)(ch + u) * no];
#pragma unroll
                for (int u = 0; u < 32; ++u) acc += v[u];
            }
            for (; ch < chend; ++ch) acc += pr[(size_t)ch * no];
        }
        acc = acc + __shfl_xor(acc, 16, 64);
        acc = acc + __shfl_xor(acc, 32, 64);
        float gRt[12];
        const int lane0 = (threadIdx.x & 63) & ~15;
#pragma unroll
        for (int k = 0; k < 12; ++k) gRt[k] = __shfl(acc, lane0 + k, 64);
        // every lane of the group runs the (cheap) Gram-Schmidt backward; lane c < 6 then owns rotation
        // entry c and lanes 6..8 the translation entries: nine independent Adam updates instead of a
        // chain of nine on one lane
        if (c >= 9 || tq != 0) return;
        float g6[6];
        r6d_backward(a.p6d + 6 * (size_t)e, gRt, g6);
        const int base6 = 3 * a.H + a.H + a.P * a.H, baset = base6 + 6 * a.B * a.P;
        if (c < 6) {
            float g = g6[0];
#pragma unroll
            for (int k = 1; k < 6; ++k) g = (c == k) ? g6[k] : g;
            a.g6d[6 * (size_t)e + c] = g;
            if (ad.enabled)
                adam_update(ad.p6d + 6 * (size_t)e + c, g, ad.m + base6 + 6 * e + c, ad.v + base6 + 6 * e + c,
                            ad.trans_lr, ss_tr, bc2s, ad.beta1, ad.beta2, ad.eps, ad.weight_decay);
        } else {
            const int k = c - 6;
            const float g = k == 0 ? gRt[9] : (k == 1 ? gRt[10] : gRt[11]);
            a.gt[3 * (size_t)e + k] = g;
            if (ad.enabled)
                adam_update(ad.pt + 3 * (size_t)e + k, g, ad.m + baset + 3 * e + k, ad.v + baset + 3 * e + k,
                            ad.trans_lr, ss_tr, bc2s, ad.beta1, ad.beta2, ad.eps, ad.weight_decay);
        }
    }
}

// 256 threads of column sums / pose gradients / Adam (base_bwd_finalize_body) + TWO MORE WAVES for the bookkeeping.
template <bool BATCH>
__global__ __launch_bounds__(384) void base_bwd_finalize_kernel(Batched<BaseBwdArgs> ab, Batched<FinalizeAdam> adb, Batched<StepBook> bkb) {
    const BaseBwdArgs &a = ab.a[BATCH ? blockIdx.y : 0];
    const FinalizeAdam &ad = adb.a[BATCH ? blockIdx.y : 0];
    const StepBook &bk = bkb.a[BATCH ? blockIdx.y : 0];
    // Bookkeeping is done by the LAST workgroup to finish (ticket): by then every other workgroup has
    // consumed this iteration's counters.  Every workgroup prepares it speculatively -- the loss partials
    // are summed (fifth wave) and the next temperature / bias corrections computed (sixth wave: two double-
    // precision pow, a sqrt, a cosine: 2.2 us on one lane) -- beside the four waves that run the main work
    // (1.8 us), not in front of them (profiles/r05_small_kernel_clocks.txt); the winner only has to store.
    __shared__ int s_last;
    __shared__ double s_book[4];       // recon, flow (x lambda), bias corrections of the next step
    __shared__ float s_tau;
    __shared__ long s_it;
    const int tid = threadIdx.x;
#ifdef REART_PHASE_CLOCK
    if (blockIdx.x == 1 && blockIdx.y == 0 && threadIdx.x == 0) { g_phase_ts[0][14] = __builtin_amdgcn_s_memtime(); g_phase_ts[0][8] = g_phase_ts[0][14]; }
#endif
    if (tid >= 320) {                  // sixth wave: the scalars of the next step (no memory but the iteration counter)
        if (!bk.enabled) return;
        if (tid == 320) {
            const long it = (long)bk.iter[0];
            s_it = it;
            s_book[2] = 1.0 - pow((double)bk.beta1, (double)(it + 2));   // Adam step count of the next iteration
            s_book[3] = sqrt(1.0 - pow((double)bk.beta2, (double)(it + 2)));
            // iteration i (0-based) uses tau_cosine(i+1, ...) (run_robot.py:157)
            s_tau = bk.fixed_tau > 0.f ? bk.fixed_tau : reart_tau_schedule(it + 2, bk.n_iter, bk.end_tau, bk.start_tau);
        }
    } else if (tid >= 256) {           // fifth wave: the loss partials
        if (!bk.enabled) return;
        const int l = tid - 256;
        // fixed assignment of terms to lanes + fixed-order tree: deterministic sums
        double recon = 0.0, flow = 0.0;
        for (int b = l; b < bk.n_frame_part; b += 64) recon += bk.frame_loss[b];
        for (int i = l; i < bk.n_flow_part; i += 64) flow += bk.flow_part[i];
        recon = reart_wave_sum_d(recon);
        flow = reart_wave_sum_d(flow);
        if (l == 0) { s_book[0] = recon; s_book[1] = flow * (double)bk.lambda_flow; }
    } else base_bwd_finalize_body(a, ad);
#ifdef REART_PHASE_CLOCK
    if (blockIdx.x == 1 && blockIdx.y == 0 && threadIdx.x == 0) g_phase_ts[0][9] = __builtin_amdgcn_s_memtime();
#endif
    if (!bk.enabled) return;
    __syncthreads();   // every thread of this workgroup has read the counters it needs (values already used); s_book is written
    if (tid == 0) s_last = (atomicAdd(bk.ticket, 1u) == gridDim.x - 1) ? 1 : 0;
    __syncthreads();
#ifdef REART_PHASE_CLOCK
    if (blockIdx.x == 1 && blockIdx.y == 0 && threadIdx.x == 0) g_phase_ts[0][10] = __builtin_amdgcn_s_memtime();   // after the ticket's round trip
#endif
    if (!s_last || tid != 0) return;
    const long it = s_it;
    if (bk.losses && bk.ring > 0) {
        float *row = bk.losses + 4 * (size_t)(it % bk.ring);
        row[0] = (float)s_book[0]; row[1] = (float)s_book[1]; row[2] = (float)(s_book[0] + s_book[1]); row[3] = bk.tau[0];
    }
    bk.iter[0] = it + 1;
    bk.bias_corr[0] = s_book[2];
    bk.bias_corr[1] = s_book[3];
    bk.tau[0] = s_tau;
    *bk.ticket = 0u;
}

static size_t base_bwd_ws_layout(int N, int P, int B, int H, size_t *o_rt, size_t *o_part) {
    const int nchunk = reart_div_up(N, 16);   // room for the 16-point form (four times the partial rows of the 64-point form)
    size_t off = 0;
    *o_rt = off; off += reart_align_up(sizeof(float) * 12 * (size_t)B * P, 256);
    *o_part = off; off += reart_align_up(sizeof(float) * (size_t)nchunk * n_out(P, H, B), 256);
    return off;
}

extern "C" size_t reart_base_backward_workspace_bytes(int N, int P, int B, int H) {
    if (N <= 0 || P <= 0 || B <= 0 || H <= 0) return 0;
    size_t a, b;
    return base_bwd_ws_layout(N, P, B, H, &a, &b);
}

template <int PP>
static int launch_bwd_block(const BaseBwdArgs *ak, int K, size_t lds, hipStream_t st) {
    if (lds > REART_LDS_DEFAULT_CAP &&
        (hipFuncSetAttribute((const void *)base_bwd_block_kernel<PP, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024) != hipSuccess ||
         hipFuncSetAttribute((const void *)base_bwd_block_kernel<PP, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024) != hipSuccess))
        return REART_ERR_LAUNCH;
    constexpr int W = (((PP > 0) ? PP : 32) + FW_PG - 1) / FW_PG;
    if (K == 1) hipLaunchKernelGGL((base_bwd_block_kernel<PP, false>), dim3(ak[0].nchunk), dim3(64 * W), lds, st, reart_batched(ak, 1));
    else hipLaunchKernelGGL((base_bwd_block_kernel<PP, true>), dim3(ak[0].nchunk, K), dim3(64 * W), lds, st, reart_batched(ak, K));
    return REART_OK;
}

// K instances of one shape (K = 1: the plain entry); workspaces[k] is instance k's backward workspace.
// a.rt_table == NULL: the table is built into the workspace first (one extra tiny launch per instance)
int reart_base_backward_batch(const BaseBwdArgs *args, const FinalizeAdam *adam, const StepBook *book, void *const *workspaces,
                              size_t workspace_bytes, int K, hipStream_t st) {
    if (K < 1 || K > REART_BATCH_MAX) return REART_ERR_INVALID_ARG;
    BaseBwdArgs ak[REART_BATCH_MAX];
    for (int k = 0; k < K; ++k) ak[k] = args[k];
    const BaseBwdArgs &a0 = ak[0];
    if (a0.P > 32) return REART_ERR_UNSUPPORTED;
    size_t o_rt, o_part;
    const size_t need = base_bwd_ws_layout(a0.N, a0.P, a0.B, a0.H, &o_rt, &o_part);
    if (!workspaces || workspace_bytes < need) return REART_ERR_INVALID_ARG;
    for (int k = 0; k < K; ++k) {
        BaseBwdArgs &a = ak[k];
        a.cpts = (a.cpts == 64 || a.cpts == 32 || a.cpts == 16) ? a.cpts : 32;   // points per backward workgroup
        a.nchunk = reart_div_up(a.N, a.cpts);
        if (a.N != a0.N || a.P != a0.P || a.B != a0.B || a.H != a0.H || a.cpts != a0.cpts || !workspaces[k]) return REART_ERR_INVALID_ARG;
        char *ws = (char *)workspaces[k];
        a.partial = (float *)(ws + o_part);
        if (!a.rt_table) {
            float *table = (float *)(ws + o_rt);
            hipLaunchKernelGGL(rt_table_kernel, dim3(reart_div_up(a.B * a.P, 256)), dim3(256), 0, st, a.p6d, a.pt,
                               a.B * a.P, table);
            a.rt_table = table;
        }
    }
    const int PMAX = (a0.P == 20 || a0.P == 10 || a0.P == 8) ? a0.P : 32;
    const size_t lds = sizeof(float) * ((size_t)(a0.H + PMAX) * BW_LD + RED_CHUNK * 5 + PMAX + 4 +
                                        (size_t)a0.B * RED_CHUNK * 3 + (size_t)a0.H * PMAX + 12 * (size_t)a0.B * a0.P);
    if (lds > 152 * 1024) return REART_ERR_UNSUPPORTED;
    int rc;
    switch (a0.P) {
        case 20: rc = launch_bwd_block<20>(ak, K, lds, st); break;
        case 10: rc = launch_bwd_block<10>(ak, K, lds, st); break;
        case 8: rc = launch_bwd_block<8>(ak, K, lds, st); break;
        default: rc = launch_bwd_block<0>(ak, K, lds, st); break;
    }
    if (rc != REART_OK) return rc;
    Batched<FinalizeAdam> adb = {};
    Batched<StepBook> bkb = {};
    for (int k = 0; k < K; ++k) {
        if (adam) adb.a[k] = adam[k];
        if (book) bkb.a[k] = book[k];
    }
    // weights: one thread per entry; poses: 16 lanes per (frame, part); nW is rounded up to a multiple of 64
    // inside the kernel's indexing so that a 16-lane group never straddles a wave
    const int nfin = (int)reart_align_up((size_t)4 * (a0.P * a0.H + 4 * a0.H), 64) + 64 * a0.B * a0.P;
    if (K == 1) hipLaunchKernelGGL(base_bwd_finalize_kernel<false>, dim3(reart_div_up(nfin, 256)), dim3(384), 0, st, reart_batched(ak, 1), adb, bkb);
    else hipLaunchKernelGGL(base_bwd_finalize_kernel<true>, dim3(reart_div_up(nfin, 256), K), dim3(384), 0, st, reart_batched(ak, K), adb, bkb);
    REART_CHECK_LAUNCH();
    return REART_OK;
}

int reart_base_backward_ex(BaseBwdArgs a, const FinalizeAdam *adam, const StepBook *book, void *workspace,
                           size_t workspace_bytes, hipStream_t st) {
    void *w[1] = {workspace};
    return reart_base_backward_batch(&a, adam, book, w, workspace_bytes, 1, st);
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
    return reart_base_backward_ex(a, nullptr, nullptr, workspace, workspace_bytes, (hipStream_t)stream);
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

// reart_adam_step for up to 8 tensors in ONE launch (the pointer and size arrays are host arrays, read at the call): the three
// parameter tensors of the kinematic model were three launches, each a 5 us slot of the iteration whatever its size.
extern "C" int reart_adam_step_multi(int count, float *const *param, const float *const *grad, float *const *exp_avg,
                                     float *const *exp_avg_sq, const int *n, const float *lr, int step, float beta1, float beta2,
                                     float eps, void *stream) {
    if (count < 0 || count > 8 || step < 1) return REART_ERR_INVALID_ARG;
    if (count == 0) return REART_OK;
    if (!param || !grad || !exp_avg || !exp_avg_sq || !n || !lr) return REART_ERR_INVALID_ARG;
    AdamArgs a = {};
    for (int k = 0; k < count; ++k) {
        if (n[k] < 0) return REART_ERR_INVALID_ARG;
        if (n[k] > 0 && (!param[k] || !grad[k] || !exp_avg[k] || !exp_avg_sq[k])) return REART_ERR_INVALID_ARG;
        a.seg[k] = {param[k], grad[k], exp_avg[k], exp_avg_sq[k], n[k], lr[k]};
    }
    a.nseg = count; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.step = step;
    return reart_adam_ex(a, (hipStream_t)stream);
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
