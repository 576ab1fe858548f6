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
// Tile: 128 rows x 32 NB cols per workgroup (4 waves, each 32 x 32 NB = NB accumulators sharing
// the A fragment), K step 16 through LDS, the global loads of step i+1 in flight (registers) while step i
// runs on the matrix cores; 16-byte loads for contiguous row segments.  The A tile can be GATHERED on the fly from the ball
// query indices (grouped features | relative xyz), so the grouped tensor [B,S,K,C] of the
// reference (up to 400 MB per scale at T=20) is never materialised.
#include "common.h"
#include "internal.h"
#include <math.h>

#define GM_BM 128
// K step per layer (template parameter BK): 8 for the Cin = 6 first layers (they pad to the step), 32 for wide layers
// (half the barriers per flop), 16 otherwise.  One step for all was measured: 32 everywhere is slower than 16 everywhere
// (8.3 vs 6.7 ms for the extractor: the Cin = 6 layers pad twice as far).
// LDA = BK + 1: column reads by 32 lanes hit 32 different banks

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

// NB = number of 32-column accumulators per wave: the workgroup tile is 128 rows x 32 NB columns.  The
// layers pick the NB that wastes the fewest columns (Cout = 32 -> 1, 64 -> 2, 96 -> 3); wider layers use
// NB = 2 (NB = 4 reuses the A fragment four times but halves the resident waves: measured slower).
template <int NB, int BK>
__global__ __launch_bounds__(256) void mlp_gemm_kernel(GemmArgs a) {
    constexpr int BN = 32 * NB, LDB = BN + 4;
    constexpr int GM_BK = BK, GM_LDA = BK + 1;
    __shared__ float As[GM_BM * GM_LDA];
    __shared__ float Bs[GM_BK * LDB];
    __shared__ float Pm[4][BN];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int row0 = blockIdx.x * GM_BM, col0 = blockIdx.y * BN;
    f16v c[NB];
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) c[n][r] = 0.f;
    constexpr int AK = GM_BK / 2;                     // k per thread of the A tile
    const int ar = tid >> 1, ak = (tid & 1) * AK;     // A tile: 128 rows x 2 half-rows of AK
    constexpr int NBF = GM_BK * BN / 4;               // float4 of the B tile
    constexpr int BPT = (NBF + 255) / 256;            // per thread
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
    auto load_b = [&](int k0, int f) {                // float4 number f of the tile: row f / (BN/4), column group f % (BN/4)
        const int k = k0 + f / (BN / 4), cc = col0 + (f % (BN / 4)) * 4;
        float4 w = {0.f, 0.f, 0.f, 0.f};
        if (f < NBF && k < a.Cin) {
            if (cc + 3 < a.Cout && (a.Cout & 3) == 0) {
                w = *(const float4 *)(a.Wt + (size_t)k * a.Cout + cc);
            } else {
                if (cc < a.Cout) w.x = a.Wt[(size_t)k * a.Cout + cc];
                if (cc + 1 < a.Cout) w.y = a.Wt[(size_t)k * a.Cout + cc + 1];
                if (cc + 2 < a.Cout) w.z = a.Wt[(size_t)k * a.Cout + cc + 2];
                if (cc + 3 < a.Cout) w.w = a.Wt[(size_t)k * a.Cout + cc + 3];
            }
        }
        return w;
    };
    // register double buffering: the loads of K-step i+1 are in flight while step i runs on the matrix cores
    float av8[AK];
    float4 bw[BPT];
    load_a(0, av8);
#pragma unroll
    for (int h = 0; h < BPT; ++h) bw[h] = load_b(0, tid + 256 * h);
    for (int k0 = 0; k0 < a.Cin; k0 += GM_BK) {
#pragma unroll
        for (int u = 0; u < AK; ++u) As[ar * GM_LDA + ak + u] = av8[u];
#pragma unroll
        for (int h = 0; h < BPT; ++h) {
            const int f = tid + 256 * h;
            if (f < NBF) *(float4 *)(Bs + (f / (BN / 4)) * LDB + (f % (BN / 4)) * 4) = bw[h];
        }
        __syncthreads();
        if (k0 + GM_BK < a.Cin) {
            load_a(k0 + GM_BK, av8);
#pragma unroll
            for (int h = 0; h < BPT; ++h) bw[h] = load_b(k0 + GM_BK, tid + 256 * h);
        }
#pragma unroll
        for (int kk = 0; kk < GM_BK; kk += 2) {
            const int kl = kk + (lane >> 5);
            const float av = As[(wv * 32 + (lane & 31)) * GM_LDA + kl];
#pragma unroll
            for (int n = 0; n < NB; ++n)
                c[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, Bs[kl * LDB + 32 * n + (lane & 31)], c[n], 0, 0, 0);
        }
        __syncthreads();
    }
    // epilogue: C/D layout row = (reg&3) + 8*(reg>>2) + 4*(lane>>5), col = lane & 31
    float mx[NB];
#pragma unroll
    for (int n = 0; n < NB; ++n) {
        const int cn = col0 + 32 * n + (lane & 31);
        const float bias = (a.bias && cn < a.Cout) ? a.bias[cn] : 0.f;
        mx[n] = -INFINITY;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int r = row0 + wv * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
            float v = c[n][reg] + bias;
            if (a.relu) v = v > 0.f ? v : 0.f;
            if (a.pool_k) {
                if (r < a.rows) mx[n] = fmaxf(mx[n], v);
            } else if (r < a.rows && cn < a.Cout) {
                a.Y[(size_t)r * a.ldy + a.ycol0 + cn] = v;
            }
        }
    }
    if (!a.pool_k) return;
    // max over the wave's 32 rows, then over pool_k / 32 waves
#pragma unroll
    for (int n = 0; n < NB; ++n) {
        mx[n] = fmaxf(mx[n], __shfl_xor(mx[n], 32, 64));
        if (lane < 32) Pm[wv][32 * n + lane] = mx[n];
    }
    __syncthreads();
    const int wpg = a.pool_k / 32;                 // waves per pooled group: 1, 2 or 4
    const int groups = 4 / wpg;
    for (int e = tid; e < groups * BN; e += 256) {
        const int g = e / BN, cc = e % BN;
        float m = -INFINITY;
        for (int w = 0; w < wpg; ++w) m = fmaxf(m, Pm[g * wpg + w][cc]);
        const int prow = (row0 + g * a.pool_k) / a.pool_k;
        if (row0 + g * a.pool_k < a.rows && col0 + cc < a.Cout) a.Y[(size_t)prow * a.ldy + a.ycol0 + col0 + cc] = m;
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
    // measured: NB = 4 for ALL wide layers is slower (7.5 vs 6.7 ms: half the resident waves), and for the gathered wide-K
    // first layers of sa2 (323 -> 128, whose A tile NB = 2 stages twice) alone as well (5.69 vs 5.57 ms per forward)
    // Cout = 196 (sa2's 128 -> 196): four 64-column blocks compute 256 columns for 196; ONE block of seven 32-column
    // accumulators computes 224 and stages the A tile once
    const int NB = Cout <= 32 ? 1 : (Cout <= 64 ? 2 : (Cout <= 96 ? 3 : ((Cout > 192 && Cout <= 224) ? 7 : 2)));
    const int BK = Cin <= 8 ? 8 : 16;   // measured: 32 for the wide layers is slower (7.76 vs 6.72 ms for the extractor)
    const dim3 grid(reart_div_up(rows, GM_BM), reart_div_up(Cout, 32 * NB));
    hipStream_t st = (hipStream_t)stream;
#define GM_LAUNCH(NBv, BKv) hipLaunchKernelGGL((mlp_gemm_kernel<NBv, BKv>), grid, dim3(256), 0, st, a)
#define GM_PICK(NBv) do { if (BK == 8) GM_LAUNCH(NBv, 8); else if (BK == 32) GM_LAUNCH(NBv, 32); else GM_LAUNCH(NBv, 16); } while (0)
    switch (NB) {
        case 1: GM_PICK(1); break;
        case 2: GM_PICK(2); break;
        case 4: GM_PICK(4); break;
        case 7: GM_PICK(7); break;
        default: GM_PICK(3); break;
    }
#undef GM_PICK
#undef GM_LAUNCH
    REART_CHECK_LAUNCH();
    return REART_OK;
}

// ---------------------------------------------------------------------------------------
// Three layers in ONE launch for the first set-abstraction level (PointNetSetAbstractionMsg of sa1,
// networks/pointnet2_utils.py:257-295 with networks/feature_extractor.py:19-21: grouped [features(3) | xyz - centre(3)]
// -> C1 -> C2 -> C3 -> max over the K samples of a group).  Layer by layer the two narrow layers of every scale were bound
// by memory, not by the matrix cores: their activations ([B S K, C] floats, 0.6 .. 1.2 GB per scale at T = 20) were
// written by one launch and read back by the next at 3-4 TB/s.  Here a wave keeps its 32 rows through all three layers:
//   * the weights of the three layers live in LDS for the lifetime of a (persistent) workgroup, which loops over row tiles;
//   * layer 1 (K = 6) takes its A fragments straight from the gather; its output goes through the wave's PRIVATE
//     32 x C activation tile in LDS -- from the accumulator layout (row = f(register, lane / 32), column = lane % 32) to
//     the A-fragment layout (row = lane % 32, k = lane / 32) -- and feeds layer 2, whose output takes the same way into
//     layer 3; no barrier between the layers (LDS operations of a wave complete in order);
//   * the epilogue of layer 3 is the single-layer kernel's: bias, ReLU, max over the wave's rows, the waves of a group
//     meet in LDS.
// Every accumulator sees its k in the same ascending order through the same instruction as in mlp_gemm_kernel (whose
// K padding only adds exact zeros), so the result equals the three-launch path BIT FOR BIT (tests/test_extractor_gpu.py).
// LDS strides: weights C + 4 (as in mlp_gemm_kernel), activations C + 2 (= 2 mod 4: the 32 rows of a fragment fall on 32
// distinct even banks, its second k on the odd ones).
struct Chain3Args {
    const int64_t *idx; int K, S, Npts;
    const float *F, *Q, *C;           // features [B*Npts,3], xyz [B*Npts,3], centres [B*S,3]
    const float *W1, *b1, *W2, *b2, *W3, *b3;   // transposed weights [Cin,Cout], biases
    int rows;
    float *Y; int ldy, ycol0;
};
__host__ __device__ constexpr int chain_ldb(int c) { return c + 4; }   // like mlp_gemm_kernel's B tile
__host__ __device__ constexpr int chain_lda(int c) { return c + 2; }   // (measured in mlp_chain_wide_kernel: an odd stride, conflict-free for ds_read_b32's 32 banks, is 1 % slower)

// Weights in LDS as B-fragment images: [k][lr][S] with S = 1, 2 or 4 floats (the NB values W[k][32 n + lr] of one lane next
// to each other), so one LDS read of 4, 8 or 16 bytes fetches a step's B operands (see mlp_chain_wide_kernel: the
// instructions issued between MFMAs are what idles the matrix cores with one wave per SIMD).
__host__ __device__ constexpr int chain_s(int nb) { return nb <= 1 ? 1 : (nb == 2 ? 2 : 4); }
template <int NB>
__device__ __forceinline__ void chain_bfrag(const float *__restrict__ bp, float (&b)[NB]) {
    if (NB == 1) b[0] = bp[0];
    else if (NB == 2) { const float2 v = *(const float2 *)bp; b[0] = v.x; b[1] = v.y; }
    else {
        const float4 v = *(const float4 *)bp;
        b[0] = v.x; b[1] = v.y; b[2] = v.z;
        if (NB > 3) b[3] = v.w;
    }
}
// (The loop is left to the compiler's scheduler: with two to four accumulators a group of MFMAs is too short for the
// hand-fenced two-stage pipeline of mlp_chain_wide_kernel, which was measured slower here, 939 vs 874 us.)
template <int NB, int CIN>
__device__ __forceinline__ void chain_layer(const float *__restrict__ Hw, int lda, const float *__restrict__ Wimg, f16v (&acc)[NB], int lane) {
    constexpr int S = chain_s(NB), KSTR = 32 * S;
    const int kh = lane >> 5, lr = lane & 31;
    const float *ap = Hw + lr * lda + kh, *bp = Wimg + (kh * 32 + lr) * S;
#pragma unroll 4
    for (int kk = 0; kk < CIN; kk += 2) {
        const float av = ap[kk];
        float b[NB];
        chain_bfrag<NB>(bp + kk * KSTR, b);
#pragma unroll
        for (int n = 0; n < NB; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b[n], acc[n], 0, 0, 0);
    }
}
// bias + ReLU of an accumulator tile, written into the wave's activation tile in A-fragment order
template <int NB>
__device__ __forceinline__ void chain_store(float *__restrict__ Hw, int lda, const float *__restrict__ bias, const f16v (&acc)[NB], int lane) {
    const int kh = lane >> 5, lr = lane & 31;
#pragma unroll
    for (int n = 0; n < NB; ++n) {
        const float bv = bias[32 * n + lr];
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int row = (reg & 3) + 8 * (reg >> 2) + 4 * kh;
            const float v = acc[n][reg] + bv;
            Hw[row * lda + 32 * n + lr] = v > 0.f ? v : 0.f;
        }
    }
}

template <int C1, int C2, int C3, int PK>
__global__ __launch_bounds__(256) void mlp_chain3_kernel(Chain3Args a) {
    constexpr int NB1 = C1 / 32, NB2 = C2 / 32, NB3 = C3 / 32;
    constexpr int LDW1 = chain_ldb(C1), S2 = chain_s(NB2), S3 = chain_s(NB3);
    constexpr int LDA = chain_lda(C1 > C2 ? C1 : C2);
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *W2s = sm;                       // [C1][32][S2]: B-fragment image of W2
    float *W3s = W2s + C1 * 32 * S2;       // [C2][32][S3]
    float *W1s = W3s + C2 * 32 * S3;       // [6][LDW1]
    float *Bs = W1s + 6 * LDW1;            // b1 | b2 | b3
    float *H = Bs + (C1 + C2 + C3);        // [4][32][LDA]
    float *Pm = H + 4 * 32 * LDA;          // [4][C3]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, kh = lane >> 5, lr = lane & 31;
    for (int e = tid; e < 6 * C1; e += 256) W1s[(e / C1) * LDW1 + e % C1] = a.W1[e];
    for (int e = tid; e < C1 * 32 * S2; e += 256) {                      // image element (k, lr, n) <- W2[k][32 n + lr]
        const int k = e / (32 * S2), lr2 = (e / S2) & 31, n = e % S2;
        W2s[e] = n < NB2 ? a.W2[k * C2 + 32 * n + lr2] : 0.f;
    }
    for (int e = tid; e < C2 * 32 * S3; e += 256) {
        const int k = e / (32 * S3), lr2 = (e / S3) & 31, n = e % S3;
        W3s[e] = n < NB3 ? a.W3[k * C3 + 32 * n + lr2] : 0.f;
    }
    for (int e = tid; e < C1; e += 256) Bs[e] = a.b1[e];
    for (int e = tid; e < C2; e += 256) Bs[C1 + e] = a.b2[e];
    for (int e = tid; e < C3; e += 256) Bs[C1 + C2 + e] = a.b3[e];
    __syncthreads();
    float *Hw = H + wv * 32 * LDA;
    const int ntiles = (a.rows + GM_BM - 1) / GM_BM;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int row0 = tile * GM_BM, r = row0 + wv * 32 + lr;
        const bool live = row0 + wv * 32 < a.rows;          // rows % 32 == 0: a wave's rows are all inside or all outside
        float mx[NB3];
#pragma unroll
        for (int n = 0; n < NB3; ++n) mx[n] = -INFINITY;
        if (live) {
            // ---- layer 1: A fragments from the gather, row r = [F(3) | Q - C(3)], this lane's k = kh, 2 + kh, 4 + kh
            const int bb = r / (a.S * a.K);
            const size_t prow = (size_t)bb * a.Npts + (size_t)a.idx[r];
            const float *f = a.F + prow * 3, *q = a.Q + prow * 3, *c = a.C + (size_t)(r / a.K) * 3;
            const float x0 = f[0], x1 = f[1], x2 = f[2], x3 = q[0] - c[0], x4 = q[1] - c[1], x5 = q[2] - c[2];
            const float xa[3] = {kh ? x1 : x0, kh ? x3 : x2, kh ? x5 : x4};
            {
                f16v acc[NB1];
#pragma unroll
                for (int n = 0; n < NB1; ++n)
#pragma unroll
                    for (int g = 0; g < 16; ++g) acc[n][g] = 0.f;
#pragma unroll
                for (int s3 = 0; s3 < 3; ++s3)
#pragma unroll
                    for (int n = 0; n < NB1; ++n)
                        acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[s3], W1s[(2 * s3 + kh) * LDW1 + 32 * n + lr], acc[n], 0, 0, 0);
                chain_store<NB1>(Hw, LDA, Bs, acc, lane);
            }
            {   // ---- layer 2
                f16v acc[NB2];
#pragma unroll
                for (int n = 0; n < NB2; ++n)
#pragma unroll
                    for (int g = 0; g < 16; ++g) acc[n][g] = 0.f;
                chain_layer<NB2, C1>(Hw, LDA, W2s, acc, lane);
                chain_store<NB2>(Hw, LDA, Bs + C1, acc, lane);
            }
            {   // ---- layer 3 + bias + ReLU + max over the wave's 32 rows
                f16v acc[NB3];
#pragma unroll
                for (int n = 0; n < NB3; ++n)
#pragma unroll
                    for (int g = 0; g < 16; ++g) acc[n][g] = 0.f;
                chain_layer<NB3, C2>(Hw, LDA, W3s, acc, lane);
#pragma unroll
                for (int n = 0; n < NB3; ++n) {
                    const float bv = Bs[C1 + C2 + 32 * n + lr];
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        const float v = acc[n][reg] + bv;
                        mx[n] = fmaxf(mx[n], v > 0.f ? v : 0.f);
                    }
                }
            }
        }
#pragma unroll
        for (int n = 0; n < NB3; ++n) {
            mx[n] = fmaxf(mx[n], __shfl_xor(mx[n], 32, 64));
            if (lane < 32) Pm[wv * C3 + 32 * n + lane] = mx[n];
        }
        __syncthreads();
        constexpr int wpg = PK / 32, groups = 4 / wpg;         // waves per pooled group: 1, 2 or 4
        for (int e = tid; e < groups * C3; e += 256) {
            const int g = e / C3, cc = e % C3;
            float m = -INFINITY;
#pragma unroll
            for (int w = 0; w < wpg; ++w) m = fmaxf(m, Pm[(g * wpg + w) * C3 + cc]);
            if (row0 + g * PK < a.rows) a.Y[(size_t)((row0 + g * PK) / PK) * a.ldy + a.ycol0 + cc] = m;
        }
        __syncthreads();                                       // Pm is rewritten by the next tile
    }
}

template <int C1, int C2, int C3, int PK>
static int chain3_launch(const Chain3Args &a, hipStream_t st) {
    constexpr size_t lds = sizeof(float) * (6 * chain_ldb(C1) + C1 * 32 * chain_s(C2 / 32) + C2 * 32 * chain_s(C3 / 32) + (C1 + C2 + C3) +
                                            4 * 32 * chain_lda(C1 > C2 ? C1 : C2) + 4 * C3);
    static_assert(lds <= 152 * 1024, "chain3: weights + activation tiles must fit the LDS of one compute unit");
    if (lds > REART_LDS_DEFAULT_CAP &&
        hipFuncSetAttribute((const void *)mlp_chain3_kernel<C1, C2, C3, PK>, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024) != hipSuccess)
        return REART_ERR_LAUNCH;
    const int ntiles = reart_div_up(a.rows, GM_BM);
    const int wgs = (int)(160 * 1024 / (lds + 1024));          // workgroups one compute unit holds (LDS bound)
    const int grid = ntiles < 256 * wgs ? ntiles : 256 * wgs;   // persistent: every workgroup loads the weights once
    hipLaunchKernelGGL((mlp_chain3_kernel<C1, C2, C3, PK>), dim3(grid), dim3(256), lds, st, a);
    REART_CHECK_LAUNCH();
    return REART_OK;
}

extern "C" int reart_mlp_chain3(const int64_t *gather_idx, int K, int S, int Npts, const float *F, const float *Q, const float *C,
                                const float *W1t, const float *b1, int C1, const float *W2t, const float *b2, int C2,
                                const float *W3t, const float *b3, int C3, int rows, float *Y, int ldy, int ycol0, void *stream) {
    if (rows < 0 || K < 1 || S < 1 || Npts < 1) return REART_ERR_INVALID_ARG;
    if (rows == 0) return REART_OK;
    if (!gather_idx || !F || !Q || !C || !W1t || !b1 || !W2t || !b2 || !W3t || !b3 || !Y) return REART_ERR_INVALID_ARG;
    if (ycol0 < 0 || ldy < ycol0 + C3 || rows % K != 0) return REART_ERR_INVALID_ARG;
    Chain3Args a = {gather_idx, K, S, Npts, F, Q, C, W1t, b1, W2t, b2, W3t, b3, rows, Y, ldy, ycol0};
    hipStream_t st = (hipStream_t)stream;
    // the three scales of the extractor's sa1 (networks/feature_extractor.py:19-21)
    if (C1 == 32 && C2 == 32 && C3 == 64 && K == 32) return chain3_launch<32, 32, 64, 32>(a, st);
    if (C1 == 64 && C2 == 64 && C3 == 128 && K == 64) return chain3_launch<64, 64, 128, 64>(a, st);
    if (C1 == 64 && C2 == 96 && C3 == 128 && K == 128) return chain3_launch<64, 96, 128, 128>(a, st);
    return REART_ERR_UNSUPPORTED;
}

// ---------------------------------------------------------------------------------------
// The same fusion for the SECOND set-abstraction level (sa2, networks/feature_extractor.py:22-23: grouped
// [features(320) | xyz - centre(3)] -> 128 -> 128 | 196 -> 256 -> max over the K samples): its weights (up to 200 KB a layer)
// do not fit LDS next to the activations, so they stream through it in 16-row slabs like mlp_gemm_kernel's B tile (double
// buffered: one barrier per slab) while a wave's 32 rows stay in its private activation tile from layer to layer.
// Layer by layer the middle activations of the K = 128 scale ([622 592, 128] and [622 592, 196] floats at T = 20: 0.8 GB)
// were written and read back, and the 128 -> 196 layer ran at a third of the matrix cores' rate (seven accumulators, a
// 100 KB store per workgroup).  Here one workgroup owns 128 rows (one group of K = 128 samples, or two of K = 64) through all
// three layers with every output column in registers: 4, then 4 or 7, then 8 accumulators per wave.
// The k order of every accumulator is mlp_gemm_kernel's (ascending, slabs of 16), so the bits are the layer-by-layer path's.
struct ChainWideArgs {
    const int64_t *idx; int K, S, Npts;
    const float *F; int D;            // features [B*Npts, D], D % 4 == 0
    const float *Q, *C;               // xyz [B*Npts,3], centres [B*S,3]
    const float *W1, *b1, *W2, *b2, *W3, *b3;   // transposed weights [D+3, C1], [C1, C2], [C2, C3], biases
    int rows;                         // % 128 == 0
    float *Y; int ldy, ycol0;
};
#define CW_BK 16
// diagnostic build (-DCW_CLOCK): where a wave of mlp_chain_wide_kernel spends its cycles (tools/cw_clock.py)
#ifdef CW_CLOCK
__device__ unsigned long long cw_clock_acc[16];
#define CW_T(v) const long long v = clock64()
struct CwClk { long long v[10]; };
#define CW_CLK_ARG , CwClk &clk
#define CW_CLK_PASS , clk
#define CW_ADD(i, d) (clk.v[i] += (d))
#define CW_FLUSH(i, d) do { if ((threadIdx.x & 63) == 0) atomicAdd(&cw_clock_acc[i], (unsigned long long)(d)); } while (0)
extern "C" int reart_debug_cw_clock(unsigned long long *out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(cw_clock_acc), sizeof(cw_clock_acc)) != hipSuccess) return -1;
    if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(cw_clock_acc), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#else
#define CW_T(v) ((void)0)
#define CW_ADD(i, d) ((void)0)
#define CW_FLUSH(i, d) ((void)0)
#define CW_CLK_ARG
#define CW_CLK_PASS
#endif
#define CW_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#define CW_LDAS (CW_BK + 1)
// Weight slabs in LDS: the B fragment of lane (k-half kh, column lr) for step kk is the NB values W[kk + kh][32 n + lr],
// n = 0 .. NB-1.  They are stored next to each other ([k][lr][n], S = 4 or 8 floats per (k, lr)) so that ONE or TWO 16-byte
// LDS reads fetch them -- with the row-major slab ([k][column]) it took NB / 2 ds_read2_b32, and the instructions a wave
// issues between its MFMAs are what idles the matrix cores (tools/ubench_mfma.hip: 64 cycles per MFMA from registers, 92 -
// 102 with a fragment read per accumulator).  The weights come in that order from an image of the matrix (cw_image_kernel,
// once per call), so the slab's global -> LDS copy is a straight 16-byte copy.  LDS stride per (k, lr): 12 floats for S = 8
// (the 16-lane groups of ds_read_b128 then cover the 64 banks exactly), 4 for S = 4.
template <int NBp> struct CwSlab {
    static constexpr int S = NBp <= 4 ? 4 : 8, SL = S == 8 ? 12 : 4;
    static constexpr int KSTR = 32 * SL;                       // floats per k row in LDS
    static constexpr int NBF = CW_BK * 32 * S / 4, BPT = (NBF + 255) / 256;
};
#define CW_SLAB_FLOATS (CW_BK * 32 * 12)
// image of a transposed weight matrix W [Cin, Cout]: img[k][lr][n] = W[k][32 n + lr] (zero beyond Cout), n < S
__global__ __launch_bounds__(256) void cw_image_kernel(const float *__restrict__ W1, int Cin1, int Cout1, int S1, float *__restrict__ img1,
                                                       const float *__restrict__ W2, int Cin2, int Cout2, int S2, float *__restrict__ img2,
                                                       const float *__restrict__ W3, int Cin3, int Cout3, int S3, float *__restrict__ img3) {
    const float *W = blockIdx.y == 0 ? W1 : (blockIdx.y == 1 ? W2 : W3);
    float *img = blockIdx.y == 0 ? img1 : (blockIdx.y == 1 ? img2 : img3);
    const int Cin = blockIdx.y == 0 ? Cin1 : (blockIdx.y == 1 ? Cin2 : Cin3), Cout = blockIdx.y == 0 ? Cout1 : (blockIdx.y == 1 ? Cout2 : Cout3);
    const int S = blockIdx.y == 0 ? S1 : (blockIdx.y == 1 ? S2 : S3);
    for (int e = blockIdx.x * 256 + threadIdx.x; e < Cin * 32 * S; e += gridDim.x * 256) {
        const int k = e / (32 * S), lr = (e / S) & 31, n = e % S, col = 32 * n + lr;
        img[e] = col < Cout ? W[(size_t)k * Cout + col] : 0.f;
    }
}
// rows [k0, k0 + 16) of a weight image.  No branches: beyond Cin the address is clamped to the last row -- such a row only
// ever meets an A value of exactly zero (the gather slabs and cw_slab_* see to that), so what is loaded there does not reach
// a result (a finite weight times zero adds a zero of either sign to the accumulator).
template <int NBp>
__device__ __forceinline__ void cw_load_b(const float *__restrict__ img, int Cin, int k0, int tid, float4 (&bw)[CwSlab<NBp>::BPT]) {
    typedef CwSlab<NBp> SL;
#pragma unroll
    for (int h = 0; h < SL::BPT; ++h) {
        const int f = tid + 256 * h < SL::NBF ? tid + 256 * h : SL::NBF - 1;
        const int k = min(k0 + f / (8 * SL::S), Cin - 1), off = f % (8 * SL::S);
        const float4 w = *(const float4 *)(img + ((size_t)k * 32 * SL::S + 4 * off));      // (assigning the load straight to bw[h] kept the array in scratch)
        bw[h] = w;
    }
}
template <int NBp>
__device__ __forceinline__ void cw_store_b(float *__restrict__ Bs, int tid, const float4 (&bw)[CwSlab<NBp>::BPT]) {
    typedef CwSlab<NBp> SL;
#pragma unroll
    for (int h = 0; h < SL::BPT; ++h) {
        const int f = tid + 256 * h, k = f / (8 * SL::S), rem = f % (8 * SL::S), lr = rem / (SL::S / 4), q = rem % (SL::S / 4);
        if (f < SL::NBF) *(float4 *)(Bs + (k * 32 + lr) * SL::SL + 4 * q) = bw[h];
    }
}
// The MFMAs of one slab: A value of step kk at ap[kk], B values at bp[kk * KSTR + n].  One wave per SIMD is resident (the
// tiles fill the LDS), so nothing but the wave's own instruction stream hides the LDS latency: the fragments of step kk + 2
// are fetched into a second register set while the matrix cores have step kk.
template <int NBp>
__device__ __forceinline__ void cw_frag(const float *__restrict__ ap, const float *__restrict__ bp, float &av, float (&bv)[NBp]) {
    av = *ap;
#pragma unroll
    for (int q = 0; q < CwSlab<NBp>::S / 4; ++q) {
        const float4 v = *(const float4 *)(bp + 4 * q);
        if (4 * q < NBp) bv[4 * q] = v.x;
        if (4 * q + 1 < NBp) bv[4 * q + 1] = v.y;
        if (4 * q + 2 < NBp) bv[4 * q + 2] = v.z;
        if (4 * q + 3 < NBp) bv[4 * q + 3] = v.w;
    }
}
template <int NBp> struct CwFrags { float a0, a1, b0[NBp], b1[NBp]; };
// one k-step of a slab: the first MFMA of the group goes out, THEN the LDS reads of the next step are issued (the compiler
// waits with lgkmcnt(0) before a group, i.e. also for reads issued just ahead of it: issued behind the group's first MFMA
// they have the rest of the group -- 192 cycles with four accumulators, 448 with eight -- to arrive)
template <int NBp>
__device__ __forceinline__ void cw_kstep(float a, const float (&b)[NBp], f16v (&acc)[NBp], const float *__restrict__ ap_next,
                                         const float *__restrict__ bp_next, bool fetch, float &a_next, float (&b_next)[NBp]) {
    CW_SCHED_FENCE();
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[0], acc[0], 0, 0, 0);
    CW_SCHED_FENCE();
    if (fetch) cw_frag<NBp>(ap_next, bp_next, a_next, b_next);
    CW_SCHED_FENCE();
#pragma unroll
    for (int n = 1; n < NBp; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[n], acc[n], 0, 0, 0);
    CW_SCHED_FENCE();
}
template <int NBp>
__device__ __forceinline__ void cw_slab_head(const float *__restrict__ ap, const float *__restrict__ bp, int kn, f16v (&acc)[NBp], CwFrags<NBp> &r) {
    constexpr int LDB = CwSlab<NBp>::KSTR;
    const int kh = (threadIdx.x >> 5) & 1;
    cw_frag<NBp>(ap, bp, r.a0, r.b0);
    r.a0 = kh < kn ? r.a0 : 0.f;
    cw_kstep<NBp>(r.a0, r.b0, acc, ap + 2, bp + 2 * LDB, true, r.a1, r.b1);
}
template <int NBp>
__device__ __forceinline__ void cw_slab_tail(const float *__restrict__ ap, const float *__restrict__ bp, int kn, f16v (&acc)[NBp], CwFrags<NBp> &r) {
    constexpr int LDB = CwSlab<NBp>::KSTR;
    const int kh = (threadIdx.x >> 5) & 1;
#pragma unroll
    for (int kk = 2; kk < CW_BK; kk += 4) {
        r.a1 = kk + kh < kn ? r.a1 : 0.f;
        cw_kstep<NBp>(r.a1, r.b1, acc, ap + (kk + 2), bp + (kk + 2) * LDB, kk + 2 < CW_BK, r.a0, r.b0);
        if (kk + 2 < CW_BK) {
            r.a0 = kk + 2 + kh < kn ? r.a0 : 0.f;
            cw_kstep<NBp>(r.a0, r.b0, acc, ap + (kk + 4), bp + (kk + 4) * LDB, kk + 4 < CW_BK, r.a1, r.b1);
        }
    }
}
// one layer whose A operand is the wave's activation tile: Cin of its columns against the slabs of W [Cin, Cout]
// (bw: the layer's first slab, fetched by the caller while the previous layer was still on the matrix cores)
template <int NBp>
__device__ __forceinline__ void cw_layer(const float *__restrict__ Hw, int lda, const float *__restrict__ W, int Cin, int Cout,
                                         float *__restrict__ Bs2, int &buf, f16v (&acc)[NBp], int tid, int lane,
                                         float4 (&bw)[CwSlab<NBp>::BPT] CW_CLK_ARG) {
    typedef CwSlab<NBp> SL;
    const int kh = lane >> 5, lr = lane & 31;
    for (int k0 = 0; k0 < Cin; k0 += CW_BK, buf ^= 1) {
        float *Bs = Bs2 + buf * CW_SLAB_FLOATS;
        CW_T(t0);
        cw_store_b<NBp>(Bs, tid, bw);
        CW_T(t1);
        __syncthreads();
        CW_T(t2);
        const int kn = Cin - k0 < CW_BK ? Cin - k0 : CW_BK;
        CwFrags<NBp> fr;
        cw_slab_head<NBp>(Hw + lr * lda + k0 + kh, Bs + (kh * 32 + lr) * SL::SL, kn, acc, fr);
        CW_T(t3);
        if (k0 + CW_BK < Cin) cw_load_b<NBp>(W, Cin, k0 + CW_BK, tid, bw);
        cw_slab_tail<NBp>(Hw + lr * lda + k0 + kh, Bs + (kh * 32 + lr) * SL::SL, kn, acc, fr);
        CW_T(t4);
        CW_ADD(0, t1 - t0); CW_ADD(1, t2 - t1); CW_ADD(2, t3 - t2); CW_ADD(3, t4 - t3); CW_ADD(4, 1);
    }
}
// bias + ReLU of an accumulator tile into the wave's activation tile, columns < Cout only
template <int NBp>
__device__ __forceinline__ void cw_store(float *__restrict__ Hw, int lda, const float *__restrict__ bias, int Cout, const f16v (&acc)[NBp], int lane) {
    const int kh = lane >> 5, lr = lane & 31;
#pragma unroll
    for (int n = 0; n < NBp; ++n) {
        const int cn = 32 * n + lr;
        if (cn < Cout) {
            const float bv = bias[cn];
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int row = (reg & 3) + 8 * (reg >> 2) + 4 * kh;
                const float v = acc[n][reg] + bv;
                Hw[row * lda + cn] = v > 0.f ? v : 0.f;
            }
        }
    }
}

template <int C1, int C2, int C3, int PK>
__global__ __launch_bounds__(256) void mlp_chain_wide_kernel(ChainWideArgs a) {
    constexpr int NB1 = C1 / 32, NB2 = (C2 + 31) / 32, NB3 = C3 / 32;
    static_assert(C1 % 32 == 0 && C3 % 32 == 0 && C2 % 4 == 0 && NB1 <= 8 && NB2 <= 8 && NB3 <= 8, "chain_wide: widths");
    constexpr int LDA = chain_lda(C1 > C2 ? C1 : C2);
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *Bs2 = sm;                                   // [2][16][32][12]: weight slabs
    float *H = Bs2 + 2 * CW_SLAB_FLOATS;               // [4][32][LDA]: the waves' activation tiles
    float *As2 = H;                                    // [2][128][17]: the gathered A slabs of layer 1 live in the (not yet used) tiles
    float *Pm = H + 4 * 32 * LDA;                      // [4][C3]
    static_assert(2 * GM_BM * CW_LDAS <= 4 * 32 * LDA, "chain_wide: the gather slabs alias the activation tiles");
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, kh = lane >> 5, lr = lane & 31;
    float *Hw = H + wv * 32 * LDA;
    CW_T(tk0);
#ifdef CW_CLOCK
    CwClk clk = {{0, 0, 0, 0, 0, 0, 0, 0, 0, 0}};
#endif
    const int row0 = blockIdx.x * GM_BM;
    // kernel arguments into locals: a lambda that captured the argument struct by reference kept it in scratch, and every
    // pointer loaded back from there had lost its address space (flat loads count against the LDS counter too)
    const int D = a.D, Cin1 = D + 3;
    const float *const W1 = a.W1, *const W2 = a.W2, *const W3 = a.W3, *const b1 = a.b1, *const b2 = a.b2, *const b3 = a.b3;
    int buf = 0;
    float4 bw2[CwSlab<NB2>::BPT], bw3[CwSlab<NB3>::BPT];
    cw_load_b<NB2>(W2, C1, 0, tid, bw2);           // layer 2's first slab waits in registers through layer 1
    {   // ---- layer 1: A from the gather (mlp_gemm_kernel's staging: thread <- 8 consecutive k of one row)
        const int ar = tid >> 1, ak = (tid & 1) * 8, r = row0 + ar;
        const size_t prow = (size_t)(r / (a.S * a.K)) * a.Npts + (size_t)a.idx[r];
        const float *f = a.F + prow * a.D, *q = a.Q + prow * 3, *c = a.C + (size_t)(r / a.K) * 3;
        // the row's last three columns, xyz - centre, as bit patterns: the slab that holds them is put together with masks
        // (a chain of selects over run-time values became a table in scratch, and the pointers kept next to it went flat)
        const int x0 = __float_as_int(q[0] - c[0]), x1 = __float_as_int(q[1] - c[1]), x2 = __float_as_int(q[2] - c[2]);
        auto load_a = [=](int k0, float (&v)[8]) __attribute__((always_inline)) {
            const int k = k0 + ak;
            if (k + 8 <= D) {
                const float4 p0 = *(const float4 *)(f + k), p1 = *(const float4 *)(f + k + 4);
                v[0] = p0.x; v[1] = p0.y; v[2] = p0.z; v[3] = p0.w; v[4] = p1.x; v[5] = p1.y; v[6] = p1.z; v[7] = p1.w;
            } else {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int kx = k + u - D;
                    const int fv = __float_as_int(f[kx < 0 ? k + u : D - 1]);          // a valid address either way
                    v[u] = __int_as_float((fv & -(int)(kx < 0)) | (x0 & -(int)(kx == 0)) | (x1 & -(int)(kx == 1)) | (x2 & -(int)(kx == 2)));
                }
            }
        };
        f16v acc[NB1];
#pragma unroll
        for (int n = 0; n < NB1; ++n)
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[n][g] = 0.f;
        typedef CwSlab<NB1> SL;
        // The gathered rows come from all over a cloud's feature matrix (25 MB at T = 20: beyond L2), and with one workgroup
        // per compute unit nobody else hides that latency: a slab keeps the matrix cores busy for 2048 cycles, a row takes
        // longer to arrive.  Three slabs of rows are in flight (three register sets, the loop unrolled over them).
        float av0[8], av1[8], av2[8];
        float4 bw[SL::BPT];
        load_a(0, av0);
        load_a(CW_BK, av1);                            // (beyond Cin1 a set is simply never stored)
        load_a(2 * CW_BK, av2);
        cw_load_b<NB1>(W1, Cin1, 0, tid, bw);
#define CW_GATHER_SLAB(K0, AV)                                                                                              \
        if ((K0) < Cin1) {                                                                                                   \
            float *As = As2 + buf * (GM_BM * CW_LDAS), *Bs = Bs2 + buf * CW_SLAB_FLOATS;                                     \
            CW_T(g0);                                                                                                        \
            _Pragma("unroll") for (int u = 0; u < 8; ++u) As[ar * CW_LDAS + ak + u] = AV[u];                                 \
            cw_store_b<NB1>(Bs, tid, bw);                                                                                    \
            CW_T(g1);                                                                                                        \
            __syncthreads();                                                                                                 \
            CW_T(g2);                                                                                                        \
            CwFrags<NB1> fr;                                                                                                 \
            cw_slab_head<NB1>(As + (wv * 32 + lr) * CW_LDAS + kh, Bs + (kh * 32 + lr) * SL::SL, CW_BK, acc, fr);             \
            CW_T(g3);                                                                                                        \
            if ((K0) + 3 * CW_BK < Cin1) load_a((K0) + 3 * CW_BK, AV);                                                       \
            if ((K0) + CW_BK < Cin1) cw_load_b<NB1>(W1, Cin1, (K0) + CW_BK, tid, bw);                                        \
            cw_slab_tail<NB1>(As + (wv * 32 + lr) * CW_LDAS + kh, Bs + (kh * 32 + lr) * SL::SL, CW_BK, acc, fr);             \
            CW_T(g4);                                                                                                        \
            CW_ADD(5, g1 - g0); CW_ADD(6, g2 - g1); CW_ADD(7, g3 - g2); CW_ADD(8, g4 - g3); CW_ADD(9, 1);                    \
            buf ^= 1;                                                                                                        \
        }
        for (int k0 = 0; k0 < Cin1; k0 += 3 * CW_BK) {
            CW_GATHER_SLAB(k0, av0)
            CW_GATHER_SLAB(k0 + CW_BK, av1)
            CW_GATHER_SLAB(k0 + 2 * CW_BK, av2)
        }
#undef CW_GATHER_SLAB
        __syncthreads();                               // the gather slabs share the tiles' space: everybody has read them
        cw_store<NB1>(Hw, LDA, b1, C1, acc, lane);
    }
    {   // ---- layer 2
        f16v acc[NB2];
#pragma unroll
        for (int n = 0; n < NB2; ++n)
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[n][g] = 0.f;
        cw_load_b<NB3>(W3, C2, 0, tid, bw3);                      // layer 3's first slab a layer ahead, like layer 2's
        cw_layer<NB2>(Hw, LDA, W2, C1, C2, Bs2, buf, acc, tid, lane, bw2 CW_CLK_PASS);
        cw_store<NB2>(Hw, LDA, b2, C2, acc, lane);   // the wave's own tile, after its own last read of it
    }
    float mx[NB3];
    {   // ---- layer 3 + bias + ReLU + max over the wave's 32 rows
        f16v acc[NB3];
#pragma unroll
        for (int n = 0; n < NB3; ++n)
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[n][g] = 0.f;
        cw_layer<NB3>(Hw, LDA, W3, C2, C3, Bs2, buf, acc, tid, lane, bw3 CW_CLK_PASS);
#pragma unroll
        for (int n = 0; n < NB3; ++n) {
            const float bv = b3[32 * n + lr];
            mx[n] = -INFINITY;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const float v = acc[n][reg] + bv;
                mx[n] = fmaxf(mx[n], v > 0.f ? v : 0.f);
            }
        }
    }
#pragma unroll
    for (int n = 0; n < NB3; ++n) {
        mx[n] = fmaxf(mx[n], __shfl_xor(mx[n], 32, 64));
        if (lane < 32) Pm[wv * C3 + 32 * n + lane] = mx[n];
    }
    __syncthreads();
    constexpr int wpg = PK / 32, groups = 4 / wpg;             // waves per pooled group: 2 or 4
    for (int e = tid; e < groups * C3; e += 256) {
        const int g = e / C3, cc = e % C3;
        float m = -INFINITY;
#pragma unroll
        for (int w = 0; w < wpg; ++w) m = fmaxf(m, Pm[(g * wpg + w) * C3 + cc]);
        a.Y[(size_t)((row0 + g * PK) / PK) * a.ldy + a.ycol0 + cc] = m;
    }
    CW_T(tk1);
#ifdef CW_CLOCK
    for (int i = 0; i < 5; ++i) CW_FLUSH(i, clk.v[i]);
    for (int i = 5; i < 10; ++i) CW_FLUSH(i + 3, clk.v[i]);
#endif
    CW_FLUSH(5, tk1 - tk0); CW_FLUSH(6, 1);
}

static size_t cw_image_floats(int Cin, int Cout) { return (size_t)Cin * 32 * (Cout <= 128 ? 4 : 8); }
extern "C" size_t reart_mlp_chain3_wide_workspace_bytes(int D, int C1, int C2, int C3) {
    if (D < 1 || C1 < 1 || C2 < 1 || C3 < 1 || C1 > 256 || C2 > 256 || C3 > 256) return 0;
    return sizeof(float) * (reart_align_up(cw_image_floats(D + 3, C1), 64) + reart_align_up(cw_image_floats(C1, C2), 64) +
                            reart_align_up(cw_image_floats(C2, C3), 64));
}

template <int C1, int C2, int C3, int PK>
static int chain_wide_launch(ChainWideArgs a, float *ws, hipStream_t st) {
    constexpr size_t lds = sizeof(float) * (2 * CW_SLAB_FLOATS + 4 * 32 * chain_lda(C1 > C2 ? C1 : C2) + 4 * C3);
    static_assert(lds <= 152 * 1024, "chain_wide: weight slabs + activation tiles must fit the LDS of one compute unit");
    if (lds > REART_LDS_DEFAULT_CAP &&
        hipFuncSetAttribute((const void *)mlp_chain_wide_kernel<C1, C2, C3, PK>, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024) != hipSuccess)
        return REART_ERR_LAUNCH;
    // the three weight images (B fragments contiguous per lane), then the layers on them
    constexpr int S1 = CwSlab<C1 / 32>::S, S2 = CwSlab<(C2 + 31) / 32>::S, S3 = CwSlab<C3 / 32>::S;
    float *img1 = ws, *img2 = img1 + reart_align_up(cw_image_floats(a.D + 3, C1), 64), *img3 = img2 + reart_align_up(cw_image_floats(C1, C2), 64);
    hipLaunchKernelGGL(cw_image_kernel, dim3(64, 3), dim3(256), 0, st, a.W1, a.D + 3, C1, S1, img1, a.W2, C1, C2, S2, img2, a.W3, C2, C3, S3, img3);
    REART_CHECK_LAUNCH();
    a.W1 = img1; a.W2 = img2; a.W3 = img3;
    hipLaunchKernelGGL((mlp_chain_wide_kernel<C1, C2, C3, PK>), dim3(a.rows / GM_BM), dim3(256), lds, st, a);
    REART_CHECK_LAUNCH();
    return REART_OK;
}

extern "C" int reart_mlp_chain3_wide(const int64_t *gather_idx, int K, int S, int Npts, const float *F, int D, const float *Q,
                                     const float *C, const float *W1t, const float *b1, int C1, const float *W2t, const float *b2,
                                     int C2, const float *W3t, const float *b3, int C3, int rows, float *Y, int ldy, int ycol0,
                                     void *workspace, size_t workspace_bytes, void *stream) {
    if (rows < 0 || K < 1 || S < 1 || Npts < 1 || D < 4) return REART_ERR_INVALID_ARG;
    if (rows == 0) return REART_OK;
    if (!gather_idx || !F || !Q || !C || !W1t || !b1 || !W2t || !b2 || !W3t || !b3 || !Y) return REART_ERR_INVALID_ARG;
    if (ycol0 < 0 || ldy < ycol0 + C3 || rows % K != 0) return REART_ERR_INVALID_ARG;
    if (rows % GM_BM != 0 || D % 4 != 0 || (((size_t)F) & 15) != 0) return REART_ERR_UNSUPPORTED;
    const bool known = (C1 == 128 && C2 == 128 && C3 == 256 && K == 64) || (C1 == 128 && C2 == 196 && C3 == 256 && K == 128);
    if (!known) return REART_ERR_UNSUPPORTED;
    if (!workspace || (((size_t)workspace) & 15) != 0 || workspace_bytes < reart_mlp_chain3_wide_workspace_bytes(D, C1, C2, C3)) return REART_ERR_INVALID_ARG;
    ChainWideArgs a = {gather_idx, K, S, Npts, F, D, Q, C, W1t, b1, W2t, b2, W3t, b3, rows, Y, ldy, ycol0};
    hipStream_t st = (hipStream_t)stream;
    // the two scales of the extractor's sa2 (networks/feature_extractor.py:22-23)
    if (K == 64) return chain_wide_launch<128, 128, 256, 64>(a, (float *)workspace, st);
    return chain_wide_launch<128, 196, 256, 128>(a, (float *)workspace, st);
}

// ---------------------------------------------------------------------------------------
// 3-NN inverse-distance feature interpolation of PointNetFeaturePropagation
// (networks/pointnet2_utils.py:326-336) on the reference's square_distance (:33-55):
//     d = ((-2 * mm) + |q|^2) + |t|^2,   mm = fma(qz, tz, fma(qy, ty, qx * tx))   (torch's K = 3 matmul)
//     |p|^2 = ((x*x) + (y*y)) + (z*z);   the 3 smallest by (d, index);   w = 1/(d + 1e-8), w /= (w0+w1)+w2
//     out = ((p0*w0) + (p1*w1)) + p2*w2
// The matmul expansion is part of the result: the coarse cloud is an FPS subset of the fine one, so every
// coarse point has a query at distance "zero" = +-1e-7 of cancellation noise (negative values included) that
// 1/(d + 1e-8) turns into the dominant weight.  Restated bit for bit by oracle_three_interpolate and pinned by
// tests/golden/three_interp.npz (the reference's own module on this container's CPU).
// out is written into columns [col0, col0+D) of a [B*N, ldo] matrix so that the concatenation
// with the skip features (:338-342) needs no extra pass.
// ---------------------------------------------------------------------------------------
#define TNN_TILE 1024
__global__ __launch_bounds__(256) void three_nn_expanded_kernel(const float *__restrict__ xyz1,
                                                                const float *__restrict__ xyz2, int N, int S,
                                                                float *__restrict__ dist3,
                                                                int64_t *__restrict__ idx3) {
    __shared__ float4 T[TNN_TILE];                 // x, y, z, |t|^2
    const int b = blockIdx.y, n = blockIdx.x * 256 + threadIdx.x;
    const bool live = n < N;
    float qx = 0.f, qy = 0.f, qz = 0.f;
    if (live) {
        const float *q = xyz1 + ((size_t)b * N + n) * 3;
        qx = q[0]; qy = q[1]; qz = q[2];
    }
    const float sq = (qx * qx + qy * qy) + qz * qz;
    float d0 = INFINITY, d1 = INFINITY, d2 = INFINITY;
    int i0 = -1, i1 = -1, i2 = -1;
    for (int s0 = 0; s0 < S; s0 += TNN_TILE) {
        const int cnt = min(TNN_TILE, S - s0);
        __syncthreads();
        for (int e = threadIdx.x; e < cnt; e += 256) {
            const float *t = xyz2 + ((size_t)b * S + s0 + e) * 3;
            const float x = t[0], y = t[1], z = t[2];
            T[e] = make_float4(x, y, z, (x * x + y * y) + z * z);
        }
        __syncthreads();
        for (int e = 0; e < cnt; ++e) {             // wave-uniform address: LDS broadcast
            const float4 t = T[e];
            const float mm = fmaf(qz, t.z, fmaf(qy, t.y, qx * t.x));
            const float d = ((-2.0f * mm) + sq) + t.w;
            if (d < d2) {                           // strict: an equal later index never displaces
                const int j = s0 + e;
                if (d < d1) {
                    d2 = d1; i2 = i1;
                    if (d < d0) { d1 = d0; i1 = i0; d0 = d; i0 = j; }
                    else { d1 = d; i1 = j; }
                } else { d2 = d; i2 = j; }
            }
        }
    }
    if (live) {
        const size_t o = ((size_t)b * N + n) * 3;
        dist3[o] = d0; dist3[o + 1] = d1; dist3[o + 2] = d2;
        idx3[o] = i0; idx3[o + 1] = i1; idx3[o + 2] = i2;
    }
}

__global__ __launch_bounds__(256) void interp3_kernel(const float *__restrict__ dist,
                                                      const int64_t *__restrict__ idx,
                                                      const float *__restrict__ P2, int N, int S2, int D,
                                                      float *__restrict__ out, int ldo, int col0) {
    const int r = blockIdx.x, b = blockIdx.y;      // one workgroup per query point
    const size_t q = (size_t)b * N + r;
    float w[3];
    int id[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        w[k] = 1.0f / (dist[q * 3 + k] + 1e-8f);
        id[k] = (int)idx[q * 3 + k];
    }
    const float ws = (w[0] + w[1]) + w[2];
#pragma unroll
    for (int k = 0; k < 3; ++k) w[k] = w[k] / ws;
    const float *p0 = P2 + ((size_t)b * S2 + id[0]) * D;
    const float *p1 = P2 + ((size_t)b * S2 + id[1]) * D;
    const float *p2 = P2 + ((size_t)b * S2 + id[2]) * D;
    for (int c = threadIdx.x; c < D; c += 256)
        out[q * ldo + col0 + c] = (p0[c] * w[0] + p1[c] * w[1]) + p2[c] * w[2];
}

extern "C" int reart_three_nn(const float *xyz1, const float *xyz2, int B, int N, int S2, float *dist3,
                              int64_t *idx3, void *stream) {
    if (B < 0 || N < 0 || S2 < 3) return REART_ERR_INVALID_ARG;
    if (B == 0 || N == 0) return REART_OK;
    if (!xyz1 || !xyz2 || !dist3 || !idx3) return REART_ERR_INVALID_ARG;
    hipLaunchKernelGGL(three_nn_expanded_kernel, dim3(reart_div_up(N, 256), B), dim3(256), 0, (hipStream_t)stream,
                       xyz1, xyz2, N, S2, dist3, idx3);
    REART_CHECK_LAUNCH();
    return REART_OK;
}

extern "C" size_t reart_three_interpolate_workspace_bytes(int B, int N, int S2) {
    if (B <= 0 || N <= 0 || S2 <= 0) return 0;
    return reart_align_up(sizeof(float) * (size_t)B * N * 3, 256) + reart_align_up(sizeof(int64_t) * (size_t)B * N * 3, 256);
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
    int64_t *idx = (int64_t *)(ws + reart_align_up(sizeof(float) * (size_t)B * N * 3, 256));
    hipStream_t st = (hipStream_t)stream;
    const int rc = reart_three_nn(xyz1, xyz2, B, N, S2, dist, idx, stream);
    if (rc != REART_OK) return rc;
    hipLaunchKernelGGL(interp3_kernel, dim3(N, B), dim3(256), 0, st, dist, idx, points2, N, S2, D, out, ldo, col0);
    REART_CHECK_LAUNCH();
    return REART_OK;
}
