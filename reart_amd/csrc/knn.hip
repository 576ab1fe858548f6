// reart_amd/csrc/knn.hip -- brute-force K-nearest-neighbour / Chamfer kernels for gfx950.
//
// Replaces chamferdist._C.knn_points_idx / knn_points_backward (reference
// utils/chamfer.py:174,206-208) and knn_cuda.KNN.forward (run_robot.py:65-66).
//
// Design (DESIGN.md section "K-NN"):
//   * The kernel is fp32-VALU bound (8 flop per pair against 24 B per point), so the
//     layout is chosen for VALU issue, not for HBM: targets are transposed once into a
//     +INF padded SoA image [n][3][Ppad]; every wave reads them with SCALAR loads
//     (s_load_dwordx16: the target index is wave-uniform) and feeds SGPR pairs straight
//     into packed-fp32 VALU ops (v_pk_add_f32 / v_pk_mul_f32), two targets per
//     instruction.  No LDS, no VGPRs spent on targets.
//   * One query per lane, one wave per workgroup.  The target range is cut into S
//     slices so that a launch has >> 1024 (= 256 CUs x 4 SIMDs) wave-sized work items;
//     slice results are merged by a tiny second kernel in slice order, which keeps the
//     "ties -> lowest index" rule without atomics.
//   * K = 1 keeps only a running minimum and the id of the 16-target block that
//     produced it (1 min per pair instead of compare + 2 selects); the exact index is
//     recovered afterwards by rescanning that block for equality.
//   * Rounding contract: d = ((dx*dx)+(dy*dy))+(dz*dz), fp32, no FMA (file compiled with
//     -ffp-contract=off), strict '<' while scanning ascending j.
#include "common.h"
#include "internal.h"
#include "blocksort.h"
#include <math.h>




// ---------------------------------------------------------------------------------
// AoS [n][P][3]  ->  SoA [n][3][Ppad], entries j >= length padded with +INF
// ---------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void soa_kernel(SoaArgs a) {
    const SoaJob jb = a.job[blockIdx.z];
    const int b = blockIdx.y;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= jb.Ppad) return;
    int n = jb.len ? (int)jb.len[b] : jb.P;
    n = n < jb.P ? n : jb.P;
    float x = INFINITY, y = INFINITY, z = INFINITY;
    if (j < n) {
        const float *p = jb.src + ((size_t)b * jb.P + j) * 3;
        x = p[0]; y = p[1]; z = p[2];
    }
    float *o = jb.dst + (size_t)b * 3 * jb.Ppad;
    o[j] = x;
    o[jb.Ppad + j] = y;
    o[2 * (size_t)jb.Ppad + j] = z;
}

// ---------------------------------------------------------------------------------
// Slice kernel
// ---------------------------------------------------------------------------------

// Block-skip test.  box = AABB (lo xyz, hi xyz) of 64 consecutive targets, read through the scalar
// path (wave-uniform).  lb = ((ex*ex)+(ey*ey))+(ez*ez) with e = max(lo - q, q - hi, 0) per axis uses
// the SAME operations as the distance itself, and fp32 rounding is monotone, so lb <= d for every
// target inside the box: if lb > thr for all 64 lanes no target of the box can win or tie.
// (Pays off when a wave's queries are spatially coherent, i.e. clouds stored in Morton order.)
__device__ __forceinline__ bool box_skip(const float *box, float qx, float qy, float qz, float thr) {
    const float ex = fmaxf(fmaxf(box[0] - qx, qx - box[3]), 0.f);
    const float ey = fmaxf(fmaxf(box[1] - qy, qy - box[4]), 0.f);
    const float ez = fmaxf(fmaxf(box[2] - qz, qz - box[5]), 0.f);
    const float lb = (ex * ex + ey * ey) + ez * ez;
    return !__any(lb <= thr);
}

// AABB of every 64 consecutive entries of a (+INF padded) SoA image [N][3][Ppad]; an all-padding
// box is (+INF, +INF) and is always skipped.  One wave per box.
__global__ __launch_bounds__(256) void box_kernel(const float *__restrict__ soa, int Ppad,
                                                  float *__restrict__ boxes) {
    const int lane = threadIdx.x & 63;
    const int e0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 64, b = blockIdx.y;   // 64 entries per wave
    if (e0 >= Ppad) return;
    const int nbox = Ppad / NN_BOX;
    const float *p = soa + (size_t)b * 3 * Ppad + e0 + lane;
    const bool in = e0 + lane < Ppad;
    float lo[3], hi[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float v = in ? p[(size_t)c * Ppad] : INFINITY;
        lo[c] = v;
        hi[c] = (v == INFINITY) ? -INFINITY : v;
#pragma unroll
        for (int o = NN_BOX / 2; o >= 1; o >>= 1) {   // reduce within groups of NN_BOX lanes
            lo[c] = fminf(lo[c], __shfl_xor(lo[c], o, 64));
            hi[c] = fmaxf(hi[c], __shfl_xor(hi[c], o, 64));
        }
        if (hi[c] == -INFINITY) hi[c] = INFINITY;
    }
    if ((lane & (NN_BOX - 1)) == 0 && in) {
        float *o = boxes + ((size_t)b * nbox + (e0 + lane) / NN_BOX) * 8;
        o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = hi[0]; o[4] = hi[1]; o[5] = hi[2]; o[6] = 0.f; o[7] = 0.f;
    }
}

int reart_boxes_launch(const float *soa, int N, int Ppad, float *boxes, hipStream_t st) {
    hipLaunchKernelGGL(box_kernel, dim3(reart_div_up(reart_div_up(Ppad, 64), 4), N), dim3(256), 0, st, soa, Ppad, boxes);
    REART_CHECK_LAUNCH();
    return REART_OK;
}

// insertion under the precondition d < kd[KK-1] (checked by the caller)
template <int KK>
__device__ __forceinline__ void knn_insert(float (&kd)[KK], int (&ki)[KK], float d, int j) {
    if (KK == 3) {  // 2 compares + 8 selects instead of the generic 6 + 12
        const bool c1 = d < kd[1], c0 = d < kd[0];
        kd[2] = c1 ? kd[1] : d;
        ki[2] = c1 ? ki[1] : j;
        kd[1] = c0 ? kd[0] : (c1 ? d : kd[1]);
        ki[1] = c0 ? ki[0] : (c1 ? j : ki[1]);
        kd[0] = c0 ? d : kd[0];
        ki[0] = c0 ? j : ki[0];
        return;
    }
#pragma unroll
    for (int s = KK - 1; s >= 0; --s) {
        const bool lt_prev = (s > 0) && (d < kd[s > 0 ? s - 1 : 0]);
        const bool lt_cur = d < kd[s];
        kd[s] = lt_prev ? kd[s > 0 ? s - 1 : 0] : (lt_cur ? d : kd[s]);
        ki[s] = lt_prev ? ki[s > 0 ? s - 1 : 0] : (lt_cur ? j : ki[s]);
    }
}

template <int KK>
__device__ __forceinline__ void knn_write_final(const KnnJob &jb, int b, int i, int K, int euclidean,
                                                const float (&kd)[KK], const int (&ki)[KK]) {
    int n1 = jb.lenq ? (int)jb.lenq[b] : jb.P1;
    int n2 = jb.lent ? (int)jb.lent[b] : jb.P2;
    n2 = n2 < jb.P2 ? n2 : jb.P2;
    const int valid = (i < n1) ? (K < n2 ? K : n2) : 0;
    float *od = jb.dists + ((size_t)b * jb.P1 + i) * K;
    int64_t *oi = jb.idx + ((size_t)b * jb.P1 + i) * K;
#pragma unroll
    for (int k = 0; k < KK; ++k) {
        if (k < K) {
            const bool ok = k < valid;
            float d = kd[k];
            if (euclidean) d = sqrtf(d);
            od[k] = ok ? d : 0.0f;
            oi[k] = ok ? (int64_t)ki[k] : (int64_t)0;
        }
    }
}

template <int KK, bool FINAL, bool BLK = false>
__global__ __launch_bounds__(NN_BS) void knn_slice_kernel(KnnArgs a) {
    const int w = reart_xcd_remap(blockIdx.x, a.items);
    if (w < 0) return;
    const int jsel = (w >= a.items0) ? 1 : 0;
    const KnnJob jb = a.job[jsel];
    const int wl = w - (jsel ? a.items0 : 0);
    const int g = wl % jb.nqg;
    const int s = (wl / jb.nqg) % a.S;
    const int b = wl / (jb.nqg * a.S);

    const int i = g * NN_BS + threadIdx.x;
    const int ic = i < jb.P1 ? i : jb.P1 - 1;  // clamp: tail lanes redo the last query
    // optional per-batch query remap (fused step: pair f queries frame qmap[f], <0 = q_alt)
    const int qb = jb.qmap ? jb.qmap[b] : b;
    const float *qp = (qb < 0 ? jb.q_alt : jb.q + (size_t)qb * jb.P1 * 3) + (size_t)ic * 3;
    const float qx = qp[0], qy = qp[1], qz = qp[2];
    const f2 qx2 = {qx, qx}, qy2 = {qy, qy}, qz2 = {qz, qz};

    const float *tx = jb.tsoa + (size_t)b * 3 * jb.Ppad;
    const float *ty = tx + jb.Ppad;
    const float *tz = ty + jb.Ppad;
    // ragged targets: per-batch slice length from the batch's own target count
    const int algn = jb.boxes ? NN_BOX : NN_UB;
    const int Lb = jb.tlen ? (((jb.tlen[b] + a.S - 1) / a.S + algn - 1) / algn) * algn : jb.L;
    const float *bx = jb.boxes ? jb.boxes + (size_t)b * (jb.Ppad / NN_BOX) * 8 : nullptr;
    const int j0 = s * Lb, j1 = j0 + Lb;

    float kd[KK];
    int ki[KK];
#pragma unroll
    for (int k = 0; k < KK; ++k) { kd[k] = INFINITY; ki[k] = 0; }

    if (KK == 1) {
        int blk = j0;
        for (int jo = j0; jo < j1; jo += NN_BOX) {
            if (bx && box_skip(bx + (jo / NN_BOX) * 8, qx, qy, qz, kd[0])) continue;
            const int je = (jo + NN_BOX < j1) ? jo + NN_BOX : j1;
        for (int j = jo; j < je; j += NN_UB) {
            float m = INFINITY;
#pragma unroll
            for (int u = 0; u < NN_UB; u += 2) {
                const f2 dx = qx2 - *(const f2 *)(tx + j + u);
                const f2 dy = qy2 - *(const f2 *)(ty + j + u);
                const f2 dz = qz2 - *(const f2 *)(tz + j + u);
                const f2 d = (dx * dx + dy * dy) + dz * dz;
                m = fminf(fminf(m, d.x), d.y);
            }
            if (m < kd[0]) { kd[0] = m; blk = j; }
        }
        }
        // recover the exact (lowest) index inside the winning block
        int bi = blk;
#pragma unroll
        for (int u = NN_UB - 1; u >= 0; --u) {
            const float d = reart_sqdist3(qx, qy, qz, tx[blk + u], ty[blk + u], tz[blk + u]);
            if (d == kd[0]) bi = blk + u;
        }
        ki[0] = bi;
    } else {
        // K > 1.  Phase A keeps the KK best BLOCKS of 8 targets by (block minimum, block id):
        // one min per pair and one branch-free insertion per block, instead of a divergent
        // insertion test per target.  The KK nearest targets lie inside those blocks (any
        // other block's minimum key exceeds the KK-th smallest target key), so phase B rescans
        // only KK*8 targets per query with an exact (d, index) insertion.
        constexpr int UBK = 8;
        float bm[KK];
        int bb[KK];
#pragma unroll
        for (int k = 0; k < KK; ++k) { bm[k] = INFINITY; bb[k] = -1; }
        for (int jo = j0; jo < j1; jo += NN_BOX) {
            if (bx && box_skip(bx + (jo / NN_BOX) * 8, qx, qy, qz, bm[KK - 1])) continue;
            const int je = (jo + NN_BOX < j1) ? jo + NN_BOX : j1;
        for (int j = jo; j < je; j += UBK) {
            float m = INFINITY;
#pragma unroll
            for (int u = 0; u < UBK; u += 2) {
                const f2 dx = qx2 - *(const f2 *)(tx + j + u);
                const f2 dy = qy2 - *(const f2 *)(ty + j + u);
                const f2 dz = qz2 - *(const f2 *)(tz + j + u);
                const f2 d = (dx * dx + dy * dy) + dz * dz;
                m = fminf(fminf(m, d.x), d.y);
            }
            // strict '<': on equal minima the earlier (lower-index) block stays ahead
#pragma unroll
            for (int s = KK - 1; s >= 0; --s) {
                const bool lt_prev = (s > 0) && (m < bm[s > 0 ? s - 1 : 0]);
                const bool lt_cur = m < bm[s];
                bm[s] = lt_prev ? bm[s > 0 ? s - 1 : 0] : (lt_cur ? m : bm[s]);
                bb[s] = lt_prev ? bb[s > 0 ? s - 1 : 0] : (lt_cur ? j : bb[s]);
            }
        }
        }
        if (BLK) {  // fused step: the consumer merges the slices' blocks and rescans once per query
#pragma unroll
            for (int k = 0; k < KK; ++k) { kd[k] = bm[k]; ki[k] = bb[k]; }
        } else
#pragma unroll
        for (int c = 0; c < KK; ++c) {
            const int blk = bb[c];
            if (blk < 0) continue;
#pragma unroll
            for (int u = 0; u < UBK; ++u) {
                const float d = reart_sqdist3(qx, qy, qz, tx[blk + u], ty[blk + u], tz[blk + u]);
                const int jj = blk + u;
                // candidates arrive in block-rank order, not index order: compare the full key
                if (d < kd[KK - 1] || (d == kd[KK - 1] && d < INFINITY && jj < ki[KK - 1])) {
#pragma unroll
                    for (int s = KK - 1; s >= 0; --s) {
                        const int sp = s > 0 ? s - 1 : 0;
                        const bool lt_prev = (s > 0) && (d < kd[sp] || (d == kd[sp] && jj < ki[sp]));
                        const bool lt_cur = d < kd[s] || (d == kd[s] && jj < ki[s]);
                        kd[s] = lt_prev ? kd[sp] : (lt_cur ? d : kd[s]);
                        ki[s] = lt_prev ? ki[sp] : (lt_cur ? jj : ki[s]);
                    }
                }
            }
        }
    }

    if (i >= jb.P1) return;
    if (FINAL) {
        knn_write_final<KK>(jb, b, i, a.K, a.euclidean, kd, ki);
    } else {
        const size_t o = (((size_t)s * a.N + b) * jb.P1 + i) * KK;
#pragma unroll
        for (int k = 0; k < KK; ++k) { jb.pd[o + k] = kd[k]; jb.pi[o + k] = ki[k]; }
    }
}

// merge the S partial lists of each query in slice order (ascending index ranges)
template <int KK>
__global__ __launch_bounds__(256) void knn_merge_kernel(KnnArgs a) {
    const KnnJob jb = a.job[blockIdx.z];
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= jb.P1) return;
    float kd[KK];
    int ki[KK];
#pragma unroll
    for (int k = 0; k < KK; ++k) { kd[k] = INFINITY; ki[k] = 0; }
    for (int s = 0; s < a.S; ++s) {
        const size_t o = (((size_t)s * a.N + b) * jb.P1 + i) * KK;
#pragma unroll
        for (int k = 0; k < KK; ++k) {
            const float d = jb.pd[o + k];
            const int j = jb.pi[o + k];
            if (d < kd[KK - 1]) knn_insert<KK>(kd, ki, d, j);
        }
    }
    knn_write_final<KK>(jb, b, i, a.K, a.euclidean, kd, ki);
}

// ---------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------
static int knn_round_k(int K) {
    const int opts[] = {1, 2, 3, 4, 8, 16};
    for (int o : opts)
        if (K <= o) return o;
    return -1;
}

// number of target slices: enough wave-sized work items to balance 1024 SIMDs
// K > 1 restarts its sorted list in every slice (each restart re-runs the insertion path for
// the first ~64*K*ln targets), so it wants fewer, longer slices than K = 1.
static int knn_pick_split(long waves, int P2, int K) {
    int S = 1;
    const long want = 8192;  // measured: K = 3 also prefers many short slices (S=8: 82 us, S=4: 90, S=2: 106)
    while (waves * S < want && S < 16) S *= 2;
    while (S > 1 && reart_div_up(P2, S) < 128) S /= 2;  // keep slices meaningful
    return S < 1 ? 1 : S;
}

int reart_knn_pick_split(long waves, int P2, int K) { return knn_pick_split(waves, P2, K); }

struct KnnPlan {
    int KK, S;
    int L[2], Ppad[2];
    size_t off_soa[2], off_pd[2], off_pi[2], total;
};

// job j: queries P1[j] against targets P2[j]
static int knn_plan(int njobs, int N, const int *P1, const int *P2, int K, KnnPlan *pl) {
    pl->KK = knn_round_k(K);
    if (pl->KK < 0) return REART_ERR_UNSUPPORTED;
    long waves = 0;
    int minP2 = 1 << 30;
    for (int j = 0; j < njobs; ++j) {
        waves += (long)N * reart_div_up(P1[j], NN_BS);
        minP2 = P2[j] < minP2 ? P2[j] : minP2;
    }
    pl->S = knn_pick_split(waves, minP2, K);
    size_t off = 0;
    for (int j = 0; j < njobs; ++j) {
        pl->L[j] = (int)reart_align_up((size_t)reart_div_up(P2[j], pl->S), NN_UB);
        pl->Ppad[j] = pl->L[j] * pl->S;
        pl->off_soa[j] = off;
        off += reart_align_up((size_t)N * 3 * pl->Ppad[j] * sizeof(float), 256);
    }
    for (int j = 0; j < njobs; ++j) {
        pl->off_pd[j] = pl->off_pi[j] = 0;
        if (pl->S > 1) {
            const size_t n = (size_t)pl->S * N * P1[j] * pl->KK;
            pl->off_pd[j] = off;
            off += reart_align_up(n * sizeof(float), 256);
            pl->off_pi[j] = off;
            off += reart_align_up(n * sizeof(int), 256);
        }
    }
    pl->total = off;
    return REART_OK;
}

// partial-only launch used by the fused step (merge happens in its consumer kernels)
int reart_knn_launch_slices(const KnnArgs &a, int KK, hipStream_t st) {
    const int grid = reart_xcd_grid(a.items);
    switch (KK) {
        case 1: hipLaunchKernelGGL((knn_slice_kernel<1, false>), dim3(grid), dim3(NN_BS), 0, st, a); break;
        case 3: hipLaunchKernelGGL((knn_slice_kernel<3, false, true>), dim3(grid), dim3(NN_BS), 0, st, a); break;
        default: return REART_ERR_UNSUPPORTED;
    }
    REART_CHECK_LAUNCH();
    return REART_OK;
}
int reart_soa_launch(const SoaArgs &sa, int maxPpad, int N, int njobs, hipStream_t st) {
    hipLaunchKernelGGL(soa_kernel, dim3(reart_div_up(maxPpad, 256), N, njobs), dim3(256), 0, st, sa);
    REART_CHECK_LAUNCH();
    return REART_OK;
}

template <int KK>
static void knn_launch(const KnnArgs &a, int njobs, int maxP1, hipStream_t st) {
    const int grid = reart_xcd_grid(a.items);
    if (a.S == 1) {
        hipLaunchKernelGGL((knn_slice_kernel<KK, true>), dim3(grid), dim3(NN_BS), 0, st, a);
    } else {
        hipLaunchKernelGGL((knn_slice_kernel<KK, false>), dim3(grid), dim3(NN_BS), 0, st, a);
        hipLaunchKernelGGL((knn_merge_kernel<KK>), dim3(reart_div_up(maxP1, 256), a.N, njobs),
                           dim3(256), 0, st, a);
    }
}

// Generic driver: njobs in {1,2}
int reart_knn_run(int njobs, const float *const *q, const float *const *t,
                   const int64_t *const *lenq, const int64_t *const *lent, int N, const int *P1,
                   const int *P2, int K, int euclidean, float *const *dists, int64_t *const *idx,
                   void *workspace, size_t workspace_bytes, hipStream_t st) {
    KnnPlan pl;
    int rc = knn_plan(njobs, N, P1, P2, K, &pl);
    if (rc != REART_OK) return rc;
    if (pl.total > workspace_bytes || (pl.total && !workspace)) return REART_ERR_INVALID_ARG;
    char *ws = (char *)workspace;

    SoaArgs sa;
    KnnArgs a;
    int maxPpad = 0, maxP1 = 0;
    a.N = N; a.S = pl.S; a.K = K; a.euclidean = euclidean;
    a.items0 = 0; a.items = 0;
    for (int j = 0; j < 2; ++j) {
        const int jj = j < njobs ? j : 0;
        sa.job[j].src = t[jj];
        sa.job[j].len = lent ? lent[jj] : nullptr;
        sa.job[j].dst = (float *)(ws + pl.off_soa[jj]);
        sa.job[j].P = P2[jj];
        sa.job[j].Ppad = pl.Ppad[jj];
        KnnJob &kj = a.job[j];
        kj.q = q[jj];
        kj.q_alt = nullptr; kj.qmap = nullptr; kj.tlen = nullptr; kj.boxes = nullptr;
        kj.tsoa = (const float *)(ws + pl.off_soa[jj]);
        kj.lenq = lenq ? lenq[jj] : nullptr;
        kj.lent = lent ? lent[jj] : nullptr;
        kj.P1 = P1[jj]; kj.P2 = P2[jj]; kj.Ppad = pl.Ppad[jj]; kj.L = pl.L[jj];
        kj.nqg = reart_div_up(P1[jj], NN_BS);
        kj.pd = pl.S > 1 ? (float *)(ws + pl.off_pd[jj]) : nullptr;
        kj.pi = pl.S > 1 ? (int *)(ws + pl.off_pi[jj]) : nullptr;
        kj.dists = dists[jj];
        kj.idx = idx[jj];
        if (j < njobs) {
            const int items = N * kj.nqg * pl.S;
            if (j == 0) a.items0 = items;
            a.items += items;
            maxPpad = pl.Ppad[j] > maxPpad ? pl.Ppad[j] : maxPpad;
            maxP1 = P1[j] > maxP1 ? P1[j] : maxP1;
        }
    }
    hipLaunchKernelGGL(soa_kernel, dim3(reart_div_up(maxPpad, 256), N, njobs), dim3(256), 0, st, sa);
    switch (pl.KK) {
        case 1: knn_launch<1>(a, njobs, maxP1, st); break;
        case 2: knn_launch<2>(a, njobs, maxP1, st); break;
        case 3: knn_launch<3>(a, njobs, maxP1, st); break;
        case 4: knn_launch<4>(a, njobs, maxP1, st); break;
        case 8: knn_launch<8>(a, njobs, maxP1, st); break;
        case 16: knn_launch<16>(a, njobs, maxP1, st); break;
        default: return REART_ERR_UNSUPPORTED;
    }
    REART_CHECK_LAUNCH();
    return REART_OK;
}

extern "C" size_t reart_knn_points_workspace_bytes(int N, int P1, int P2, int K) {
    KnnPlan pl;
    if (N <= 0 || P1 <= 0 || P2 <= 0 || knn_plan(1, N, &P1, &P2, K, &pl) != REART_OK) return 0;
    return pl.total;
}

extern "C" size_t reart_chamfer_bidir_workspace_bytes(int N, int P) {
    KnnPlan pl;
    const int Ps[2] = {P, P};
    if (N <= 0 || P <= 0 || knn_plan(2, N, Ps, Ps, 1, &pl) != REART_OK) return 0;
    return pl.total;
}

extern "C" int reart_knn_points_idx(const float *p1, const float *p2, const int64_t *lengths1,
                                    const int64_t *lengths2, int N, int P1, int P2, int D, int K,
                                    float *dists, int64_t *idx, void *workspace,
                                    size_t workspace_bytes, void *stream) {
    if (N < 0 || P1 < 0 || P2 < 0 || K < 1) return REART_ERR_INVALID_ARG;
    if (D != 3 || K > REART_MAX_K) return REART_ERR_UNSUPPORTED;
    if (N == 0 || P1 == 0) return REART_OK;
    if (!dists || !idx) return REART_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (P2 == 0) {  // nothing to search: zero padded outputs (utils/chamfer.py:163-170)
        if (hipMemsetAsync(dists, 0, sizeof(float) * (size_t)N * P1 * K, st) != hipSuccess ||
            hipMemsetAsync(idx, 0, sizeof(int64_t) * (size_t)N * P1 * K, st) != hipSuccess)
            return REART_ERR_LAUNCH;
        return REART_OK;
    }
    if (!p1 || !p2) return REART_ERR_INVALID_ARG;
    const int64_t *lq[1] = {lengths1}, *lt[1] = {lengths2};
    return reart_knn_run(1, &p1, &p2, lq, lt, N, &P1, &P2, K, 0, &dists, &idx, workspace, workspace_bytes, st);
}

extern "C" int reart_chamfer_bidir(const float *x, const float *y, int N, int P, float *d_xy,
                                   int64_t *i_xy, float *d_yx, int64_t *i_yx, void *workspace,
                                   size_t workspace_bytes, void *stream) {
    if (N < 0 || P < 0) return REART_ERR_INVALID_ARG;
    if (N == 0 || P == 0) return REART_OK;
    if (!x || !y || !d_xy || !i_xy || !d_yx || !i_yx) return REART_ERR_INVALID_ARG;
    const float *q[2] = {x, y}, *t[2] = {y, x};
    const int Ps[2] = {P, P};
    float *dd[2] = {d_xy, d_yx};
    int64_t *ii[2] = {i_xy, i_yx};
    return reart_knn_run(2, q, t, nullptr, nullptr, N, Ps, Ps, 1, 0, dd, ii, workspace, workspace_bytes,
                   (hipStream_t)stream);
}

extern "C" int reart_knn_cuda(const float *ref, const float *query, int B, int nr, int nq, int D,
                              int k, int euclidean, float *dist, int64_t *idx, void *workspace,
                              size_t workspace_bytes, void *stream) {
    if (B < 0 || nr < 0 || nq < 0 || k < 1) return REART_ERR_INVALID_ARG;
    if (D != 3 || k > REART_MAX_K) return REART_ERR_UNSUPPORTED;
    if (k > nr) return REART_ERR_INVALID_ARG;  // knn_cuda asserts k <= number of references
    if (B == 0 || nq == 0) return REART_OK;
    if (!ref || !query || !dist || !idx) return REART_ERR_INVALID_ARG;
    return reart_knn_run(1, &query, &ref, nullptr, nullptr, B, &nq, &nr, k, euclidean ? 1 : 0, &dist, &idx,
                   workspace, workspace_bytes, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------
// Backward of knn_points (utils/chamfer.py:195-209), deterministic.
//   grad_p1[n,i] = sum_k 2 g[n,i,k] (p1[n,i] - p2[n,idx])
//   grad_p2[n,j] = - sum_{(i,k): idx[n,i,k]==j} 2 g[n,i,k] (p1[n,i] - p2[n,j])
// One workgroup per batch element does a counting sort of the (i,k) pairs by target
// index in the caller's workspace, sorts every bucket ascending and accumulates in that
// order -- the same order the CPU loop uses, with no floating-point atomics.
// ---------------------------------------------------------------------------------
#define BWD_BS RS_BS

__global__ __launch_bounds__(BWD_BS) void knn_bwd_kernel(
    const float *__restrict__ p1, const float *__restrict__ p2, const int64_t *__restrict__ len1,
    const int64_t *__restrict__ len2, const int64_t *__restrict__ idx,
    const float *__restrict__ gd, int P1, int P2, int K, int nbits, float *__restrict__ g1,
    float *__restrict__ g2, int *__restrict__ ws) {
    __shared__ int s_cnt[RS_DIG * RS_BS];
    __shared__ int s_wave[RS_BS / 64];
    const int n = blockIdx.x, tid = threadIdx.x;
    int n1 = len1 ? (int)len1[n] : P1;
    int n2 = len2 ? (int)len2[n] : P2;
    n1 = n1 < P1 ? n1 : P1;
    n2 = n2 < P2 ? n2 : P2;
    const int kk = K < n2 ? K : n2;
    p1 += (size_t)n * P1 * 3; p2 += (size_t)n * P2 * 3;
    idx += (size_t)n * P1 * K; gd += (size_t)n * P1 * K;
    g1 += (size_t)n * P1 * 3; g2 += (size_t)n * P2 * 3;
    int *cnt = ws + (size_t)n * (2 * (size_t)P2 + 2 * (size_t)P1 * K);  // [P2]
    int *off = cnt + P2;                                                 // [P2]
    int *bufA = off + P2, *bufB = bufA + (size_t)P1 * K;                 // [P1*K] each

    for (int j = tid; j < P2; j += BWD_BS) cnt[j] = 0;
    __syncthreads();
    // grad_p1 and bucket counts (integer atomics: order-independent result)
    for (int i = tid; i < P1; i += BWD_BS) {
        float ax = 0.f, ay = 0.f, az = 0.f;
        if (i < n1) {
            const float x = p1[3 * i], y = p1[3 * i + 1], z = p1[3 * i + 2];
            for (int k = 0; k < kk; ++k) {
                const int j = (int)idx[(size_t)i * K + k];
                const float c = 2.0f * gd[(size_t)i * K + k];
                ax += c * (x - p2[3 * j]);
                ay += c * (y - p2[3 * j + 1]);
                az += c * (z - p2[3 * j + 2]);
                atomicAdd(&cnt[j], 1);
            }
        }
        g1[3 * i] = ax; g1[3 * i + 1] = ay; g1[3 * i + 2] = az;
    }
    __syncthreads();
    // exclusive scan of cnt -> off
    const int chunk = (P2 + BWD_BS - 1) / BWD_BS;
    const int c0 = tid * chunk < P2 ? tid * chunk : P2, c1 = (c0 + chunk < P2) ? c0 + chunk : P2;
    int tot = 0;
    for (int j = c0; j < c1; ++j) tot += cnt[j];
    int run = block_excl_scan(tot, s_wave, nullptr);
    for (int j = c0; j < c1; ++j) { off[j] = run; run += cnt[j]; }
    // valid (i,k) pairs, id e = i*kk + k, stably sorted by target index
    const int M = n1 * kk;
    const int *sorted = block_stable_sort_ids(M, nbits, bufA, bufB, s_cnt, s_wave, [&](int e) {
        return (int)idx[(size_t)(e / kk) * K + (e % kk)];
    });
    for (int j = tid; j < P2; j += BWD_BS) {
        const int o = off[j], c = cnt[j];
        float ax = 0.f, ay = 0.f, az = 0.f;
        const float x = p2[3 * j], y = p2[3 * j + 1], z = p2[3 * j + 2];
        for (int a = 0; a < c; ++a) {
            const int e = sorted[o + a];
            const int i = e / kk, k = e % kk;
            const float cf = 2.0f * gd[(size_t)i * K + k];
            ax -= cf * (p1[3 * i] - x);
            ay -= cf * (p1[3 * i + 1] - y);
            az -= cf * (p1[3 * i + 2] - z);
        }
        g2[3 * j] = ax; g2[3 * j + 1] = ay; g2[3 * j + 2] = az;
    }
}

extern "C" size_t reart_knn_points_backward_workspace_bytes(int N, int P1, int P2, int K) {
    if (N <= 0 || P1 < 0 || P2 < 0 || K < 1) return 0;
    return sizeof(int) * (size_t)N * (2 * (size_t)P2 + 2 * (size_t)P1 * K);
}

extern "C" int reart_knn_points_backward(const float *p1, const float *p2, const int64_t *lengths1,
                                         const int64_t *lengths2, const int64_t *idx,
                                         const float *grad_dists, int N, int P1, int P2, int D,
                                         int K, float *grad_p1, float *grad_p2, void *workspace,
                                         size_t workspace_bytes, void *stream) {
    if (N < 0 || P1 < 0 || P2 < 0 || K < 1) return REART_ERR_INVALID_ARG;
    if (D != 3) return REART_ERR_UNSUPPORTED;
    if (N == 0) return REART_OK;
    hipStream_t st = (hipStream_t)stream;
    if (P1 == 0 || P2 == 0) {
        if (P1 && hipMemsetAsync(grad_p1, 0, sizeof(float) * (size_t)N * P1 * 3, st) != hipSuccess)
            return REART_ERR_LAUNCH;
        if (P2 && hipMemsetAsync(grad_p2, 0, sizeof(float) * (size_t)N * P2 * 3, st) != hipSuccess)
            return REART_ERR_LAUNCH;
        return REART_OK;
    }
    if (!p1 || !p2 || !idx || !grad_dists || !grad_p1 || !grad_p2 || !workspace)
        return REART_ERR_INVALID_ARG;
    if (workspace_bytes < reart_knn_points_backward_workspace_bytes(N, P1, P2, K))
        return REART_ERR_INVALID_ARG;
    hipLaunchKernelGGL(knn_bwd_kernel, dim3(N), dim3(BWD_BS), 0, st, p1, p2, lengths1, lengths2, idx,
                       grad_dists, P1, P2, K, reart_bits_for(P2), grad_p1, grad_p2, (int *)workspace);
    REART_CHECK_LAUNCH();
    return REART_OK;
}
