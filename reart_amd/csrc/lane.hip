// reart_amd/csrc/lane.hip -- EXACT nearest-neighbour search, warm-started and box-pruned PER QUERY.
//
// Same contract and same answers as knn.hip / prune.hip (distance ((dx*dx)+(dy*dy))+(dz*dz) in fp32,
// strict '<', ties -> lowest index; reference utils/chamfer.py:78-94, utils/flow_utils.py:158), for the
// repeated searches of the relaxation loop (run_robot.py:154-221).
//
// prune.hip drops target boxes per WAVE: a box of 16 targets is scanned by all 64 lanes when any lane
// needs it, and on the Morton-ordered clouds of the loop only ~10 % of the lanes do.  Here every lane
// keeps its OWN candidate list, so a scan step does 64 useful (query, box) pairs:
//
//   workgroup = 4 waves = 256 queries of one cloud pair; the target cloud (SoA, <= 4096 points per
//   chunk) and its 16-target boxes are staged in LDS once per workgroup, every later gather is an LDS
//   read (the per-lane box walks are chains of dependent gathers: ~100 cycles from LDS, ~2000 from L2).
//   Per wave:
//     1. warm start: thr = max_k d(q, t[seed_k]) from the previous iteration's neighbours;
//     2. 64-target super boxes (lane l builds super box l from its 4 boxes): box-to-box filter against
//        the 4 query-group boxes (one ballot), then point-to-box against each lane's own thr -> a
//        64-bit mask of super boxes PER LANE;
//     3. every lane walks its own mask: 4 point-to-box tests per super box, survivors are appended to
//        the lane's list in LDS;
//     4. every lane scans its own list (16 targets per entry, read from LDS as 12 x 16 bytes).
//
// Exactness: identical argument to prune.hip -- all lower bounds use the operations of the distance
// itself (monotone in fp32), a box is dropped only when lb > thr strictly, thr is an upper bound of the
// K-th neighbour distance, and each lane visits its boxes in ascending index order with strict '<'.
#include "common.h"
#include "internal.h"
#include <math.h>

#define LN_WAVES 4
#define LN_CHUNK 4096                 // targets staged in LDS at a time
#define LN_NBOX (LN_CHUNK / NN_BOX)   // 256 boxes per chunk
#define LN_CAP 16                     // list slots per lane (flushed when nearly full)
// LDS layout against bank conflicts of the per-lane 16-byte gathers: a box of 16 targets occupies 20
// floats per axis (start banks 20 id mod 64: 16 distinct quads, the minimum for 64 x 16 bytes), and the
// box corners are two float4 arrays (lo.xyz, hi.x | hi.y, hi.z) with a 16-byte stride.
#define LN_TS 20
#define LN_AX (LN_NBOX * LN_TS)       // floats per axis
static_assert(NN_BOX == 16 && LN_NBOX == 256, "lane.hip: 64 super boxes of 4 boxes of 16 targets per chunk");

#ifdef REART_PRUNE_STATS
__device__ unsigned long long g_lane_stats[8];
extern "C" int reart_debug_lane_stats(unsigned long long *out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lane_stats), sizeof(g_lane_stats)) != hipSuccess) return REART_ERR_LAUNCH;
    if (reset) { unsigned long long z[8] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_lane_stats), z, sizeof(z)); }
    return REART_OK;
}
#define LANE_STAT(k, v) do { if ((threadIdx.x & 63) == 0) atomicAdd(&g_lane_stats[k], (unsigned long long)(v)); } while (0)
#else
#define LANE_STAT(k, v) do { } while (0)
#endif

__device__ __forceinline__ float ln_rl(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ float ln_lb(float lo0, float lo1, float lo2, float hi0, float hi1, float hi2,
                                       float qlo0, float qlo1, float qlo2, float qhi0, float qhi1, float qhi2) {
    const float ex = fmaxf(fmaxf(lo0 - qhi0, qlo0 - hi0), 0.f);
    const float ey = fmaxf(fmaxf(lo1 - qhi1, qlo1 - hi1), 0.f);
    const float ez = fmaxf(fmaxf(lo2 - qhi2, qlo2 - hi2), 0.f);
    return (ex * ex + ey * ey) + ez * ez;
}
__device__ __forceinline__ float ln_min4(float m, float4 d) { return fminf(fminf(fminf(m, d.x), fminf(d.y, d.z)), d.w); }
__device__ __forceinline__ float4 ln_d4(float qx, float qy, float qz, float4 x, float4 y, float4 z) {
    float4 d;
    d.x = reart_sqdist3(qx, qy, qz, x.x, y.x, z.x);
    d.y = reart_sqdist3(qx, qy, qz, x.y, y.y, z.y);
    d.z = reart_sqdist3(qx, qy, qz, x.z, y.z, z.z);
    d.w = reart_sqdist3(qx, qy, qz, x.w, y.w, z.w);
    return d;
}

// KK = 1: (distance, exact index) per query.  KK = 3: top-3 BLOCKS of 8 targets (block minimum, first
// index of the block), rescanned by the consumer (flow_blend_kernel).  Output layout = the partial
// lists of knn.hip with S = 1.
template <int KK>
__global__ __launch_bounds__(64 * LN_WAVES) void knn_lane_kernel(KnnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lsm[];
    float *s_t = lsm;                                        // [3][LN_NBOX][LN_TS]
    float4 *s_ba = (float4 *)(s_t + 3 * LN_AX);              // [LN_NBOX] lo.x lo.y lo.z hi.x
    float4 *s_bb = s_ba + LN_NBOX;                           // [LN_NBOX] hi.y hi.z - -
    unsigned short *s_list = (unsigned short *)(s_bb + LN_NBOX);        // [LN_WAVES][LN_CAP][64]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // the two jobs alternate so that every XCD gets the same mix (see prune.hip)
    const int w = blockIdx.x;
    const bool two = a.items > a.items0;
    const int jsel = two ? (w & 1) : 0;
    const KnnJob jb = a.job[jsel];
    const int wl = two ? (w >> 1) : w;
    const int wgs = (jb.nqg + LN_WAVES - 1) / LN_WAVES;      // workgroups per cloud
    const int b = wl / wgs, g = (wl - b * wgs) * LN_WAVES + wv;
    const bool wave_live = g < jb.nqg;                       // idle waves still help staging

    const int i = g * NN_BS + lane;
    const int ic = i < jb.P1 ? i : jb.P1 - 1;
    const int qb = jb.qmap ? jb.qmap[b] : b;
    const float *qp = (qb < 0 ? jb.q_alt : jb.q + (size_t)qb * jb.P1 * 3) + (size_t)(ic < 0 ? 0 : ic) * 3;
    const float qx = qp[0], qy = qp[1], qz = qp[2];

    const float *tx = jb.tsoa + (size_t)b * 3 * jb.Ppad;
    const float *ty = tx + jb.Ppad;
    const float *tz = ty + jb.Ppad;
    const int n2 = jb.tlen ? jb.tlen[b] : jb.P2;
    const float *bx = jb.boxes + (size_t)b * (jb.Ppad / NN_BOX) * 8;
    const int nbox_all = jb.Ppad / NN_BOX;

    // ---- warm start
    float thr = 0.f;
    {
        const int *sd = jb.seed + ((size_t)b * jb.P1 + ic) * KK;
        int sj[KK];
        bool ok = true;
#pragma unroll
        for (int k = 0; k < KK; ++k) {
            sj[k] = sd[k];
            ok = ok && sj[k] >= 0 && sj[k] < n2;
#pragma unroll
            for (int k2 = 0; k2 < k; ++k2) ok = ok && sj[k] != sj[k2];
        }
#pragma unroll
        for (int k = 0; k < KK; ++k) {
            const int j = ok ? sj[k] : 0;
            thr = fmaxf(thr, reart_sqdist3(qx, qy, qz, tx[j], ty[j], tz[j]));
        }
        if (!ok || !(thr >= 0.f)) thr = INFINITY;   // unusable seeds / NaN: no pruning for this lane
    }
    // query-group boxes (4 groups of 16 lanes); the group thresholds are refreshed per chunk
    float gl0 = qx, gl1 = qy, gl2 = qz, gh0 = qx, gh1 = qy, gh2 = qz;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
        gl0 = fminf(gl0, __shfl_xor(gl0, o, 64)); gh0 = fmaxf(gh0, __shfl_xor(gh0, o, 64));
        gl1 = fminf(gl1, __shfl_xor(gl1, o, 64)); gh1 = fmaxf(gh1, __shfl_xor(gh1, o, 64));
        gl2 = fminf(gl2, __shfl_xor(gl2, o, 64)); gh2 = fmaxf(gh2, __shfl_xor(gh2, o, 64));
    }

    float bm[KK];
    int bb[KK];
#pragma unroll
    for (int k = 0; k < KK; ++k) { bm[k] = INFINITY; bb[k] = -1; }
    unsigned short *my_list = s_list + (size_t)wv * LN_CAP * 64 + lane;

    for (int c0 = 0; c0 < n2; c0 += LN_CHUNK) {
        __syncthreads();   // the previous chunk has been consumed
        // ---- stage targets and boxes of this chunk (+INF beyond the padded row)
        for (int e = tid * 4; e < 3 * LN_CHUNK; e += 64 * LN_WAVES * 4) {
            const int ax = e / LN_CHUNK, off = e - ax * LN_CHUNK;
            float4 v = {INFINITY, INFINITY, INFINITY, INFINITY};
            if (c0 + off < jb.Ppad) v = *(const float4 *)(tx + (size_t)ax * jb.Ppad + c0 + off);
            *(float4 *)(s_t + ax * LN_AX + (off >> 4) * LN_TS + (off & 15)) = v;
        }
        for (int e = tid; e < LN_NBOX * 2; e += 64 * LN_WAVES) {
            float4 v = {INFINITY, INFINITY, INFINITY, INFINITY};
            if (c0 / NN_BOX + (e >> 1) < nbox_all) v = *(const float4 *)(bx + (size_t)(c0 / NN_BOX) * 8 + e * 4);
            ((e & 1) ? s_bb : s_ba)[e >> 1] = v;
        }
        __syncthreads();
        if (!wave_live) continue;
        const float *s_tx = s_t, *s_ty = s_t + LN_AX, *s_tz = s_t + 2 * LN_AX;

        // ---- super box `lane` = union of boxes 4 lane .. 4 lane + 3 (a padded box is (+INF,+INF))
        float lo0 = INFINITY, lo1 = INFINITY, lo2 = INFINITY, hi0 = -INFINITY, hi1 = -INFINITY, hi2 = -INFINITY;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float4 A = s_ba[4 * lane + k], Bv = s_bb[4 * lane + k];
            lo0 = fminf(lo0, A.x); lo1 = fminf(lo1, A.y); lo2 = fminf(lo2, A.z);
            hi0 = fmaxf(hi0, A.w == INFINITY ? -INFINITY : A.w);
            hi1 = fmaxf(hi1, Bv.x == INFINITY ? -INFINITY : Bv.x);
            hi2 = fmaxf(hi2, Bv.y == INFINITY ? -INFINITY : Bv.y);
        }
        if (hi0 == -INFINITY) { hi0 = INFINITY; hi1 = INFINITY; hi2 = INFINITY; }
        // ---- coarse filter: box-to-box against the 4 query groups
        float gt = thr;
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) gt = fmaxf(gt, __shfl_xor(gt, o, 64));
        bool pass = false;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float lb = ln_lb(lo0, lo1, lo2, hi0, hi1, hi2, ln_rl(gl0, 16 * q), ln_rl(gl1, 16 * q), ln_rl(gl2, 16 * q),
                                   ln_rl(gh0, 16 * q), ln_rl(gh1, 16 * q), ln_rl(gh2, 16 * q));
            pass = pass || (lb <= ln_rl(gt, 16 * q));
        }
        unsigned long long wm = __ballot(pass);
        LANE_STAT(KK == 1 ? 0 : 4, __builtin_popcountll(wm));
        // ---- per-lane mask of super boxes
        unsigned int mlo = 0u, mhi = 0u;
        while (wm) {
            const int bit = __builtin_ctzll(wm);
            wm &= wm - 1;
            const float lb = ln_lb(ln_rl(lo0, bit), ln_rl(lo1, bit), ln_rl(lo2, bit), ln_rl(hi0, bit), ln_rl(hi1, bit),
                                   ln_rl(hi2, bit), qx, qy, qz, qx, qy, qz);
            const bool p = lb <= thr;
            if (bit < 32) mlo |= p ? (1u << bit) : 0u;
            else mhi |= p ? (1u << (bit - 32)) : 0u;
        }
        // ---- walk the lane's own super boxes; survivors go to the lane's list; scan when nearly full
        int cnt = 0;
        bool more = true;
        while (more) {
            const bool has = (mlo | mhi) != 0u;
            if (__any(has)) {
                int sb = 0;
                if (has) {
                    sb = mlo ? __builtin_ctz(mlo) : 32 + __builtin_ctz(mhi);
                    if (mlo) mlo &= mlo - 1; else mhi &= mhi - 1;
                }
                LANE_STAT(KK == 1 ? 1 : 5, 1);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float4 A = s_ba[4 * sb + k], Bv = s_bb[4 * sb + k];
                    const float lb = ln_lb(A.x, A.y, A.z, A.w, Bv.x, Bv.y, qx, qy, qz, qx, qy, qz);
                    if (has && lb <= thr) { my_list[cnt * 64] = (unsigned short)(4 * sb + k); ++cnt; }
                }
            }
            more = __any(has);
            if (!more || __any(cnt > LN_CAP - 4)) {
#ifdef REART_PRUNE_STATS
                atomicAdd(&g_lane_stats[KK == 1 ? 3 : 7], (unsigned long long)cnt);   // total (query, box) pairs
#endif
                // ---- scan the lists: step t handles entry t of every lane that has one
                for (int t = 0; __any(t < cnt); ++t) {
                    LANE_STAT(KK == 1 ? 2 : 6, 1);
                    if (t < cnt) {
                        const int id = my_list[t * 64];
                        const int j0 = id * NN_BOX, l0 = id * LN_TS;
                        float4 X[4], Y[4], Z[4];
#pragma unroll
                        for (int h = 0; h < 4; ++h) {
                            X[h] = *(const float4 *)(s_tx + l0 + 4 * h);
                            Y[h] = *(const float4 *)(s_ty + l0 + 4 * h);
                            Z[h] = *(const float4 *)(s_tz + l0 + 4 * h);
                        }
                        if (KK == 1) {
                            float m = INFINITY;
#pragma unroll
                            for (int h = 0; h < 4; ++h) m = ln_min4(m, ln_d4(qx, qy, qz, X[h], Y[h], Z[h]));
                            if (m < bm[0]) { bm[0] = m; bb[0] = c0 + j0; }   // ascending visits: ties keep the lower box
                            thr = fminf(thr, m);
                        } else {
#pragma unroll
                            for (int h2 = 0; h2 < 2; ++h2) {
                                float m = ln_min4(INFINITY, ln_d4(qx, qy, qz, X[2 * h2], Y[2 * h2], Z[2 * h2]));
                                m = ln_min4(m, ln_d4(qx, qy, qz, X[2 * h2 + 1], Y[2 * h2 + 1], Z[2 * h2 + 1]));
                                const int jblk = c0 + j0 + 8 * h2;
#pragma unroll
                                for (int c = KK - 1; c >= 0; --c) {
                                    const int cp = c > 0 ? c - 1 : 0;
                                    const bool lt_prev = (c > 0) && (m < bm[cp]);
                                    const bool lt_cur = m < bm[c];
                                    bm[c] = lt_prev ? bm[cp] : (lt_cur ? m : bm[c]);
                                    bb[c] = lt_prev ? bb[cp] : (lt_cur ? jblk : bb[c]);
                                }
                            }
                            thr = fminf(thr, bm[KK - 1]);   // KK distinct blocks hold KK distinct targets <= bm[KK-1]
                        }
                    }
                }
                cnt = 0;
            }
        }
    }

    if (!wave_live) return;
    if (KK == 1) {
        // exact (lowest) index inside the winning box
        int bi = 0x7fffffff;
        const int blk = bb[0] < 0 ? 0 : bb[0];
#pragma unroll
        for (int h = 3; h >= 0; --h) {
            const float4 x = *(const float4 *)(tx + blk + 4 * h), y = *(const float4 *)(ty + blk + 4 * h),
                         z = *(const float4 *)(tz + blk + 4 * h);
            const float4 d = ln_d4(qx, qy, qz, x, y, z);
            if (d.w == bm[0]) bi = blk + 4 * h + 3;
            if (d.z == bm[0]) bi = blk + 4 * h + 2;
            if (d.y == bm[0]) bi = blk + 4 * h + 1;
            if (d.x == bm[0]) bi = blk + 4 * h;
        }
        bb[0] = bb[0] < 0 ? 0x7fffffff : bi;
    }
    if (i >= jb.P1) return;
    const size_t o = ((size_t)b * jb.P1 + i) * KK;
#pragma unroll
    for (int k = 0; k < KK; ++k) { jb.pd[o + k] = bm[k]; jb.pi[o + k] = bb[k]; }
}

template <int KK>
static int lane_launch(const KnnArgs &a, hipStream_t st) {
    const size_t lds = sizeof(float) * (3 * LN_AX + LN_NBOX * 8) + sizeof(unsigned short) * LN_WAVES * LN_CAP * 64;
    if (lds > REART_LDS_DEFAULT_CAP &&
        hipFuncSetAttribute((const void *)knn_lane_kernel<KK>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            152 * 1024) != hipSuccess)
        return REART_ERR_LAUNCH;
    const int njobs = a.items > a.items0 ? 2 : 1;
    const int wgs = (a.job[0].nqg + LN_WAVES - 1) / LN_WAVES;
    hipLaunchKernelGGL((knn_lane_kernel<KK>), dim3(njobs * a.N * wgs), dim3(64 * LN_WAVES), lds, st, a);
    REART_CHECK_LAUNCH();
    return REART_OK;
}

// a.S must be 1 (the outputs are final per query, in the S = 1 partial-list layout); with two jobs
// both must have the same number of query groups (they alternate over the grid).
int reart_knn_launch_lane(const KnnArgs &a, int KK, hipStream_t st) {
    const int njobs = a.items > a.items0 ? 2 : 1;
    for (int j = 0; j < njobs; ++j)
        if (!a.job[j].boxes || !a.job[j].seed || (a.job[j].Ppad % NN_BOX) != 0) return REART_ERR_INVALID_ARG;
    if (a.S != 1 || (njobs == 2 && a.job[0].nqg != a.job[1].nqg)) return REART_ERR_INVALID_ARG;
    switch (KK) {
        case 1: return lane_launch<1>(a, st);
        case 3: return lane_launch<3>(a, st);
        default: return REART_ERR_UNSUPPORTED;
    }
}
