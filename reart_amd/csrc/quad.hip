// reart_amd/csrc/quad.hip -- the warm-started, box-pruned EXACT search of prune.hip with a finer work
// shape: one wave = 16 queries x 4 box slots.
//
// Same contract and same answers as knn.hip / prune.hip (distance ((dx*dx)+(dy*dy))+(dz*dz) in fp32,
// strict '<', ties -> lowest index; reference utils/chamfer.py:78-94, utils/flow_utils.py:158).
//
// prune.hip gives a wave 64 queries and one quarter of the target boxes (slice s of 4): a box is scanned
// by all 64 lanes when ANY of the 64 queries needs it.  Since its targets reach the ALUs through LDS
// (not through wave-uniform scalar registers any more), different lanes may work on different boxes at
// no extra cost.  Here lane = (query q = lane & 15, slot s = lane >> 4): the wave owns 16 neighbouring
// queries (one k-d leaf) and ALL boxes of the cloud, slot s taking the boxes == s (mod 4).  Per step the
// four slots test / scan four different boxes, and a box is scanned when one of only 16 queries needs it:
//   1. warm start thr and the 16-query group box (shuffles inside 16 lanes);
//   2. coarse filter (128 boxes per pass): lane m tests boxes 2m, 2m+1 against the group box -> four
//      masks (one per slot class), corners parked in LDS;
//   3. precise filter: per step every slot pops its next candidate, each lane tests ITS query against
//      ITS slot's box (one point-to-box bound per lane: four boxes per step), a ballot tells which slots
//      need their box -> four masks of boxes to scan;
//   4. scan: per step every slot pops its next box, the 16 lanes of the slot load its 48 coordinates,
//      park them in LDS and read them back as broadcasts; brute-force inner loop in packed fp32;
//   5. the four slots of a query merge by the (distance, index) key (two shuffles); one record per query
//      leaves the kernel (no slice partials).
// Exactness: as in prune.hip -- monotone lower bounds, strict comparisons, thr an upper bound of the
// K-th neighbour distance, ascending visits inside a slot and a full-key merge across slots.
#include "common.h"
#include "internal.h"
#include <math.h>

typedef float f2 __attribute__((ext_vector_type(2)));
#ifdef REART_PRUNE_STATS
__device__ unsigned long long g_quad_stats[8];
extern "C" int reart_debug_quad_stats(unsigned long long *out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_quad_stats), sizeof(g_quad_stats)) != hipSuccess) return REART_ERR_LAUNCH;
    if (reset) { unsigned long long z[8] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_quad_stats), z, sizeof(z)); }
    return REART_OK;
}
#define QUAD_STAT(k, v) do { if ((threadIdx.x & 63) == 0) atomicAdd(&g_quad_stats[k], (unsigned long long)(v)); } while (0)
#else
#define QUAD_STAT(k, v) do { } while (0)
#endif
#define QD_CHUNK 128   // boxes handled per pass (2 per lane): 4 KB of corners in LDS per wave

__device__ __forceinline__ float qd_lb(float4 A, float4 Bv, float qlo0, float qlo1, float qlo2, float qhi0, float qhi1,
                                       float qhi2) {
    const float ex = fmaxf(fmaxf(A.x - qhi0, qlo0 - A.w), 0.f);
    const float ey = fmaxf(fmaxf(A.y - qhi1, qlo1 - Bv.x), 0.f);
    const float ez = fmaxf(fmaxf(A.z - qhi2, qlo2 - Bv.y), 0.f);
    return (ex * ex + ey * ey) + ez * ez;
}
// pop the lowest set bit of each of four masks (uniform)
__device__ __forceinline__ void qd_pop4(unsigned long long (&m)[4], int (&bit)[4], bool (&has)[4]) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        has[s] = m[s] != 0ull;
        bit[s] = has[s] ? __builtin_ctzll(m[s]) : 0;
        m[s] &= m[s] - 1ull;   // 0 stays 0
    }
}

// KK = 1: (distance, exact index) per query.  KK = 3: the 3 best BLOCKS of 8 targets (block minimum,
// first index), rescanned by the consumer.  Output = the S = 1 partial-list layout of knn.hip.
struct QuadLds {
    float4 ca[QD_CHUNK], cb[QD_CHUNK];   // box corners of the current pass (lo.xyz hi.x | hi.y hi.z - -)
    float tg[4 * 48];                    // targets of the four boxes of a step (x[16] | y[16] | z[16] per slot)
};
template <int KK>
__device__ __forceinline__ void knn_quad_body(const KnnArgs &a, const int w, QuadLds &L) {
    float4 *s_ca = L.ca, *s_cb = L.cb;
    float *s_tg = L.tg;
    if (w >= a.items) return;
    const bool two = a.items > a.items0;
    const int jsel = two ? (w & 1) : 0;
    const KnnJob jb = a.job[jsel];
    const int wl = two ? (w >> 1) : w;
    const int nq16 = (jb.P1 + 15) >> 4;
    const int b = wl / nq16, g = wl - b * nq16;
    const int lane = threadIdx.x & 63, q = lane & 15, slot = lane >> 4;

    const int i = g * 16 + q;
    const int ic = i < jb.P1 ? i : jb.P1 - 1;
    const int qb = jb.qmap ? jb.qmap[b] : b;
    const float *qp = (qb < 0 ? jb.q_alt : jb.q + (size_t)qb * jb.P1 * 3) + (size_t)ic * 3;
    const float qx = qp[0], qy = qp[1], qz = qp[2];
    const f2 qx2 = {qx, qx}, qy2 = {qy, qy}, qz2 = {qz, qz};

    const float *tx = jb.tsoa + (size_t)b * 3 * jb.Ppad;
    const float *ty = tx + jb.Ppad;
    const float *tz = ty + jb.Ppad;
    const int n2 = jb.tlen ? jb.tlen[b] : jb.P2;
    const int nbox = (n2 + NN_BOX - 1) / NN_BOX, nbox_all = jb.Ppad / NN_BOX;
    const float *bx = jb.boxes + (size_t)b * nbox_all * 8;

    // ---- warm start (identical in the four slots of a query)
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
    // ---- box of the 16 queries and their largest bound (every lane ends up with them)
    float gl0 = qx, gl1 = qy, gl2 = qz, gh0 = qx, gh1 = qy, gh2 = qz, gt = thr;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
        gl0 = fminf(gl0, __shfl_xor(gl0, o, 64)); gh0 = fmaxf(gh0, __shfl_xor(gh0, o, 64));
        gl1 = fminf(gl1, __shfl_xor(gl1, o, 64)); gh1 = fmaxf(gh1, __shfl_xor(gh1, o, 64));
        gl2 = fminf(gl2, __shfl_xor(gl2, o, 64)); gh2 = fmaxf(gh2, __shfl_xor(gh2, o, 64));
        gt = fmaxf(gt, __shfl_xor(gt, o, 64));
    }

    float bm[KK];
    int bb[KK];
#pragma unroll
    for (int k = 0; k < KK; ++k) { bm[k] = INFINITY; bb[k] = 0x7fffffff; }

    for (int cb = 0; cb < nbox; cb += QD_CHUNK) {
        // ---- coarse filter: lane m holds the pass's boxes 2m and 2m+1 (local ids); box l belongs to slot
        // class l & 3, so even lanes hold classes 0,1 and odd lanes classes 2,3.  A mask bit is a lane index:
        // local box id = 2 * bit + (class & 1).  Corners go to LDS for the per-lane tests below.
        unsigned long long cm[4];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int bid = cb + 2 * lane + j;
            float4 A = {INFINITY, INFINITY, INFINITY, INFINITY}, Bv = A;
            if (bid < nbox_all) {
                A = *(const float4 *)(bx + (size_t)bid * 8);
                Bv = *(const float4 *)(bx + (size_t)bid * 8 + 4);
            }
            s_ca[2 * lane + j] = A;
            s_cb[2 * lane + j] = Bv;
            const float lb = qd_lb(A, Bv, gl0, gl1, gl2, gh0, gh1, gh2);
            const unsigned long long bal = __ballot(bid < nbox && lb <= gt);
            cm[j] = bal & 0x5555555555555555ull;
            cm[2 + j] = bal & 0xaaaaaaaaaaaaaaaaull;
        }
        QUAD_STAT(KK == 1 ? 0 : 4, 1);
        QUAD_STAT(KK == 1 ? 1 : 5, __builtin_popcountll(cm[0]) + __builtin_popcountll(cm[1]) + __builtin_popcountll(cm[2]) + __builtin_popcountll(cm[3]));
        // ---- precise filter: four boxes per step, one per slot, each lane its own query
        unsigned long long pm[4] = {0ull, 0ull, 0ull, 0ull};
        while ((cm[0] | cm[1]) | (cm[2] | cm[3])) {
            int bit[4];
            bool has[4];
            qd_pop4(cm, bit, has);
            QUAD_STAT(KK == 1 ? 2 : 6, 1);
            const int mb = slot == 0 ? bit[0] : (slot == 1 ? bit[1] : (slot == 2 ? bit[2] : bit[3]));
            const bool hv = slot == 0 ? has[0] : (slot == 1 ? has[1] : (slot == 2 ? has[2] : has[3]));
            const int bl = 2 * mb + (slot & 1);
            const float lb = qd_lb(s_ca[bl], s_cb[bl], qx, qy, qz, qx, qy, qz);
            const unsigned long long bal = __ballot(hv && lb <= thr);
#pragma unroll
            for (int s = 0; s < 4; ++s)
                if ((bal >> (16 * s)) & 0xffffull) pm[s] |= 1ull << bit[s];
        }
        // ---- scan: four boxes per step
        while ((pm[0] | pm[1]) | (pm[2] | pm[3])) {
            int bit[4];
            bool has[4];
            qd_pop4(pm, bit, has);
            QUAD_STAT(KK == 1 ? 3 : 7, 1);
            const int mb = slot == 0 ? bit[0] : (slot == 1 ? bit[1] : (slot == 2 ? bit[2] : bit[3]));
            const bool hv = slot == 0 ? has[0] : (slot == 1 ? has[1] : (slot == 2 ? has[2] : has[3]));
            const int j0 = hv ? (cb + 2 * mb + (slot & 1)) * NN_BOX : 0;
            // the 16 lanes of a slot fetch the 48 coordinates of its box and park them in LDS
            const float vx = tx[j0 + q], vy = ty[j0 + q], vz = tz[j0 + q];
            float *sg = s_tg + slot * 48;
            sg[q] = vx; sg[16 + q] = vy; sg[32 + q] = vz;
            const float *lx = sg, *ly = sg + 16, *lz = sg + 32;   // read back as broadcasts inside the slot
            if (KK == 1) {
                float m = INFINITY;
#pragma unroll
                for (int u = 0; u < NN_BOX; u += 2) {
                    const f2 dx = qx2 - *(const f2 *)(lx + u);
                    const f2 dy = qy2 - *(const f2 *)(ly + u);
                    const f2 dz = qz2 - *(const f2 *)(lz + u);
                    const f2 d = (dx * dx + dy * dy) + dz * dz;
                    m = fminf(fminf(m, d.x), d.y);
                }
                if (hv && m < bm[0]) { bm[0] = m; bb[0] = j0; }   // ascending visits inside a slot: ties keep the lower box
            } else {
#pragma unroll
                for (int h = 0; h < NN_BOX; h += 8) {
                    float m = INFINITY;
#pragma unroll
                    for (int u = 0; u < 8; u += 2) {
                        const f2 dx = qx2 - *(const f2 *)(lx + h + u);
                        const f2 dy = qy2 - *(const f2 *)(ly + h + u);
                        const f2 dz = qz2 - *(const f2 *)(lz + h + u);
                        const f2 d = (dx * dx + dy * dy) + dz * dz;
                        m = fminf(fminf(m, d.x), d.y);
                    }
                    if (!hv) m = INFINITY;
#pragma unroll
                    for (int c = KK - 1; c >= 0; --c) {
                        const int cp = c > 0 ? c - 1 : 0;
                        const bool lt_prev = (c > 0) && (m < bm[cp]);
                        const bool lt_cur = m < bm[c];
                        bm[c] = lt_prev ? bm[cp] : (lt_cur ? m : bm[c]);
                        bb[c] = lt_prev ? bb[cp] : (lt_cur ? j0 + h : bb[c]);
                    }
                }
            }
        }
        // between chunks the bound may shrink to what has been found (KK distinct blocks hold KK distinct
        // targets no farther than bm[KK-1]); the four slots of a query share the tightest one
        if (cb + QD_CHUNK < nbox) {
            float t = bm[KK - 1];
            if (KK > 1) t = INFINITY;      // a slot's K-th block bound is not a bound for the query's K-th
            t = fminf(t, __shfl_xor(t, 16, 64));
            t = fminf(t, __shfl_xor(t, 32, 64));
            thr = fminf(thr, t);
        }
    }

    if (KK == 1) {
        // exact (lowest) index inside the slot's winning box, then the four slots of the query merge by key
        int bi = 0x7fffffff;
        const int blk = bb[0] == 0x7fffffff ? 0 : bb[0];
#pragma unroll
        for (int h = 3; h >= 0; --h) {
            const float4 x = *(const float4 *)(tx + blk + 4 * h), y = *(const float4 *)(ty + blk + 4 * h),
                         z = *(const float4 *)(tz + blk + 4 * h);
            if (reart_sqdist3(qx, qy, qz, x.w, y.w, z.w) == bm[0]) bi = blk + 4 * h + 3;
            if (reart_sqdist3(qx, qy, qz, x.z, y.z, z.z) == bm[0]) bi = blk + 4 * h + 2;
            if (reart_sqdist3(qx, qy, qz, x.y, y.y, z.y) == bm[0]) bi = blk + 4 * h + 1;
            if (reart_sqdist3(qx, qy, qz, x.x, y.x, z.x) == bm[0]) bi = blk + 4 * h;
        }
        if (bb[0] == 0x7fffffff) bi = 0x7fffffff;
        float d = bm[0];
#pragma unroll
        for (int o = 16; o <= 32; o <<= 1) {
            const float od = __shfl_xor(d, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            const bool l = (od < d) | ((od == d) & (oi < bi));
            d = l ? od : d; bi = l ? oi : bi;
        }
        bm[0] = d; bb[0] = bi;
    } else {
        // merge the four slots' ascending 3-lists by (minimum, block index)
#pragma unroll
        for (int o = 16; o <= 32; o <<= 1) {
            float od[KK];
            int oi[KK];
#pragma unroll
            for (int k = 0; k < KK; ++k) { od[k] = __shfl_xor(bm[k], o, 64); oi[k] = __shfl_xor(bb[k], o, 64); }
#pragma unroll
            for (int k = 0; k < KK; ++k) {
                const float d = od[k];
                const int j = oi[k];
#pragma unroll
                for (int c = KK - 1; c >= 0; --c) {
                    const int cp = c > 0 ? c - 1 : 0;
                    const bool lp = (c > 0) && ((d < bm[cp]) | ((d == bm[cp]) & (j < bb[cp])));
                    const bool lc = (d < bm[c]) | ((d == bm[c]) & (j < bb[c]));
                    bm[c] = lp ? bm[cp] : (lc ? d : bm[c]);
                    bb[c] = lp ? bb[cp] : (lc ? j : bb[c]);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < KK; ++k)
            if (bb[k] == 0x7fffffff) bb[k] = -1;   // consumer convention: no block
    }
    if (slot != 0 || i >= jb.P1) return;
    const size_t o = ((size_t)b * jb.P1 + i) * KK;
#pragma unroll
    for (int k = 0; k < KK; ++k) { jb.pd[o + k] = bm[k]; jb.pi[o + k] = bb[k]; }
}

template <int KK>
__global__ __launch_bounds__(64) void knn_quad_kernel(KnnArgs a) {
    __shared__ QuadLds L;
    knn_quad_body<KK>(a, blockIdx.x, L);
}

// both searches of an iteration in one launch (see prune.hip): two K = 1 items alternate with one K = 3 item
struct KnnQuadPairArgs { KnnArgs k1, k3; int mixed; };
__global__ __launch_bounds__(64) void knn_quad_pair_kernel(KnnQuadPairArgs a) {
    __shared__ QuadLds L;
    const int w = blockIdx.x;
    if (w < a.mixed) {
        const int q = w / 3, r = w - 3 * q;
        if (r < 2) knn_quad_body<1>(a.k1, 2 * q + r, L);
        else knn_quad_body<3>(a.k3, q, L);
    } else {
        const int d1 = 2 * (a.mixed / 3), d3 = a.mixed / 3;
        const int v = w - a.mixed;
        if (v < a.k1.items - d1) knn_quad_body<1>(a.k1, d1 + v, L);
        else if (v - (a.k1.items - d1) < a.k3.items - d3) knn_quad_body<3>(a.k3, d3 + (v - (a.k1.items - d1)), L);
    }
}

// a.S must be 1; a.items0 / a.items count 16-query groups: items0 = N * ceil(P1 / 16) per job
static int quad_check(const KnnArgs &a) {
    const int njobs = a.items > a.items0 ? 2 : 1;
    for (int j = 0; j < njobs; ++j)
        if (!a.job[j].boxes || !a.job[j].seed || (a.job[j].Ppad % NN_BOX) != 0) return REART_ERR_INVALID_ARG;
    if (a.S != 1 || (a.items != a.items0 && a.items != 2 * a.items0)) return REART_ERR_INVALID_ARG;
    return REART_OK;
}
int reart_knn_launch_quad(const KnnArgs &a, int KK, hipStream_t st) {
    const int rc = quad_check(a);
    if (rc != REART_OK) return rc;
    switch (KK) {
        case 1: hipLaunchKernelGGL((knn_quad_kernel<1>), dim3(a.items), dim3(64), 0, st, a); break;
        case 3: hipLaunchKernelGGL((knn_quad_kernel<3>), dim3(a.items), dim3(64), 0, st, a); break;
        default: return REART_ERR_UNSUPPORTED;
    }
    REART_CHECK_LAUNCH();
    return REART_OK;
}
int reart_knn_launch_quad_pair(const KnnArgs &k1, const KnnArgs &k3, hipStream_t st) {
    int rc = quad_check(k1);
    if (rc == REART_OK) rc = quad_check(k3);
    if (rc != REART_OK) return rc;
    if (k1.items != 2 * k1.items0 || k3.items != k3.items0) return REART_ERR_INVALID_ARG;
    KnnQuadPairArgs a;
    a.k1 = k1; a.k3 = k3;
    const int m = (k1.items / 2 < k3.items) ? k1.items / 2 : k3.items;
    a.mixed = 3 * m;
    hipLaunchKernelGGL(knn_quad_pair_kernel, dim3(k1.items + k3.items), dim3(64), 0, st, a);
    REART_CHECK_LAUNCH();
    return REART_OK;
}
