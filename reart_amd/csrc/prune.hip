// reart_amd/csrc/prune.hip -- EXACT nearest-neighbour search with bounding-box pruning and a
// warm start, for the relaxation loop (reference run_robot.py:154-221: every iteration repeats the
// Chamfer search of utils/chamfer.py:78-94 and the k=3 search of utils/flow_utils.py:158 on clouds
// that moved a little since the previous iteration).
//
// Same answer, bit for bit, as the brute-force kernels of knn.hip (same distance expression,
// strict '<', ties -> lowest index); what changes is how many targets are looked at.
//
//   * Clouds are stored in Morton order (host side, relax.py), so 16 consecutive targets form a
//     compact box and the 64 queries of a wave are neighbours.
//   * Warm start: the neighbour indices of the PREVIOUS iteration (any valid indices would do)
//     give each query an upper bound thr on its K-th neighbour distance before anything is
//     scanned:  thr = max_k d(q, t[seed_k])  over K distinct seeds.
//   * Coarse filter, one box per lane: the wave's queries are summarised by 4 group boxes
//     (16 lanes each) with the group's largest thr; a target box survives if its box-to-box
//     lower bound is <= that thr for some group.  One ballot gives the survivor mask of 64 boxes.
//   * Precise filter, one query per lane: point-to-box lower bound against the lane's own thr
//     (which keeps shrinking as better candidates are found); the box is scanned if any lane
//     needs it.  Scanning is the brute-force inner loop of knn.hip in packed fp32; the 48 coordinates
//     of a scanned box are fetched through the vector memory path (lane l < 48 loads one float) into
//     a small LDS buffer and read back as broadcasts (a chain of scalar loads per scanned box -- the
//     brute-force kernel's operand path -- stalled on scalar-cache misses here: boxes are visited in
//     a data-dependent order and each is one cache line per axis).
//
// Exactness.  Both lower bounds are evaluated with the operations of the distance itself,
//   lb = ((ex*ex)+(ey*ey))+(ez*ez),  e = max(lo - q, q - hi, 0)  per axis,
// and fp32 subtraction, multiplication and addition are monotone, so lb <= d(q,t) for every
// target t inside the box (and the box-to-box bound is <= the point-to-box bound of every query
// of the group).  A box is dropped only when lb > thr strictly; a target that beats or TIES the
// current K-th candidate has d <= thr and therefore sits in a box that is scanned.  Boxes are
// dealt to the S slices round-robin; the consumer merges slice results by the full (d, index) key.
#include "common.h"
#include "internal.h"
#include <math.h>
#include <string.h>

typedef float f2 __attribute__((ext_vector_type(2)));
#define PR_WPB 1   // waves per workgroup (independent work items, no barrier); measured: 4 is slower (a
                   // workgroup's slots are held until its slowest wave ends)
#define PR_PF 8    // boxes whose targets are prefetched together
#ifndef PR_PREFETCH
#define PR_PREFETCH 0   // 1: fetch the targets of every surviving box PR_PF at a time, before the precise tests.  0: fetch a
                        // box's targets when it is scanned -- fewer instructions and no loads for the 60 % of the boxes
                        // the precise test rejects; with 4 waves per SIMD the exposed latency is covered (56 -> 53 us)
#endif

#ifdef REART_PRUNE_STATS   // diagnostic build only (tools/prune_stats.py): how much the filters let through
__device__ unsigned long long g_prune_stats[8];
extern "C" int reart_debug_prune_stats(unsigned long long *out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prune_stats), sizeof(g_prune_stats)) != hipSuccess) return REART_ERR_LAUNCH;
    if (reset) { unsigned long long z[8] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_prune_stats), z, sizeof(z)); }
    return REART_OK;
}
#define PRUNE_STAT(k, v) do { if ((threadIdx.x & 63) == 0) atomicAdd(&g_prune_stats[k], (unsigned long long)(v)); } while (0)
#else
#define PRUNE_STAT(k, v) do { } while (0)
#endif
#ifdef REART_ITEM_CLOCK   // diagnostic build only (tools/item_clock.py): wave lifetime of every work item
// (start, end in s_memtime ticks, XCC id) of the last pair launch, items in launch order
__device__ unsigned long long g_item_clock[3 * 16384];
extern "C" int reart_debug_item_clock(unsigned long long *out, int n) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_item_clock), sizeof(unsigned long long) * 3 * (n < 16384 ? n : 16384)) == hipSuccess
               ? REART_OK : REART_ERR_LAUNCH;
}
#endif

// min / max of a lane's value with a DPP-permuted copy in ONE instruction (v_min_f32_dpp).  Written through the compiler
// the same step is three (v_mov_b32_dpp, a canonicalising v_max v,v -- fminf of a value of unknown origin -- and the
// v_min); the item prologue alone has 28 such steps.  Only used where EXEC is full (every source lane is live); the
// s_nop covers the two wait states a DPP read needs after a VALU write of the same register, which the hazard
// recogniser does not insert for inline assembly.  No signalling NaNs exist here; quiet NaNs behave like fminf / fmaxf.
#define PR_DPP_OP(NAME, OP, CTRLTXT)                                                                              \
    __device__ __forceinline__ float NAME(float v) {                                                              \
        float r;                                                                                                  \
        asm("s_nop 1\n\t" OP " %0, %1, %1 " CTRLTXT " row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v));            \
        return r;                                                                                                 \
    }
PR_DPP_OP(pr_min_q1, "v_min_f32_dpp", "quad_perm:[1,0,3,2]")
PR_DPP_OP(pr_min_q2, "v_min_f32_dpp", "quad_perm:[2,3,0,1]")
PR_DPP_OP(pr_min_hm, "v_min_f32_dpp", "row_half_mirror")
PR_DPP_OP(pr_min_rm, "v_min_f32_dpp", "row_mirror")
PR_DPP_OP(pr_max_q1, "v_max_f32_dpp", "quad_perm:[1,0,3,2]")
PR_DPP_OP(pr_max_q2, "v_max_f32_dpp", "quad_perm:[2,3,0,1]")
PR_DPP_OP(pr_max_hm, "v_max_f32_dpp", "row_half_mirror")
PR_DPP_OP(pr_max_rm, "v_max_f32_dpp", "row_mirror")
__device__ __forceinline__ float pr_row16_min(float v) { return pr_min_rm(pr_min_hm(pr_min_q2(pr_min_q1(v)))); }
__device__ __forceinline__ float pr_row16_max(float v) { return pr_max_rm(pr_max_hm(pr_max_q2(pr_max_q1(v)))); }

__device__ __forceinline__ float rl(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

__device__ __forceinline__ float box_lb(float lo0, float lo1, float lo2, float hi0, float hi1, float hi2,
                                        float qlo0, float qlo1, float qlo2, float qhi0, float qhi1, float qhi2) {
    const float ex = fmaxf(fmaxf(lo0 - qhi0, qlo0 - hi0), 0.f);
    const float ey = fmaxf(fmaxf(lo1 - qhi1, qlo1 - hi1), 0.f);
    const float ez = fmaxf(fmaxf(lo2 - qhi2, qlo2 - hi2), 0.f);
    return (ex * ex + ey * ey) + ez * ez;
}

// KK = 1: partial (distance, exact index) per slice.  KK = 3: partial top-3 BLOCKS of 8 targets
// (block minimum, first index of the block), rescanned by the consumer (flow_blend_kernel).
template <int KK>
__device__ __forceinline__ void knn_pruned_body(const KnnArgs &a, const int w) {
    // Work per item varies (it depends on how tight the warm start is), and the working set fits
    // every XCD's L2: no XCD-contiguous remap here -- consecutive items go to different XCDs and
    // the two jobs alternate, so every XCD gets the same mix of light and heavy items.
    if (w >= a.items) return;
    const bool two = a.items > a.items0;
    const int jsel = two ? (w & 1) : 0;
    const KnnJob jb = a.job[jsel];
    const int wl = two ? (w >> 1) : w;
    const int s = wl % a.S;                      // slices of one query group are neighbours
    const int gpos = wl / a.S;                                   // launch position of the (batch, query group) pair
    const int grp = jb.border ? jb.border[gpos] : gpos;
    const int b = grp / jb.nqg, g = grp - b * jb.nqg;
    const int lane = threadIdx.x & 63;

    const int i = g * NN_BS + lane;
    const int ic = i < jb.P1 ? i : jb.P1 - 1;
    const int qb = jb.qmap ? jb.qmap[b] : b;
    const float *qp = (qb < 0 ? jb.q_alt : jb.q + (size_t)qb * jb.P1 * 3) + (size_t)ic * 3;
    const float qx = qp[0], qy = qp[1], qz = qp[2];
    const f2 qx2 = {qx, qx}, qy2 = {qy, qy}, qz2 = {qz, qz};

    const float *tx = jb.tsoa + (size_t)b * 3 * jb.Ppad;
    const float *ty = tx + jb.Ppad;
    const float *tz = ty + jb.Ppad;
    const int n2 = jb.tlen ? jb.tlen[b] : jb.P2;
    const int nbox = (n2 + NN_BOX - 1) / NN_BOX;
    const float *bx = jb.boxes + (size_t)b * (jb.Ppad / NN_BOX) * 8;

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

    // ---- group summaries: 4 groups of 16 lanes
    const float gl0 = pr_row16_min(qx), gl1 = pr_row16_min(qy), gl2 = pr_row16_min(qz);
    const float gh0 = pr_row16_max(qx), gh1 = pr_row16_max(qy), gh2 = pr_row16_max(qz);
    const float gt = pr_row16_max(thr);
    float G[4][7];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        G[q][0] = rl(gl0, 16 * q); G[q][1] = rl(gl1, 16 * q); G[q][2] = rl(gl2, 16 * q);
        G[q][3] = rl(gh0, 16 * q); G[q][4] = rl(gh1, 16 * q); G[q][5] = rl(gh2, 16 * q);
        G[q][6] = rl(gt, 16 * q);
    }

    float bm[KK];
    int bb[KK];
#pragma unroll
    for (int k = 0; k < KK; ++k) { bm[k] = INFINITY; bb[k] = -1; }
    __shared__ __attribute__((aligned(16))) float s_tg_all[PR_WPB * PR_PF * 48];
    float *s_tg = s_tg_all + (threadIdx.x >> 6) * (PR_PF * 48);
    const float *tx_g = tx;

    int work = 16;   // uniform work counter of this item (prologue ~ 16 tests)
    const int sparse_max = a.sparse < 8 ? a.sparse : 8;   // boxes needed by at most this many queries take the sparse scan (0: off)
    const int sparse16_max = a.sparse > 8 ? a.sparse : 0;   // 9 .. 16 needers: the 16-query form
    const int per = 64 * a.S;
    for (int base = 0; base < nbox; base += per) {
        // ---- coarse filter: lane l looks at box base + l*S + s
        const int bid = base + lane * a.S + s;
        bool pass = false;
        float lo0 = INFINITY, lo1 = INFINITY, lo2 = INFINITY, hi0 = INFINITY, hi1 = INFINITY, hi2 = INFINITY;
        if (bid < nbox) {
            const float4 A = *(const float4 *)(bx + (size_t)bid * 8);
            const float4 Bv = *(const float4 *)(bx + (size_t)bid * 8 + 4);
            lo0 = A.x; lo1 = A.y; lo2 = A.z; hi0 = A.w; hi1 = Bv.x; hi2 = Bv.y;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float lb = box_lb(lo0, lo1, lo2, hi0, hi1, hi2, G[q][0], G[q][1], G[q][2], G[q][3], G[q][4], G[q][5]);
                pass = pass || (lb <= G[q][6]);
            }
        }
        unsigned long long mask = __ballot(pass);
        PRUNE_STAT(KK == 1 ? 0 : 3, 0);                                  // coarse rounds (K = 1)
        if (KK == 1) PRUNE_STAT(1, __builtin_popcountll(mask));          // boxes passing the coarse filter
        if (KK == 1) PRUNE_STAT(3, __builtin_popcountll(__ballot(thr == INFINITY)));  // lanes without a bound
        // scan of one box: the brute-force inner loop of knn.hip on its 16 targets
        auto scan_box = [&](const int bit, const int slot) {
            work += 5;                                                   // a scan costs about five tests
            if (KK == 1) PRUNE_STAT(2, 1);                               // boxes scanned
            const int j0 = (base + bit * a.S + s) * NN_BOX;
            // the box's 16 targets were staged in LDS slot `slot` (x[16] | y[16] | z[16]); all lanes read
            // the same addresses (broadcast) -- the operands of the packed ops are VGPR pairs
            if (!PR_PREFETCH) {
                const int l = lane < 48 ? lane : 47;
                const float v = tx_g[(size_t)(l >> 4) * jb.Ppad + j0 + (l & 15)];
                if (lane < 48) s_tg[lane] = v;
            }
            const float *tx = s_tg + (PR_PREFETCH ? slot * 48 : 0) - j0, *ty = tx + 16, *tz = tx + 32;
            if (KK == 1) {
                // minimum of each half of the box: the final rescan then looks at 8 targets instead of 16
                float mh[2] = {INFINITY, INFINITY};
#pragma unroll
                for (int u = 0; u < NN_BOX; u += 2) {
                    const f2 dx = qx2 - *(const f2 *)(tx + j0 + u);
                    const f2 dy = qy2 - *(const f2 *)(ty + j0 + u);
                    const f2 dz = qz2 - *(const f2 *)(tz + j0 + u);
                    const f2 d = (dx * dx + dy * dy) + dz * dz;
                    mh[u >> 3] = fminf(fminf(mh[u >> 3], d.x), d.y);
                }
                const float m = fminf(mh[0], mh[1]);
                // ascending visits: ties keep the lower box, and inside a box the lower half
                if (m < bm[0]) { bm[0] = m; bb[0] = j0 + (mh[1] < mh[0] ? 8 : 0); }
                thr = fminf(thr, m);
            } else {
                constexpr int UBK = 8;
#pragma unroll
                for (int h = 0; h < NN_BOX; h += UBK) {
                    float m = INFINITY;
#pragma unroll
                    for (int u = 0; u < UBK; u += 2) {
                        const f2 dx = qx2 - *(const f2 *)(tx + j0 + h + u);
                        const f2 dy = qy2 - *(const f2 *)(ty + j0 + h + u);
                        const f2 dz = qz2 - *(const f2 *)(tz + j0 + h + u);
                        const f2 d = (dx * dx + dy * dy) + dz * dz;
                        m = fminf(fminf(m, d.x), d.y);
                    }
#pragma unroll
                    for (int c = KK - 1; c >= 0; --c) {
                        const int cp = c > 0 ? c - 1 : 0;
                        const bool lt_prev = (c > 0) && (m < bm[cp]);
                        const bool lt_cur = m < bm[c];
                        bm[c] = lt_prev ? bm[cp] : (lt_cur ? m : bm[c]);
                        bb[c] = lt_prev ? bb[cp] : (lt_cur ? j0 + h : bb[c]);
                    }
                }
                // KK distinct blocks hold KK distinct targets no farther than bm[KK-1]
                thr = fminf(thr, bm[KK - 1]);
            }
        };
        // sparse form of the scan for a box that at most PR_SPARSE of the 64 queries need (40-60 % of the scanned
        // boxes, tools/prune_stats.py): instead of all 64 lanes walking the 16 targets for the sake of a few,
        // 8 lanes x 2 targets (packed) evaluate ONE needing query against the box, eight such queries side by side.
        // The needing lanes park their coordinates in LDS by their rank among the needers, row r of 8 lanes reads
        // query r, the row minimum (same distance expression, min is order-free) goes back to its lane through one
        // ds_bpermute.  A lane that does not need the box (lb > thr) is not updated: its minimum over this box
        // would exceed thr and can never be (or tie with) one of its K nearest after the slice merge.
        auto sparse_box = [&](const int bit, const unsigned long long need, const bool nd) {
            work += 2;
            if (KK == 1) PRUNE_STAT(2, 1);
            const int j0 = (base + bit * a.S + s) * NN_BOX;
            const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(need >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)need, 0u));
            float *s_qc = s_tg + 64;                                     // [8][4] behind the staged box of the dense scan
            if (nd) *(float4 *)(s_qc + 4 * rank) = make_float4(qx, qy, qz, 0.f);
            const int u = lane & 7;
            const float *tp = tx_g + j0 + 2 * u;
            const f2 txv = *(const f2 *)tp, tyv = *(const f2 *)(tp + jb.Ppad), tzv = *(const f2 *)(tp + 2 * (size_t)jb.Ppad);
            const float4 qc = *(const float4 *)(s_qc + 4 * (lane >> 3));
            const f2 cx = {qc.x, qc.x}, cy = {qc.y, qc.y}, cz = {qc.z, qc.z};
            const f2 dx = cx - txv, dy = cy - tyv, dz = cz - tzv;
            const f2 d = (dx * dx + dy * dy) + dz * dz;
            float m = fminf(d.x, d.y);
            m = pr_min_q2(pr_min_q1(m));                                 // quad: targets 8h .. 8h+7 of the box
            if (KK == 1) {
                // the two half-box minima (lanes 0..3 / 4..7 of the row) come back separately: the winner's half
                const float r0 = __int_as_float(__builtin_amdgcn_ds_bpermute(rank << 5, __float_as_int(m)));
                const float r1 = __int_as_float(__builtin_amdgcn_ds_bpermute((rank << 5) + 16, __float_as_int(m)));
                const float r = nd ? fminf(r0, r1) : INFINITY;
                if (r < bm[0]) { bm[0] = r; bb[0] = j0 + (r1 < r0 ? 8 : 0); }
                thr = fminf(thr, r);
            } else {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    float r = __int_as_float(__builtin_amdgcn_ds_bpermute((rank << 5) + (h << 4), __float_as_int(m)));
                    r = nd ? r : INFINITY;
#pragma unroll
                    for (int c = KK - 1; c >= 0; --c) {
                        const int cp = c > 0 ? c - 1 : 0;
                        const bool lt_prev = (c > 0) && (r < bm[cp]);
                        const bool lt_cur = r < bm[c];
                        bm[c] = lt_prev ? bm[cp] : (lt_cur ? r : bm[c]);
                        bb[c] = lt_prev ? bb[cp] : (lt_cur ? j0 + 8 * h : bb[c]);
                    }
                }
                thr = fminf(thr, bm[KK - 1]);
            }
        };
        // the same for 9..16 needing queries: 4 lanes x 4 targets per query, sixteen queries side by side
        auto sparse16_box = [&](const int bit, const unsigned long long need, const bool nd) {
            work += 3;
            if (KK == 1) PRUNE_STAT(2, 1);
            const int j0 = (base + bit * a.S + s) * NN_BOX;
            const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(need >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)need, 0u));
            float *s_qc = s_tg + 64;                                     // [16][4]
            if (nd) *(float4 *)(s_qc + 4 * rank) = make_float4(qx, qy, qz, 0.f);
            const int u = lane & 3;
            const float *tp = tx_g + j0 + 4 * u;
            const float4 txv = *(const float4 *)tp, tyv = *(const float4 *)(tp + jb.Ppad), tzv = *(const float4 *)(tp + 2 * (size_t)jb.Ppad);
            const float4 qc = *(const float4 *)(s_qc + 4 * (lane >> 2));
            const f2 cx = {qc.x, qc.x}, cy = {qc.y, qc.y}, cz = {qc.z, qc.z};
            const f2 dxa = cx - f2{txv.x, txv.y}, dya = cy - f2{tyv.x, tyv.y}, dza = cz - f2{tzv.x, tzv.y};
            const f2 dxb = cx - f2{txv.z, txv.w}, dyb = cy - f2{tyv.z, tyv.w}, dzb = cz - f2{tzv.z, tzv.w};
            const f2 da = (dxa * dxa + dya * dya) + dza * dza;
            const f2 db = (dxb * dxb + dyb * dyb) + dzb * dzb;
            float m = fminf(fminf(da.x, da.y), fminf(db.x, db.y));
            m = pr_min_q1(m);                                            // lanes {0,1}: targets 0..7, lanes {2,3}: 8..15
            if (KK == 1) {
                const float r0 = __int_as_float(__builtin_amdgcn_ds_bpermute(rank << 4, __float_as_int(m)));
                const float r1 = __int_as_float(__builtin_amdgcn_ds_bpermute((rank << 4) + 8, __float_as_int(m)));
                const float r = nd ? fminf(r0, r1) : INFINITY;
                if (r < bm[0]) { bm[0] = r; bb[0] = j0 + (r1 < r0 ? 8 : 0); }
                thr = fminf(thr, r);
            } else {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    float r = __int_as_float(__builtin_amdgcn_ds_bpermute((rank << 4) + (h << 3), __float_as_int(m)));
                    r = nd ? r : INFINITY;
#pragma unroll
                    for (int c = KK - 1; c >= 0; --c) {
                        const int cp = c > 0 ? c - 1 : 0;
                        const bool lt_prev = (c > 0) && (r < bm[cp]);
                        const bool lt_cur = r < bm[c];
                        bm[c] = lt_prev ? bm[cp] : (lt_cur ? r : bm[c]);
                        bb[c] = lt_prev ? bb[cp] : (lt_cur ? j0 + 8 * h : bb[c]);
                    }
                }
                thr = fminf(thr, bm[KK - 1]);
            }
        };
        while (mask) {
            // ---- take the next PR_PF surviving boxes and put ALL their targets in flight: lane l < 48
            // loads one float of each box (192 coalesced bytes per box) through the vector memory path,
            // then the values are parked in the wave's LDS slots.  (Targets used to come through scalar
            // loads, s_load_dwordx16 x 3 per scanned box: one dependent scalar-cache miss chain per scan;
            // measured 95 -> 80 us for the launch with the vector path alone, before prefetching.)
            unsigned long long rest = PR_PREFETCH ? mask : 0ull;
#pragma unroll
            for (int k = 0; k < PR_PF; ++k) rest &= rest - 1;            // x & (x - 1) of 0 is 0
            unsigned long long cm = mask & ~rest;                        // this chunk's boxes
            mask = rest;
            if (PR_PREFETCH) {
                const int l = lane < 48 ? lane : 47;
                const float *src = tx_g + (size_t)(l >> 4) * jb.Ppad + (l & 15);
                float pv[PR_PF];
                unsigned long long t = cm;
#pragma unroll
                for (int k = 0; k < PR_PF; ++k) {
                    const int bit = t ? __builtin_ctzll(t) : 0;          // clamped: an unused slot re-reads box `base`
                    t &= t - 1;
                    pv[k] = src[(base + bit * a.S + s) * NN_BOX];
                }
#pragma unroll
                for (int k = 0; k < PR_PF; ++k)
                    if (lane < 48) s_tg[k * 48 + lane] = pv[k];
            }
            // ---- precise filter with the lane's current bound, two boxes per step in packed fp32 (same
            // operation order per box as box_lb; the second box is judged after the first one's scan)
            int slot = 0;
            while (cm) {
            const int bA = __builtin_ctzll(cm);
            cm &= cm - 1;
            const bool hasB = cm != 0;
            const int bB = hasB ? __builtin_ctzll(cm) : bA;
            if (hasB) cm &= cm - 1;
            const f2 L0 = {rl(lo0, bA), rl(lo0, bB)}, L1 = {rl(lo1, bA), rl(lo1, bB)}, L2 = {rl(lo2, bA), rl(lo2, bB)};
            const f2 H0 = {rl(hi0, bA), rl(hi0, bB)}, H1 = {rl(hi1, bA), rl(hi1, bB)}, H2 = {rl(hi2, bA), rl(hi2, bB)};
            const f2 a0 = L0 - qx2, c0 = qx2 - H0, a1 = L1 - qy2, c1 = qy2 - H1, a2 = L2 - qz2, c2 = qz2 - H2;
            const f2 e0 = {fmaxf(fmaxf(a0.x, c0.x), 0.f), fmaxf(fmaxf(a0.y, c0.y), 0.f)};
            const f2 e1 = {fmaxf(fmaxf(a1.x, c1.x), 0.f), fmaxf(fmaxf(a1.y, c1.y), 0.f)};
            const f2 e2 = {fmaxf(fmaxf(a2.x, c2.x), 0.f), fmaxf(fmaxf(a2.y, c2.y), 0.f)};
            const f2 lb = (e0 * e0 + e1 * e1) + e2 * e2;
            work += 2;
#pragma unroll 1   // one copy of the scan code (two copies cost 30 VGPRs of occupancy)
            for (int h = 0; h < (hasB ? 2 : 1); ++h) {
                const float lbh = h ? lb.y : lb.x;
#ifdef REART_PRUNE_STATS   // how many of the 64 queries need a scanned box: 1 | 2-3 | 4-7 | 8+  (K = 1 only)
                if (KK == 1) {
                    const int nl = __builtin_popcountll(__ballot(lbh <= thr));
                    if (nl) PRUNE_STAT(nl == 1 ? 4 : (nl < 4 ? 5 : (nl < 8 ? 6 : 7)), 1);
                }
#endif
                const bool nd = lbh <= thr;
                const unsigned long long need = __ballot(nd);
                if (need) {
                    const int nneed = __builtin_popcountll(need);
                    if (nneed <= sparse_max) sparse_box(h ? bB : bA, need, nd);
                    else if (nneed <= sparse16_max) sparse16_box(h ? bB : bA, need, nd);
                    else scan_box(h ? bB : bA, slot + h);
                }
            }
            slot += 2;
            }
        }
    }

    if (KK == 1) {
        // exact (lowest) index inside the winning half box
        int bi = 0x7fffffff;
        if (bb[0] >= 0) {
            const int blk = bb[0];
#pragma unroll
            for (int u = NN_BOX / 2 - 1; u >= 0; --u) {
                const float d = reart_sqdist3(qx, qy, qz, tx[blk + u], ty[blk + u], tz[blk + u]);
                if (d == bm[0]) bi = blk + u;
            }
        }
        bb[0] = bi;
    }
    if (jb.cost && lane == 0) jb.cost[wl] = (unsigned int)work;
    if (i >= jb.P1) return;
    const size_t o = (((size_t)s * a.N + b) * jb.P1 + i) * KK;
#pragma unroll
    for (int k = 0; k < KK; ++k) { jb.pd[o + k] = bm[k]; jb.pi[o + k] = bb[k]; }
}

template <int KK>
__global__ __launch_bounds__(NN_BS * PR_WPB) void knn_pruned_kernel(KnnArgs a) {
    knn_pruned_body<KK>(a, blockIdx.x * PR_WPB + (threadIdx.x >> 6));
}

// REART_SPARSE=n: boxes needed by at most n (0..8) queries of the wave take the sparse scan; default 8, 0 = always dense
static int reart_prune_pick_sparse(void) {
    const char *env = getenv("REART_SPARSE");
    const int n = env ? atoi(env) : 16;
    return n < 0 ? 0 : (n > 16 ? 16 : n);
}

// Both searches of one iteration in ONE launch: the K = 1 Chamfer items and the K = 3 flow items are
// independent, and a single dispatch lets them share the chip without the cross-queue fork / join of
// two streams (measured ~6-12 us per dependency edge in a replayed graph).  Two K = 1 items alternate
// with one K = 3 item while both kinds last.
struct KnnPairArgs { KnnArgs k1, k3; int mixed; unsigned int *ctr; };   // mixed = 3 * min(items1 / 2, items3)
__device__ __forceinline__ void knn_pruned_pair_item(const KnnPairArgs &a, const int w) {
    // item kind and index first, then ONE call site per body (each inlined copy is ~1 k instructions)
    int kind, idx;
    if (w < a.mixed) {
        const int q = w / 3, r = w - 3 * q;
        kind = r < 2 ? 1 : 3;
        idx = r < 2 ? 2 * q + r : q;
    } else {
        const int d1 = 2 * (a.mixed / 3), d3 = a.mixed / 3;   // items already dealt
        const int v = w - a.mixed, left1 = a.k1.items - d1;
        kind = v < left1 ? 1 : 3;
        idx = v < left1 ? d1 + v : d3 + (v - left1);
        if (kind == 3 && idx >= a.k3.items) return;
    }
    if (kind == 1) knn_pruned_body<1>(a.k1, idx);
    else knn_pruned_body<3>(a.k3, idx);
}
__global__ __launch_bounds__(NN_BS * PR_WPB) void knn_pruned_pair_kernel(KnnPairArgs a) {
#ifdef REART_ITEM_CLOCK
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#endif
    knn_pruned_pair_item(a, blockIdx.x * PR_WPB + (threadIdx.x >> 6));
#ifdef REART_ITEM_CLOCK
    if (threadIdx.x == 0 && blockIdx.x < 16384) {
        g_item_clock[3 * blockIdx.x] = t0;
        g_item_clock[3 * blockIdx.x + 1] = __builtin_amdgcn_s_memtime();
        g_item_clock[3 * blockIdx.x + 2] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 0xf;   // HW_REG_XCC_ID
    }
#endif
}
// Persistent form: a fixed number of one-wave workgroups (a few per SIMD) each take every
// gridDim.x-th work item.  (Drawing items from one device counter was tried: ~15 k atomics on one
// address serialise at the memory side and tripled the kernel time.)
__global__ __launch_bounds__(NN_BS) void knn_pruned_pair_persistent_kernel(KnnPairArgs a) {
    const int total = a.k1.items + a.k3.items;
    for (int w = blockIdx.x; w < total; w += gridDim.x) knn_pruned_pair_item(a, w);
}

int reart_knn_launch_pruned_pair(const KnnArgs &k1, const KnnArgs &k3, unsigned int *counters, hipStream_t st) {
    for (int j = 0; j < 2; ++j)
        if (!k1.job[j].boxes || !k1.job[j].seed || !k3.job[j].boxes || !k3.job[j].seed) return REART_ERR_INVALID_ARG;
    if (k1.items != 2 * k1.items0 || k3.items != k3.items0) return REART_ERR_INVALID_ARG;
    KnnPairArgs a;
    a.k1 = k1; a.k3 = k3;
    a.k1.sparse = a.k3.sparse = reart_prune_pick_sparse();
    const int m = (k1.items / 2 < k3.items) ? k1.items / 2 : k3.items;
    a.mixed = 3 * m;
    a.ctr = counters;
    const char *pe = getenv("REART_PERSIST");
    // persistent waves per SIMD; 0 (default) = one workgroup per item.  Measured equal at 7 per SIMD and
    // slower below: the launch is bound by its arithmetic, not by the rate at which workgroups start.
    const int wps = pe ? atoi(pe) : 0;
    if (counters && wps > 0) {
        int nwg = 256 * 4 * wps;
        if (nwg > k1.items + k3.items) nwg = k1.items + k3.items;
        hipLaunchKernelGGL(knn_pruned_pair_persistent_kernel, dim3(nwg), dim3(NN_BS), 0, st, a);
    } else {
        hipLaunchKernelGGL(knn_pruned_pair_kernel, dim3(reart_div_up(k1.items + k3.items, PR_WPB)), dim3(NN_BS * PR_WPB), 0, st, a);
    }
    REART_CHECK_LAUNCH();
    return REART_OK;
}

int reart_knn_launch_pruned(const KnnArgs &a_in, int KK, hipStream_t st) {
    KnnArgs a = a_in;
    a.sparse = reart_prune_pick_sparse();
    const int grid = reart_div_up(a.items, PR_WPB);
    for (int j = 0; j < 2; ++j)
        if (!a.job[j].boxes || !a.job[j].seed) return REART_ERR_INVALID_ARG;
    if (a.items != a.items0 && a.items != 2 * a.items0) return REART_ERR_INVALID_ARG;   // the two jobs alternate
    switch (KK) {
        case 1: hipLaunchKernelGGL((knn_pruned_kernel<1>), dim3(grid), dim3(NN_BS * PR_WPB), 0, st, a); break;
        case 3: hipLaunchKernelGGL((knn_pruned_kernel<3>), dim3(grid), dim3(NN_BS * PR_WPB), 0, st, a); break;
        default: return REART_ERR_UNSUPPORTED;
    }
    REART_CHECK_LAUNCH();
    return REART_OK;
}

// split for the pruned search: the per-item work is small and uneven, a few slices even it out.  Every slice repeats
// the item prologue and the final rescan, so fewer slices retire fewer instructions; measured with the sparse scan
// (9.79 / 9.95 / 9.87 k it/s for S = 4 / 3 / 2; S = 8: 8.7 k)
int reart_prune_pick_split(void) {
    const char *env = getenv("REART_PRUNE_SPLIT");
    const int S = env ? atoi(env) : 3;
    return S >= 1 && S <= 16 ? S : 3;
}

int reart_prune_pick_split3(void) {
    const char *env = getenv("REART_PRUNE_SPLIT3");
    if (!env) return reart_prune_pick_split();
    const int S = atoi(env);
    return S >= 1 && S <= 16 ? S : reart_prune_pick_split();
}

// ---------------------------------------------------------------------------------------------
// Stand-alone entry: warm-started exact K-NN (K = 1 or 3) through the C ABI.
// ---------------------------------------------------------------------------------------------
// merge the S slice partials of one query by the (distance, index) key; K = 3 partials are blocks of
// 8 targets (block minimum, first index) that are rescanned here with the exact key
template <int KK>
__global__ __launch_bounds__(256) void knn_warm_finish_kernel(KnnArgs a) {
    const KnnJob jb = a.job[0];
    const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= jb.P1) return;
    float kd[KK];
    int ki[KK];
#pragma unroll
    for (int k = 0; k < KK; ++k) { kd[k] = INFINITY; ki[k] = 0x7fffffff; }
    auto insert = [&](float d, int j) {
#pragma unroll
        for (int s = KK - 1; s >= 0; --s) {
            const int sp = s > 0 ? s - 1 : 0;
            const bool lp = (s > 0) && ((d < kd[sp]) | ((d == kd[sp]) & (j < ki[sp])));
            const bool lc = (d < kd[s]) | ((d == kd[s]) & (j < ki[s]));
            kd[s] = lp ? kd[sp] : (lc ? d : kd[s]);
            ki[s] = lp ? ki[sp] : (lc ? j : ki[s]);
        }
    };
    for (int s = 0; s < a.S; ++s) {
        const size_t o = (((size_t)s * a.N + b) * jb.P1 + i) * KK;
#pragma unroll
        for (int k = 0; k < KK; ++k) {
            const float d = jb.pd[o + k];
            insert(d, d < INFINITY ? jb.pi[o + k] : 0x7fffffff);
        }
    }
    if (KK > 1) {
        const float *qp = jb.q + ((size_t)b * jb.P1 + i) * 3;
        const float qx = qp[0], qy = qp[1], qz = qp[2];
        const float *tx = jb.tsoa + (size_t)b * 3 * jb.Ppad, *ty = tx + jb.Ppad, *tz = ty + jb.Ppad;
        int blk[KK];
#pragma unroll
        for (int k = 0; k < KK; ++k) { blk[k] = kd[k] < INFINITY ? ki[k] : -1; kd[k] = INFINITY; ki[k] = 0x7fffffff; }
#pragma unroll
        for (int c = 0; c < KK; ++c) {
            if (blk[c] < 0) continue;
            for (int u = 0; u < 8; ++u) {
                const int jj = blk[c] + u;
                const float d = reart_sqdist3(qx, qy, qz, tx[jj], ty[jj], tz[jj]);
                insert(d, d < INFINITY ? jj : 0x7fffffff);
            }
        }
    }
    const int n2 = jb.P2;
    const int valid = KK < n2 ? KK : n2;
#pragma unroll
    for (int k = 0; k < KK; ++k) {
        const bool ok = k < valid && ki[k] != 0x7fffffff;
        const size_t o = ((size_t)b * jb.P1 + i) * KK + k;
        jb.dists[o] = ok ? kd[k] : 0.0f;
        jb.idx[o] = ok ? (int64_t)ki[k] : (int64_t)0;
        ((int *)jb.seed)[o] = ok ? ki[k] : -1;     // warm start of the next call
    }
}

struct WarmPlan { int S, Ppad; size_t o_soa, o_box, o_pd, o_pi, total; };
static int warm_plan(int N, int P1, int P2, int K, WarmPlan *p) {
    if (N < 0 || P1 < 0 || P2 < 0 || (K != 1 && K != 3)) return REART_ERR_INVALID_ARG;
    const char *mode = getenv("REART_SEARCH");
    p->S = (mode && !strcmp(mode, "quad")) ? 1 : reart_prune_pick_split();   // quad.hip leaves one record per query
    while (p->S > 1 && reart_div_up(P2, p->S) < 64) p->S /= 2;
    p->Ppad = (int)reart_align_up((size_t)(P2 > 0 ? P2 : 1), NN_BOX);
    size_t off = 0;
    p->o_soa = off; off += reart_align_up(sizeof(float) * 3 * (size_t)N * p->Ppad, 256);
    p->o_box = off; off += reart_align_up(sizeof(float) * 8 * (size_t)N * (p->Ppad / NN_BOX), 256);
    p->o_pd = off; off += reart_align_up(sizeof(float) * (size_t)p->S * N * P1 * K, 256);
    p->o_pi = off; off += reart_align_up(sizeof(int) * (size_t)p->S * N * P1 * K, 256);
    p->total = off;
    return REART_OK;
}

extern "C" size_t reart_knn_points_warm_workspace_bytes(int N, int P1, int P2, int K) {
    WarmPlan p;
    return warm_plan(N, P1, P2, K, &p) == REART_OK ? p.total : 0;
}

extern "C" int reart_knn_points_idx_warm(const float *p1, const float *p2, int N, int P1, int P2, int K,
                                         int32_t *seed, float *dists, int64_t *idx, void *workspace,
                                         size_t workspace_bytes, void *stream) {
    WarmPlan p;
    int rc = warm_plan(N, P1, P2, K, &p);
    if (rc != REART_OK) return rc;
    if (N == 0 || P1 == 0) return REART_OK;
    if (!p1 || !p2 || !seed || !dists || !idx || P2 < 1) return REART_ERR_INVALID_ARG;
    if (!workspace || workspace_bytes < p.total) return REART_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    char *ws = (char *)workspace;
    SoaArgs sa = {};
    for (int j = 0; j < 2; ++j) {
        sa.job[j].src = p2; sa.job[j].len = nullptr; sa.job[j].dst = (float *)(ws + p.o_soa);
        sa.job[j].P = P2; sa.job[j].Ppad = p.Ppad;
    }
    rc = reart_soa_launch(sa, p.Ppad, N, 1, st);
    if (rc != REART_OK) return rc;
    rc = reart_boxes_launch((const float *)(ws + p.o_soa), N, p.Ppad, (float *)(ws + p.o_box), st);
    if (rc != REART_OK) return rc;
    KnnArgs a = {};
    a.N = N; a.S = p.S; a.K = K; a.euclidean = 0;
    KnnJob &jb = a.job[0];
    jb.q = p1; jb.tsoa = (const float *)(ws + p.o_soa); jb.boxes = (const float *)(ws + p.o_box);
    jb.seed = seed; jb.P1 = P1; jb.P2 = P2; jb.Ppad = p.Ppad; jb.L = 0; jb.nqg = reart_div_up(P1, NN_BS);
    jb.pd = (float *)(ws + p.o_pd); jb.pi = (int *)(ws + p.o_pi); jb.dists = dists; jb.idx = idx;
    a.job[1] = jb;
    const char *mode = getenv("REART_SEARCH");
    const bool quad = mode && !strcmp(mode, "quad");
    a.items0 = quad ? N * reart_div_up(P1, 16) : N * jb.nqg * p.S; a.items = a.items0;
    rc = quad ? reart_knn_launch_quad(a, K, st) : reart_knn_launch_pruned(a, K, st);
    if (rc != REART_OK) return rc;
    const dim3 fg(reart_div_up(P1, 256), N);
    if (K == 1) hipLaunchKernelGGL((knn_warm_finish_kernel<1>), fg, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((knn_warm_finish_kernel<3>), fg, dim3(256), 0, st, a);
    REART_CHECK_LAUNCH();
    return REART_OK;
}
