// reart_amd/csrc/prune.hip -- EXACT nearest-neighbour search with bounding-box pruning and a
// warm start, for the relaxation loop (reference run_robot.py:154-221: every iteration repeats the
// Chamfer search of utils/chamfer.py:78-94 and the k=3 search of utils/flow_utils.py:158 on clouds
// that moved a little since the previous iteration).
//
// Same answer, bit for bit, as the brute-force kernels of knn.hip (same distance expression,
// strict '<', ties -> lowest index); what changes is how many targets are looked at.
//
//   * Clouds are stored in the leaf order of a balanced k-d tree (host side, relax.kd_order), so 16
//     consecutive targets form a compact box and the 64 queries of a wave are neighbours.
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
// current K-th candidate has d <= thr and therefore sits in a box that is scanned.
//
// Launch shape.  One WORKGROUP per (job, batch b, query group g of 64): its S waves take the boxes
// round-robin (slice s = boxes base + lane*S + s), keep their candidates in registers, meet once in LDS,
// and wave 0 merges them by the full (distance, index) key, does the ONE exact rescan of the winning half
// box and writes ONE record per query -- no per-slice partial lists in memory, nothing for the consumer to
// merge.  The K = 1 record (int32 index) is also the next iteration's warm-start seed, in place.
// Launch positions are dealt to the XCDs round-robin (workgroup L runs on XCD L % 8 and takes every 8th (batch, group)
// pair: every XCD sees every frame, which balances the chip whatever the frames cost; a contiguous run of frames per
// XCD -- `interleave = 0`, tune_xcd -- was measured slower: frames differ 10x in work), heaviest pairs first (counting
// sort of last iteration's work counts, step.hip).
//
//   * Neighbour seeds (round 3).  A lane's own seed is a poor bound exactly when it matters: every iteration a fifth of
//     the points re-sample their part label and jump (queries of the x -> y search, targets of the y -> x search), their
//     old neighbour is then far away, and those lanes -- 23 % of them -- used to produce 78 % of all (query, box)
//     pairs.  But the 16 lanes of a row are neighbours on the canonical shape, and the lanes that landed next to a
//     jumped point hold seeds next to ITS new neighbour.  So the lanes of a row park their seed targets in LDS and
//     every lane takes  thr = min over the row's lanes l' of  max_k d(q, t[seed_{l',k}])  -- any K distinct targets
//     bound the K-th neighbour distance, whoever found them.  Measured on dumped states (tools/sim_filter.py):
//     (query, box) pairs per wave 675 -> 194 (x -> y), 1219 -> 508 (y -> x), 983 -> 305 (flow); boxes some lane needs
//     76 -> 34; coarse survivors 112 -> 70.
#include "common.h"
#include "internal.h"
#include <math.h>
#include <string.h>

typedef float f2 __attribute__((ext_vector_type(2)));
#define PR_SMAX 4       // most waves (box slices) per search workgroup
// (a box's targets are fetched when it is scanned: prefetching every coarse survivor eight at a time was measured -- more
// instructions, loads for the 60 % of the boxes the precise test rejects, no gain; DESIGN_HISTORY.md)
#define PR_QCAP 640     // words of a wave's queue space: the first 384 hold box bounds, 256 (query, box) entries (768: one
                        // workgroup less per compute unit, 32.9 vs 32.3 us; 512: drains too small and too many, 33.7 us)
#define PR_WPE 6         // waves per SIMD the compiler must allow for (register budget: 85 -> 80 VGPRs; the kernel sits at the
                         // edge -- 78 under this bound, 82 without it, no spills either way -- and the sixth wave is worth 7 %)
// LDS per wave: staged box of a dense scan [64 floats] | the wave's queries [64][4] | result slots [3][64] u64 | queue
// The precise filter reads a surviving box's bounds back from LDS (broadcast ds_read, the LDS pipe is idle) instead of twelve
// v_readlane per box pair on the VALU, which bounds this kernel: 45.0 -> 43.6 us per launch.  The 1.5 KB come out of the
// queue's space (768 -> 384 entries); with LDS of their own a workgroup less fits a compute unit: 48.6 us.
#define PR_LDS_WAVE_BYTES (64 * 4 + 64 * 16 + 3 * 64 * 8 + PR_QCAP * 4)

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
#ifdef REART_PRUNE_PHASE   // diagnostic build only (tools/phase_prof.py; make -C reart_amd/csrc phase)
// phase clocks of the search waves (s_memtime ticks summed over waves): 0 prologue (seeds, group summaries), 1 coarse
// filter, 2 precise tests, 3 dense scans, 4 queue appends, 5 queue drains, 6 barrier wait + merge + rescan, 7 waves
__device__ unsigned long long g_prune_phase[8];
extern "C" int reart_debug_prune_phase(unsigned long long *out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prune_phase), sizeof(g_prune_phase)) != hipSuccess) return REART_ERR_LAUNCH;
    if (reset) { unsigned long long z[8] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_prune_phase), z, sizeof(z)); }
    return REART_OK;
}
#define PH_DECL unsigned long long ph_t = __builtin_amdgcn_s_memtime(), ph_acc[7] = {0, 0, 0, 0, 0, 0, 0}
#define PH(k) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); ph_acc[k] += n_ - ph_t; ph_t = n_; } while (0)
#define PH_FLUSH(from, to) do { if ((threadIdx.x & 63) == 0) for (int k_ = from; k_ <= to; ++k_) atomicAdd(&g_prune_phase[k_], ph_acc[k_]); } while (0)
#else
#define PH_DECL do { } while (0)
#define PH(k) do { } while (0)
#define PH_FLUSH(from, to) do { } while (0)
#endif

#define PR_WORK(w) ((unsigned)(w) << 20)   // the work count lives in bits 20..31 of a wave's statistics counter, the pairs below

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

// One WAVE of a search workgroup: the 64 queries of group g of batch b against the boxes of slice s (boxes
// base + lane * S + s of every round).  Leaves, per lane (= query): KK = 1 the smallest distance found and the
// half box (8 targets) that holds it; KK = 3 the three best 8-target blocks by (minimum, first index).
// `pairs` counts the query-target distance evaluations the wave executed (lanes x targets, dense and sparse forms).
// LDSV: `cloud` / `boxes_p` point at a copy of the batch's target cloud (SoA rows of `cstride` floats) and of its boxes in
// LDS (knn_cloud_kernel); otherwise at the global images.  QCAP: entries of the wave's queue.
// share: neighbour seeds (parked in the queue's space, which is free until the first coarse round: s_bb | s_q when BOXL,
// s_q otherwise; skipped when that space is too small for KK x 3 x 64 floats).
template <int KK, bool LDSV, int QCAP, bool BOXL>
__device__ __forceinline__ void knn_pruned_wave(const KnnJob &jb, const int S, const int sparse, const int b, const int g,
                                                const int s, const float *cloud, const int cstride, const float *boxes_p,
                                                float *s_tg, float *s_qc, unsigned int *s_q, float *s_bb, const int share, float *s_xthr,
                                                unsigned long long *s_key, float &qx_o, float &qy_o, float &qz_o,
                                                float (&bm)[KK], int (&bb)[KK], int &work_o, unsigned int &pairs_o) {
    struct { int S, sparse; } a = {S, sparse};
    const int lane = threadIdx.x & 63;
    PH_DECL;

    const int i = g * NN_BS + lane;
    const int ic = i < jb.P1 ? i : jb.P1 - 1;
    const int qb = jb.qmap ? jb.qmap[b] : b;
    const float *qp = (qb < 0 ? jb.q_alt : jb.q + (size_t)qb * jb.P1 * 3) + (size_t)ic * 3;
    const float qx = qp[0], qy = qp[1], qz = qp[2];
    const f2 qx2 = {qx, qx}, qy2 = {qy, qy}, qz2 = {qz, qz};

    const float *tx = cloud;
    const float *ty = tx + cstride;
    const float *tz = ty + cstride;
    const int n2 = jb.tlen ? jb.tlen[b] : jb.P2;
    const int nbox = (n2 + NN_BOX - 1) / NN_BOX;
    const float *bx = boxes_p;

    // ---- warm start
    float thr = 0.f;
    unsigned int wp = 0u;             // statistics counter, see PR_WORK below
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
        // scratch for the neighbour seeds: the result slots (s_key, 384 floats, initialised after this block) and the queue's
        // space behind them (box bounds | queue) are contiguous and idle until the first coarse round
        constexpr bool SCR_JOINT = BOXL;                         // s_bb | s_q contiguous
        constexpr bool SCR_FITS = (384 + QCAP + (SCR_JOINT ? 384 : 0)) >= KK * 3 * 64;
        if (SCR_FITS && share) {
            // ---- neighbour seeds: the K seed targets of every lane of this row of 16 (the lane's own among them), two lanes
            // per packed step.  A lane without usable seeds parks +inf (its candidates bound nothing -- but it still takes its
            // neighbours' bound); LDS operations of a wave complete in order, and every lane reads only its own row.
            float *s_scr = (float *)s_key;
#pragma unroll
            for (int k = 0; k < KK; ++k) {
                const int j = ok ? sj[k] : 0;
                s_scr[(3 * k + 0) * 64 + lane] = ok ? tx[j] : INFINITY;
                s_scr[(3 * k + 1) * 64 + lane] = ok ? ty[j] : INFINITY;
                s_scr[(3 * k + 2) * 64 + lane] = ok ? tz[j] : INFINITY;
            }
            thr = INFINITY;
            const float *row = s_scr + (lane & 48);
            // s_xthr (the workgroup's exchange array [S][64]): the S waves of the workgroup hold the same 64 queries, so each
            // evaluates every S-th candidate pair and the partial bounds meet after ONE barrier (at the start of the waves'
            // lives, where they still run side by side); without it every wave evaluates all 16 candidates
            const bool split = s_xthr != nullptr && a.S > 1;
            const int c0 = split ? 2 * s : 0, cstep = split ? 2 * a.S : 2;
            wp = 64u * KK * 2u * (unsigned)((16 - c0 + cstep - 1) / cstep);   // the seed candidates this wave evaluates
#pragma unroll 1      // rolled: unrolled, the 72 LDS loads of a K = 3 item in flight cost 25 VGPRs, i.e. two resident waves per SIMD
            for (int c = c0; c < 16; c += cstep) {
                f2 far = {0.f, 0.f};
#pragma unroll
                for (int k = 0; k < KK; ++k) {
                    const f2 dx = qx2 - *(const f2 *)(row + (3 * k + 0) * 64 + c);
                    const f2 dy = qy2 - *(const f2 *)(row + (3 * k + 1) * 64 + c);
                    const f2 dz = qz2 - *(const f2 *)(row + (3 * k + 2) * 64 + c);
                    const f2 d = (dx * dx + dy * dy) + dz * dz;
                    far = f2{fmaxf(far.x, d.x), fmaxf(far.y, d.y)};
                }
                thr = fminf(thr, fminf(far.x, far.y));
            }
            if (split) {
                s_xthr[64 * s + lane] = thr;
                __syncthreads();                     // every live wave of the workgroup runs this prologue (waves >= S have left)
                for (int t = 0; t < a.S; ++t) thr = fminf(thr, s_xthr[64 * t + lane]);
            }
            // a NaN query: fmaxf / fminf drop the NaN distances (fmaxf(0, NaN) = 0), so the bound would come out 0, not NaN --
            // test the query itself and prune nothing for it (its lower bounds are NaN and fail every compare: index -1)
            if (!(qx == qx && qy == qy && qz == qz)) thr = INFINITY;
        } else {
#pragma unroll
            for (int k = 0; k < KK; ++k) {
                const int j = ok ? sj[k] : 0;
                thr = fmaxf(thr, reart_sqdist3(qx, qy, qz, tx[j], ty[j], tz[j]));
            }
            wp = 64u * KK;
            if (!ok || !(thr >= 0.f)) thr = INFINITY;   // unusable seeds / NaN: no pruning for this lane
        }
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

#pragma unroll
    for (int k = 0; k < KK; ++k) { bm[k] = INFINITY; bb[k] = -1; }
    const float *tx_g = tx;
    // ONE uniform counter for both statistics (a scalar register less, one s_add per event): bits 0..19 the distance
    // evaluations (< 2^20 per wave), bits 20..31 the work count that orders the next launch (a heuristic: overflow is harmless)
    wp |= PR_WORK(16);                                                   // prologue ~ 16 tests

    const int queue_max = a.sparse;   // boxes needed by at most this many queries go through the (query, box) queue (0: off)
    int qcnt = 0;                     // entries in the wave's queue (uniform)
    // the wave's queries, parked once for the drain steps (a lane then evaluates ANOTHER lane's query)
    ((float4 *)s_qc)[lane] = make_float4(qx, qy, qz, 0.f);
    if (KK == 1) s_key[lane] = ~0ull;
    else { s_key[lane] = ~0ull; s_key[64 + lane] = ~0ull; s_key[128 + lane] = ~0ull; }
    // Drain: 64 entries per step, one per lane.  The lane's 16 targets come straight from the SoA rows (three 64-byte
    // lines per lane); the distance expression is the contract's.  K = 1: the (minimum, half box) of the entry meets the
    // query's other entries in an LDS atomic minimum on the 64-bit key (distance bits, block start) -- distances are
    // non-negative, so the integer order is the (distance, index) order and the result does not depend on the order of
    // arrival.  K = 3: both half-box minima go through three such minima in a row, the loser of each level moving on to
    // the next: the three slots end as the three smallest keys, again whatever the order.  Afterwards every lane folds
    // its query's slots into its registers (and its bound) and clears them.
    auto drain_queue = [&]() {
        for (int e0 = 0; e0 < qcnt; e0 += 64) {
            wp += PR_WORK(6);
            const int e = e0 + lane;
            const bool valid = e < qcnt;
            const unsigned ent = s_q[valid ? e : e0];
            const int ql = (int)(ent & 63u), j0 = (int)(ent >> 6) * NN_BOX;
            wp += (unsigned)__builtin_popcountll(__ballot(valid)) * NN_BOX;
            const float4 qc = ((const float4 *)s_qc)[ql];
            // rows addressed as (wave-uniform base) + (32-bit lane offset): the scalar-base form of global_load keeps three
            // 64-bit lane pointers (and their adds) out of the VGPRs
            const float *rx = tx_g, *ry = tx_g + cstride, *rz = tx_g + 2 * (size_t)cstride;
            const f2 cx = {qc.x, qc.x}, cy = {qc.y, qc.y}, cz = {qc.z, qc.z};
            float mh[2] = {INFINITY, INFINITY};
            // one half box (8 targets: six 16-byte loads) at a time: the twelve loads of a whole box in flight cost 24 more
            // VGPRs than the rest of the kernel needs, i.e. a resident wave per SIMD
#pragma unroll 1
            for (int hf = 0; hf < 2; ++hf) {
                float4 X[2], Y[2], Z[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const unsigned ob = 4u * ((unsigned)j0 + 8u * hf + 4u * u);          // BYTE offset: zext(ob) is the address form
                    X[u] = *(const float4 *)((const char *)rx + ob);
                    Y[u] = *(const float4 *)((const char *)ry + ob);
                    Z[u] = *(const float4 *)((const char *)rz + ob);
                }
                float m = INFINITY;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const f2 dxa = cx - f2{X[u].x, X[u].y}, dya = cy - f2{Y[u].x, Y[u].y}, dza = cz - f2{Z[u].x, Z[u].y};
                    const f2 dxb = cx - f2{X[u].z, X[u].w}, dyb = cy - f2{Y[u].z, Y[u].w}, dzb = cz - f2{Z[u].z, Z[u].w};
                    const f2 da = (dxa * dxa + dya * dya) + dza * dza;
                    const f2 db = (dxb * dxb + dyb * dyb) + dzb * dzb;
                    m = fminf(fminf(m, fminf(da.x, da.y)), fminf(db.x, db.y));
                }
                if (hf == 0) mh[0] = m; else mh[1] = m;
            }
            if (KK == 1) {
                const float m = fminf(mh[0], mh[1]);
                const unsigned long long key = ((unsigned long long)__float_as_uint(m) << 32) | (unsigned)(j0 + (mh[1] < mh[0] ? 8 : 0));
                if (valid) atomicMin(&s_key[ql], key);
            } else if (valid) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    unsigned long long key = ((unsigned long long)__float_as_uint(mh[h]) << 32) | (unsigned)(j0 + 8 * h);
#pragma unroll
                    for (int lv = 0; lv < 3; ++lv) {
                        const unsigned long long old = atomicMin(&s_key[64 * lv + ql], key);
                        key = old > key ? old : key;
                    }
                }
            }
        }
        qcnt = 0;
        // fold the slots of this lane's own query into its registers
        if (KK == 1) {
            const unsigned long long k = s_key[lane];
            const float m = __uint_as_float((unsigned)(k >> 32));
            const int blk = (int)(unsigned)k;
            if (k != ~0ull && ((m < bm[0]) || (m == bm[0] && blk < bb[0]))) { bm[0] = m; bb[0] = blk; }
            thr = fminf(thr, bm[0]);
            s_key[lane] = ~0ull;
        } else {
            float kd[3];
            int ki[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) { kd[c] = bm[c]; ki[c] = bb[c] >= 0 ? bb[c] : 0x7fffffff; }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const unsigned long long k = s_key[64 * c + lane];
                s_key[64 * c + lane] = ~0ull;
                if (k != ~0ull) reart_top3_insert(kd, ki, __uint_as_float((unsigned)(k >> 32)), (int)(unsigned)k);
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) { bm[c] = kd[c]; bb[c] = ki[c] == 0x7fffffff ? -1 : ki[c]; }
            thr = fminf(thr, bm[KK - 1]);
        }
    };
    const int per = 64 * a.S;
    PH(0);
    for (int base = 0; base < nbox; base += per) {
        // ---- coarse filter: lane l looks at box base + l*S + s
        const int bid = base + lane * a.S + s;
        bool pass = false;
        float lo0 = INFINITY, lo1 = INFINITY, lo2 = INFINITY, hi0 = INFINITY, hi1 = INFINITY, hi2 = INFINITY;
        if (bid < nbox) {
            const float4 A = *(const float4 *)(bx + (size_t)bid * 8);
            const float4 Bv = *(const float4 *)(bx + (size_t)bid * 8 + 4);
            lo0 = A.x; lo1 = A.y; lo2 = A.z; hi0 = A.w; hi1 = Bv.x; hi2 = Bv.y;
            if (BOXL) {        // the precise filter reads a surviving box's bounds back as LDS broadcasts (not VALU readlanes)
                float *o = s_bb + 6 * lane;
                o[0] = lo0; o[1] = lo1; o[2] = lo2; o[3] = hi0; o[4] = hi1; o[5] = hi2;
            }
            // two query sub-groups per packed instruction (same operation order per sub-group as box_lb)
            const f2 bl0 = {lo0, lo0}, bl1 = {lo1, lo1}, bl2 = {lo2, lo2}, bh0 = {hi0, hi0}, bh1 = {hi1, hi1}, bh2 = {hi2, hi2};
#pragma unroll
            for (int q = 0; q < 4; q += 2) {
                const f2 ql0 = {G[q][0], G[q + 1][0]}, ql1 = {G[q][1], G[q + 1][1]}, ql2 = {G[q][2], G[q + 1][2]};
                const f2 qh0 = {G[q][3], G[q + 1][3]}, qh1 = {G[q][4], G[q + 1][4]}, qh2 = {G[q][5], G[q + 1][5]};
                const f2 a0 = bl0 - qh0, c0 = ql0 - bh0, a1 = bl1 - qh1, c1 = ql1 - bh1, a2 = bl2 - qh2, c2 = ql2 - bh2;
                const f2 e0 = {fmaxf(fmaxf(a0.x, c0.x), 0.f), fmaxf(fmaxf(a0.y, c0.y), 0.f)};
                const f2 e1 = {fmaxf(fmaxf(a1.x, c1.x), 0.f), fmaxf(fmaxf(a1.y, c1.y), 0.f)};
                const f2 e2 = {fmaxf(fmaxf(a2.x, c2.x), 0.f), fmaxf(fmaxf(a2.y, c2.y), 0.f)};
                const f2 lb = (e0 * e0 + e1 * e1) + e2 * e2;
                pass = pass || (lb.x <= G[q][6]) || (lb.y <= G[q + 1][6]);
            }
        }
        unsigned long long mask = __ballot(pass);
        PH(1);
        PRUNE_STAT(KK == 1 ? 0 : 3, 0);                                  // coarse rounds (K = 1)
        if (KK == 1) PRUNE_STAT(1, __builtin_popcountll(mask));          // boxes passing the coarse filter
        if (KK == 1) PRUNE_STAT(3, __builtin_popcountll(__ballot(thr == INFINITY)));  // lanes without a bound
        // scan of one box: the brute-force inner loop of knn.hip on its 16 targets
        auto scan_box = [&](const int bit) {
            wp += PR_WORK(5) | (64u * NN_BOX);                           // a scan costs about five tests
            if (KK == 1) PRUNE_STAT(2, 1);                               // boxes scanned
            const int j0 = (base + bit * a.S + s) * NN_BOX;
            // the box's 16 targets are staged in the wave's LDS slot (x[16] | y[16] | z[16]); all lanes read
            // the same addresses (broadcast) -- the operands of the packed ops are VGPR pairs
            if (!LDSV) {
                const int l = lane < 48 ? lane : 47;
                const float v = tx_g[(size_t)(l >> 4) * cstride + j0 + (l & 15)];
                if (lane < 48) s_tg[lane] = v;
            }
            // LDSV: the cloud itself is in LDS -- the broadcast reads go straight to it
            const float *tx = LDSV ? tx_g : s_tg - j0;
            const float *ty = tx + (LDSV ? cstride : 16), *tz = ty + (LDSV ? cstride : 16);
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
        // (query, box) work queue.  Most scanned boxes are needed by a FEW of the wave's 64 queries (one to three is
        // typical: noisy points with large bounds drive the per-wave union), so walking a box with all 64 lanes for
        // their sake wastes the wave.  Such a box is not scanned here: every needing lane appends (box, lane) to the
        // wave's queue in LDS -- its rank among the needers (mbcnt of the ballot) is its slot -- and the queue is
        // drained 64 entries at a time with ONE entry per lane (drain_queue below): every lane then evaluates one
        // query against the 16 targets of one box, whatever mix of queries and boxes the entries hold.
        auto enqueue_box = [&](const int bit, const unsigned long long need, const bool nd, const int nneed) {
            wp += PR_WORK(1);
            if (KK == 1) PRUNE_STAT(2, 1);
            const int bid = base + bit * a.S + s;
            const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(need >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)need, 0u));
            if (nd) s_q[qcnt + rank] = ((unsigned)bid << 6) | (unsigned)lane;
            qcnt += nneed;
        };
        while (mask) {
            // (Targets used to come through scalar loads, s_load_dwordx16 x 3 per scanned box: one dependent scalar-cache
            // miss chain per scan; measured 95 -> 80 us for the launch with the vector path of scan_box alone.)
            unsigned long long cm = mask;                                // every surviving box of the round
            mask = 0ull;
            // ---- precise filter with the lane's current bound, two boxes per step in packed fp32 (same
            // operation order per box as box_lb; the second box is judged after the first one's scan)
            while (cm) {
            const int bA = __builtin_ctzll(cm);
            cm &= cm - 1;
            const bool hasB = cm != 0;
            const int bB = hasB ? __builtin_ctzll(cm) : bA;
            if (hasB) cm &= cm - 1;
            f2 L0, L1, L2, H0, H1, H2;
            if (BOXL) {
                const float *pA = s_bb + 6 * bA, *pB = s_bb + 6 * bB;
                L0 = f2{pA[0], pB[0]}; L1 = f2{pA[1], pB[1]}; L2 = f2{pA[2], pB[2]};
                H0 = f2{pA[3], pB[3]}; H1 = f2{pA[4], pB[4]}; H2 = f2{pA[5], pB[5]};
            } else {
                L0 = f2{rl(lo0, bA), rl(lo0, bB)}; L1 = f2{rl(lo1, bA), rl(lo1, bB)}; L2 = f2{rl(lo2, bA), rl(lo2, bB)};
                H0 = f2{rl(hi0, bA), rl(hi0, bB)}; H1 = f2{rl(hi1, bA), rl(hi1, bB)}; H2 = f2{rl(hi2, bA), rl(hi2, bB)};
            }
            const f2 a0 = L0 - qx2, c0 = qx2 - H0, a1 = L1 - qy2, c1 = qy2 - H1, a2 = L2 - qz2, c2 = qz2 - H2;
            const f2 e0 = {fmaxf(fmaxf(a0.x, c0.x), 0.f), fmaxf(fmaxf(a0.y, c0.y), 0.f)};
            const f2 e1 = {fmaxf(fmaxf(a1.x, c1.x), 0.f), fmaxf(fmaxf(a1.y, c1.y), 0.f)};
            const f2 e2 = {fmaxf(fmaxf(a2.x, c2.x), 0.f), fmaxf(fmaxf(a2.y, c2.y), 0.f)};
            const f2 lb = (e0 * e0 + e1 * e1) + e2 * e2;
            wp += PR_WORK(2);
            PH(2);
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
                    if (nneed <= queue_max) {
                        enqueue_box(h ? bB : bA, need, nd, nneed);
                        PH(4);
                        if (qcnt > QCAP - 64) { drain_queue(); PH(5); }   // uniform: qcnt is a wave-wide count
                    } else {
                        scan_box(h ? bB : bA);
                        PH(3);
                    }
                }
            }
            }
        }
    }

    PH(2);
    if (qcnt > 0) drain_queue();
    PH(5);
    PH_FLUSH(0, 5);
    qx_o = qx; qy_o = qy; qz_o = qz;
    work_o = (int)(wp >> 20); pairs_o = wp & 0xFFFFFu;
}


// ------------------------------------------------------------------------------------------------------------
// The search launch: up to two K = 1 jobs (the Chamfer directions) and one K = 3 job (the flow search);
// SearchArgs is declared in internal.h.
template <bool BATCH>
__global__ __launch_bounds__(64 * PR_SMAX, PR_WPE) void knn_group_kernel(Batched<SearchArgs> ab) {
    // instance of a batch (gridDim.x is a multiple of 8: blockIdx.x & 7 is still the XCD); a single one reads at a fixed offset
    const SearchArgs &a = ab.a[BATCH ? blockIdx.y : 0];
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];      // blockDim.x / 64 x PR_LDS_WAVE_BYTES
    // where wave t parks its candidates for the merge: its own result slots (s_key, 3 x 64 x 8 bytes), dead once its search
    // is over -- as static arrays they cost 6 KB per workgroup, i.e. the sixth resident workgroup of a compute unit
#define PARK_M(t) ((float *)(s_dyn + (size_t)(t) * PR_LDS_WAVE_BYTES + 64 * 4 + 64 * 16))
#define PARK_B(t) ((int *)(s_dyn + (size_t)(t) * PR_LDS_WAVE_BYTES + 64 * 4 + 64 * 16 + 3 * 64 * 4))
    __shared__ unsigned int s_wk[PR_SMAX][2];
    __shared__ float s_xthr[PR_SMAX][64];   // the waves' partial neighbour-seed bounds (written once, read after one barrier)
    __shared__ unsigned int s_ticket;
    __shared__ unsigned long long s_t0;
    if (threadIdx.x == 0) { s_ticket = 0u; s_t0 = a.prof ? wall_clock64() : 0ull; }
    __syncthreads();                        // at the start, where all waves of the workgroup arrive together
    const int L = blockIdx.x, x = L & 7, r = L >> 3;
    const int s = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned long long t0 = s_t0;
    // XCD-local item r: two K = 1 items alternate with one K = 3 item while both kinds last
    const int items1 = a.n1 * a.per, items3 = a.n3 * a.per;
    const int mixed = (a.n1 == 2 && a.n3) ? 3 * a.per : 0;
    int kind, idx;
    if (r < mixed) {
        const int q = r / 3, rr = r - 3 * q;
        kind = rr < 2 ? 1 : 3;
        idx = rr < 2 ? 2 * q + rr : q;
    } else {
        const int d1 = mixed ? 2 * a.per : 0, d3 = mixed ? a.per : 0;
        const int v = r - mixed, left1 = items1 - d1;
        kind = v < left1 ? 1 : 3;
        idx = v < left1 ? d1 + v : d3 + (v - left1);
    }
    const int j = kind == 1 ? (a.n1 == 2 ? (idx & 1) : 0) : 0;
    const int p = kind == 1 ? (a.n1 == 2 ? (idx >> 1) : idx) : idx;
    const int pos = a.interleave ? p * 8 + x : x * a.per + p;   // launch position: XCD x owns a contiguous chunk, or every 8th
    const bool live = p < a.per && pos < a.G && (kind == 1 ? idx < items1 : idx < items3);
    if (!live) {                                                        // padding of the last chunk (uniform per workgroup)
        if (a.prof && threadIdx.x == 0) { a.prof[2 * (size_t)L] = t0; a.prof[2 * (size_t)L + 1] = t0; a.prof_pairs[L] = 0u; }
        return;
    }
    const KnnJob &jb = kind == 1 ? a.k1[j] : a.k3;
    const int S = kind == 1 ? a.S1 : a.S3;
    if (s >= S) return;                                                 // the other kind of item uses more waves
    const int grp = jb.border ? jb.border[pos] : pos;
    const int b = grp / jb.nqg, g = grp - b * jb.nqg;
    float qx = 0.f, qy = 0.f, qz = 0.f;
    float bm[3] = {INFINITY, INFINITY, INFINITY};
    int bb[3] = {-1, -1, -1};
    int work = 0;
    unsigned int pairs = 0u;
    {
        unsigned char *wl = s_dyn + (size_t)s * PR_LDS_WAVE_BYTES;
        float *s_tg = (float *)wl;
        float *s_qc = (float *)(wl + 64 * 4);
        unsigned long long *s_key = (unsigned long long *)(wl + 64 * 4 + 64 * 16);
        unsigned int *s_q = (unsigned int *)(wl + 64 * 4 + 64 * 16 + 3 * 64 * 8);
        static_assert(384 + PR_QCAP >= 3 * 3 * 64, "neighbour seeds of a K = 3 item: 9 x 64 floats in the result slots + the queue's space");
        static_assert(PR_QCAP >= 384 + 128, "box bounds of a coarse round + at least two drain steps of queue");
        float *s_bb = (float *)s_q;          // the bounds of a coarse round's 64 boxes take the first 384 entries of the queue's space
        s_q += 384;
        constexpr int QC = PR_QCAP - 384;
        const float *cloud = jb.tsoa + (size_t)b * 3 * jb.Ppad;
        const float *boxes_p = jb.boxes + (size_t)b * (jb.Ppad / NN_BOX) * 8;
        if (kind == 1) {
            float m1[1]; int b1[1];
            knn_pruned_wave<1, false, QC, true>(jb, S, a.sparse, b, g, s, cloud, jb.Ppad, boxes_p, s_tg, s_qc, s_q, s_bb, a.share, a.share > 1 ? &s_xthr[0][0] : nullptr, s_key,
                                               qx, qy, qz, m1, b1, work, pairs);
            bm[0] = m1[0]; bb[0] = b1[0];
        } else {
            knn_pruned_wave<3, false, QC, true>(jb, S, a.sparse, b, g, s, cloud, jb.Ppad, boxes_p, s_tg, s_qc, s_q, s_bb, a.share, a.share > 1 ? &s_xthr[0][0] : nullptr, s_key,
                                               qx, qy, qz, bm, bb, work, pairs);
        }
    }
    // ---- the LAST wave to finish merges (no barrier: a wave that is done leaves at once and frees its slot; measured with
    // a barrier here, 17 % of a wave's life was waiting for its slowest sibling).  Every wave parks its candidates, then
    // takes a ticket; LDS operations of a wave complete in order, so the ticket holder S-1 sees all of them.
    if (S > 1) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { PARK_M(s)[k * 64 + lane] = bm[k]; PARK_B(s)[k * 64 + lane] = bb[k]; }
        if (lane == 0) { s_wk[s][0] = (unsigned int)work; s_wk[s][1] = pairs; }
        __threadfence_block();
        int ticket = 0;
        if (lane == 0) ticket = (int)atomicAdd(&s_ticket, 1u);
        ticket = __builtin_amdgcn_readfirstlane(ticket);
#ifdef REART_PRUNE_PHASE
        if (lane == 0) atomicAdd(&g_prune_phase[7], 1ull);
#endif
        if (ticket != S - 1) return;
        __threadfence_block();
        work = 0; pairs = 0u;
        for (int t = 0; t < S; ++t) { work += (int)s_wk[t][0]; pairs += s_wk[t][1]; }
#pragma unroll
        for (int k = 0; k < 3; ++k) { bm[k] = PARK_M(0)[k * 64 + lane]; bb[k] = PARK_B(0)[k * 64 + lane]; }
    }
    // ---- merge the slices' candidates by (minimum, first index): distinct slices hold distinct boxes, so the lower block
    // start is the lower index
    const int i = g * NN_BS + lane;
    const float *tx = jb.tsoa + (size_t)b * 3 * jb.Ppad, *ty = tx + jb.Ppad, *tz = ty + jb.Ppad;
    if (kind == 1) {
        float m = bm[0];
        int blk = bb[0];
        for (int t = 1; t < S; ++t) {
            const float mt = PARK_M(t)[lane];
            const int bt = PARK_B(t)[lane];
            const bool take = bt >= 0 && ((mt < m) || (mt == m && (blk < 0 || bt < blk)));
            m = take ? mt : m; blk = take ? bt : blk;
        }
        // exact (lowest) index inside the winning half box: ONE rescan per query
        int bi = 0x7fffffff;
        if (blk >= 0) {
#pragma unroll
            for (int u = NN_BOX / 2 - 1; u >= 0; --u) {
                const float d = reart_sqdist3(qx, qy, qz, tx[blk + u], ty[blk + u], tz[blk + u]);
                if (d == m) bi = blk + u;
            }
        }
        pairs += 64u * (NN_BOX / 2);
        if (i < jb.P1) {
            const size_t o = (size_t)b * jb.P1 + i;
            jb.pd[o] = m; jb.pi[o] = bi;
        }
    } else {
        float kd[3] = {INFINITY, INFINITY, INFINITY};
        int ki[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff};
#pragma unroll
        for (int k = 0; k < 3; ++k) reart_top3_insert(kd, ki, bm[k], bb[k] >= 0 ? bb[k] : 0x7fffffff);
        for (int t = 1; t < S; ++t) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int bt = PARK_B(t)[k * 64 + lane];
                reart_top3_insert(kd, ki, PARK_M(t)[k * 64 + lane], bt >= 0 ? bt : 0x7fffffff);
            }
        }
        if (i < jb.P1) {
            const size_t o = ((size_t)b * jb.P1 + i) * 3;
#pragma unroll
            for (int k = 0; k < 3; ++k) { jb.pd[o + k] = kd[k]; jb.pi[o + k] = kd[k] < INFINITY ? ki[k] : -1; }
        }
    }
    if (lane == 0) {
        if (jb.cost) jb.cost[pos] = (unsigned int)work;
        if (a.prof) {
            a.prof[2 * (size_t)L] = t0; a.prof[2 * (size_t)L + 1] = wall_clock64();
            a.prof_pairs[L] = pairs;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// Cloud-resident form of the same search.  Measured on the kernel above (tools/phase_prof.py): a search wave is a CHAIN
// of dependent memory round trips -- seeds -> seed targets -> boxes -> (per scanned box / per drain step) targets ->
// rescan -- and spends its life waiting for them, not computing.  A target cloud of the loop is 48 KB (4096 points,
// SoA) + 8 KB of boxes: it fits in LDS.  Here one workgroup of PC_WAVES waves copies the cloud and the boxes of ITS
// (job, batch) into LDS once (coalesced 16-byte loads); its waves are PC_WAVES / S query groups x S box slices, every
// target / box read is served by LDS, the slices of a group meet in LDS as 64-bit (distance, block) keys and the
// slice-0 wave merges them, rescans once and writes the record.  What is left of the chain in global memory: the
// query, its seeds, and the record written at the end.
// Falls back to knn_group_kernel when a cloud does not fit (reart_search_launch decides).
#define PC_WAVES 16
#define PC_QCAP 256
#define PC_WAVE_BYTES (64 * 16 + 3 * 64 * 8 + PC_QCAP * 4)      // the wave's queries | result slots | queue
__global__ __launch_bounds__(64 * PC_WAVES) void knn_cloud_kernel(SearchArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    const int L = blockIdx.x, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned long long t0 = 0ull;
    if (a.prof && threadIdx.x == 0) t0 = wall_clock64();
    // workgroup -> (job, batch, block of PC_WAVES query groups); jobs innermost so that neighbours differ in kind
    const int njobs = a.n1 + a.n3;
    const int jsel = L % njobs;
    const int rest = L / njobs;
    const int B = a.G / a.k_nqg;
    const int b = rest % B, blk = rest / B;
    const int kind = jsel < a.n1 ? 1 : 3;
    const KnnJob &jb = kind == 1 ? a.k1[jsel] : a.k3;
    const int S = a.cloud_slices, gpw = PC_WAVES / S;             // slices per group, groups per workgroup
    // ---- stage the batch's target cloud and boxes
    const int Ppad = jb.Ppad, nboxp = Ppad / NN_BOX;
    float *cl = (float *)s_dyn;                                   // [3][Ppad]
    float *bxs = cl + 3 * (size_t)Ppad;                           // [nboxp][8]
    {
        const float4 *src = (const float4 *)(jb.tsoa + (size_t)b * 3 * Ppad);
        float4 *dst = (float4 *)cl;
        const int n4 = 3 * Ppad / 4;
        for (int e = threadIdx.x; e < n4; e += 64 * PC_WAVES) dst[e] = src[e];
        const float4 *sb = (const float4 *)(jb.boxes + (size_t)b * nboxp * 8);
        float4 *db = (float4 *)bxs;
        for (int e = threadIdx.x; e < 2 * nboxp; e += 64 * PC_WAVES) db[e] = sb[e];
    }
    __syncthreads();
    const int g = blk * gpw + w / S, sl = w % S;
    unsigned int pairs = 0u;
    int work = 0;
    unsigned char *wbase = s_dyn + (size_t)(3 * Ppad + 8 * nboxp) * 4;
    unsigned long long *s_key = (unsigned long long *)(wbase + (size_t)w * PC_WAVE_BYTES + 64 * 16);
    float qx = 0.f, qy = 0.f, qz = 0.f;
    float bm[3] = {INFINITY, INFINITY, INFINITY};
    int bb[3] = {-1, -1, -1};
    if (g < jb.nqg) {
        unsigned char *wl = wbase + (size_t)w * PC_WAVE_BYTES;
        float *s_qc = (float *)wl;
        unsigned int *s_q = (unsigned int *)(wl + 64 * 16 + 3 * 64 * 8);
        if (kind == 1) {
            float m1[1]; int b1[1];
            knn_pruned_wave<1, true, PC_QCAP, false>(jb, S, a.sparse, b, g, sl, cl, Ppad, bxs, nullptr, s_qc, s_q, nullptr, a.share, nullptr, s_key, qx, qy, qz,
                                              m1, b1, work, pairs);
            bm[0] = m1[0]; bb[0] = b1[0];
        } else {
            knn_pruned_wave<3, true, PC_QCAP, false>(jb, S, a.sparse, b, g, sl, cl, Ppad, bxs, nullptr, s_qc, s_q, nullptr, a.share, nullptr, s_key, qx, qy, qz,
                                              bm, bb, work, pairs);
        }
        if (S > 1 && sl > 0) {
            // the slice's candidates as (distance bits, block) keys in its own (now idle) result slots
#pragma unroll
            for (int k = 0; k < 3; ++k)
                s_key[64 * k + lane] = bb[k] >= 0 ? (((unsigned long long)__float_as_uint(bm[k]) << 32) | (unsigned)bb[k]) : ~0ull;
            if (lane == 0) { s_key[192] = (unsigned long long)(unsigned)work; s_key[193] = (unsigned long long)pairs; }   // queue area
        }
    }
    if (S > 1) __syncthreads();
    if (g < jb.nqg && sl == 0) {
        // ---- slice 0 merges its group's slices by the full (distance, block) key: distinct slices hold distinct boxes
        for (int t = 1; t < S; ++t) {
            const unsigned long long *ok = (const unsigned long long *)(wbase + (size_t)(w + t) * PC_WAVE_BYTES + 64 * 16);
            work += (int)(unsigned)ok[192]; pairs += (unsigned)ok[193];
            if (kind == 1) {
                const unsigned long long k = ok[lane];
                const float m = __uint_as_float((unsigned)(k >> 32));
                const int bt = (int)(unsigned)k;
                if (k != ~0ull && ((m < bm[0]) || (m == bm[0] && (bb[0] < 0 || bt < bb[0])))) { bm[0] = m; bb[0] = bt; }
            } else {
                float kd[3];
                int ki[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) { kd[c] = bm[c]; ki[c] = bb[c] >= 0 ? bb[c] : 0x7fffffff; }
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const unsigned long long k = ok[64 * c + lane];
                    if (k != ~0ull) reart_top3_insert(kd, ki, __uint_as_float((unsigned)(k >> 32)), (int)(unsigned)k);
                }
#pragma unroll
                for (int c = 0; c < 3; ++c) { bm[c] = kd[c]; bb[c] = ki[c] == 0x7fffffff ? -1 : ki[c]; }
            }
        }
        const int i = g * NN_BS + lane;
        if (kind == 1) {
            // exact (lowest) index inside the winning half box
            int bi = 0x7fffffff;
            if (bb[0] >= 0) {
                const float *tx = cl, *ty = cl + Ppad, *tz = cl + 2 * (size_t)Ppad;
#pragma unroll
                for (int u = NN_BOX / 2 - 1; u >= 0; --u) {
                    const float d = reart_sqdist3(qx, qy, qz, tx[bb[0] + u], ty[bb[0] + u], tz[bb[0] + u]);
                    if (d == bm[0]) bi = bb[0] + u;
                }
            }
            pairs += 64u * (NN_BOX / 2);
            if (i < jb.P1) {
                const size_t o = (size_t)b * jb.P1 + i;
                jb.pd[o] = bm[0]; jb.pi[o] = bi;
            }
        } else if (i < jb.P1) {
            const size_t o = ((size_t)b * jb.P1 + i) * 3;
#pragma unroll
            for (int k = 0; k < 3; ++k) { jb.pd[o + k] = bm[k]; jb.pi[o + k] = bm[k] < INFINITY ? bb[k] : -1; }
        }
        if (lane == 0 && jb.cost) jb.cost[b * jb.nqg + g] = (unsigned int)work;
    } else {
        pairs = 0u;                       // counted by the group's slice-0 wave
    }
    if (a.prof) {
        __shared__ unsigned int s_pairs[PC_WAVES];
        if (lane == 0) s_pairs[w] = pairs;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned int tot = 0u;
            for (int k = 0; k < PC_WAVES; ++k) tot += s_pairs[k];
            a.prof[2 * (size_t)L] = t0; a.prof[2 * (size_t)L + 1] = wall_clock64();
            a.prof_pairs[L] = tot;
        }
    }
}

// LDS bytes of the cloud-resident form for the largest target cloud of the launch; 0 = does not fit
static size_t search_cloud_lds(const SearchArgs &a) {
    int Ppad = 0;
    for (int j = 0; j < a.n1; ++j) Ppad = a.k1[j].Ppad > Ppad ? a.k1[j].Ppad : Ppad;
    if (a.n3) Ppad = a.k3.Ppad > Ppad ? a.k3.Ppad : Ppad;
    const size_t need = (size_t)(3 * Ppad + 8 * (Ppad / NN_BOX)) * 4 + (size_t)PC_WAVES * PC_WAVE_BYTES + 256;
    return need <= 152 * 1024 ? need : 0;
}

static int search_check(SearchArgs &a) {
    if (a.n1 < 0 || a.n1 > 2 || a.n3 < 0 || a.n3 > 1 || a.n1 + a.n3 == 0 || a.G < 1) return REART_ERR_INVALID_ARG;
    if ((a.n1 && (a.S1 < 1 || a.S1 > PR_SMAX)) || (a.n3 && (a.S3 < 1 || a.S3 > PR_SMAX))) return REART_ERR_INVALID_ARG;
    for (int j = 0; j < a.n1; ++j)
        if (!a.k1[j].boxes || !a.k1[j].seed || !a.k1[j].pd || !a.k1[j].pi) return REART_ERR_INVALID_ARG;
    if (a.n3 && (!a.k3.boxes || !a.k3.seed || !a.k3.pd || !a.k3.pi)) return REART_ERR_INVALID_ARG;
    a.per = reart_div_up(a.G, 8);
    a.k_nqg = a.n1 ? a.k1[0].nqg : a.k3.nqg;
    return REART_OK;
}

int reart_search_launch(const SearchArgs &a_in, hipStream_t st) {
    SearchArgs a = a_in;
    const int rc = search_check(a);
    if (rc != REART_OK) return rc;
    const size_t cl_lds = a.cloud_resident ? search_cloud_lds(a) : 0;
    a.cloud_slices = (a.cloud_slices == 1 || a.cloud_slices == 2 || a.cloud_slices == 4 || a.cloud_slices == 8) ? a.cloud_slices : 4;
    if (cl_lds) {
        if (hipFuncSetAttribute((const void *)knn_cloud_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024) != hipSuccess)
            return REART_ERR_LAUNCH;
        hipLaunchKernelGGL(knn_cloud_kernel, dim3(reart_search_grid_cloud(a.n1, a.n3, a.G, a.k_nqg, a.cloud_slices)), dim3(64 * PC_WAVES),
                           cl_lds, st, a);
        REART_CHECK_LAUNCH();
        return REART_OK;
    }
    return reart_search_launch_batch(&a, 1, st);
}
// K searches of one geometry in one launch (group form): instance k is the grid's row k
int reart_search_launch_batch(const SearchArgs *ak, int K, hipStream_t st) {
    if (K < 1 || K > REART_BATCH_MAX) return REART_ERR_INVALID_ARG;
    Batched<SearchArgs> ab = {};
    for (int k = 0; k < K; ++k) {
        ab.a[k] = ak[k];
        const int rc = search_check(ab.a[k]);
        if (rc != REART_OK) return rc;
        const SearchArgs &a = ab.a[k], &a0 = ab.a[0];
        if (a.n1 != a0.n1 || a.n3 != a0.n3 || a.G != a0.G || a.S1 != a0.S1 || a.S3 != a0.S3) return REART_ERR_INVALID_ARG;
    }
    const SearchArgs &a = ab.a[0];
    const int S = (a.n1 ? a.S1 : 0) > (a.n3 ? a.S3 : 0) ? a.S1 : a.S3;
    const size_t lds = (size_t)S * PR_LDS_WAVE_BYTES;
    if (K == 1) hipLaunchKernelGGL(knn_group_kernel<false>, dim3(reart_search_grid(a.n1, a.n3, a.G)), dim3(64 * S), lds, st, ab);
    else hipLaunchKernelGGL(knn_group_kernel<true>, dim3(reart_search_grid(a.n1, a.n3, a.G), K), dim3(64 * S), lds, st, ab);
    REART_CHECK_LAUNCH();
    return REART_OK;
}
// workgroups the launch of `a` will have (the form is chosen exactly as in reart_search_launch)
int reart_search_workgroups(const SearchArgs &a) {
    const int nqg = a.n1 ? a.k1[0].nqg : a.k3.nqg;
    const int cs = (a.cloud_slices == 1 || a.cloud_slices == 2 || a.cloud_slices == 4 || a.cloud_slices == 8) ? a.cloud_slices : 4;
    if (a.cloud_resident && search_cloud_lds(a)) return reart_search_grid_cloud(a.n1, a.n3, a.G, nqg, cs);
    return reart_search_grid(a.n1, a.n3, a.G);
}
int reart_search_grid(int n1, int n3, int G) { return 8 * (n1 + n3) * reart_div_up(G, 8); }
int reart_search_grid_cloud(int n1, int n3, int G, int nqg, int slices) {
    return (n1 + n3) * (G / nqg) * reart_div_up(nqg, PC_WAVES / slices);
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
    {
        const size_t o = ((size_t)b * jb.P1 + i) * KK;                   // one merged record per query
#pragma unroll
        for (int k = 0; k < KK; ++k) {
            const float d = jb.pd[o + k];
            const int q = jb.pi[o + k];
            insert(d, (d < INFINITY && q >= 0) ? q : 0x7fffffff);
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
    p->S = 4;                                                             // waves per search workgroup
    while (p->S > 1 && reart_div_up(P2, p->S) < 64) p->S -= 1;
    p->Ppad = (int)reart_align_up((size_t)(P2 > 0 ? P2 : 1), NN_BOX);
    size_t off = 0;
    p->o_soa = off; off += reart_align_up(sizeof(float) * 3 * (size_t)N * p->Ppad, 256);
    p->o_box = off; off += reart_align_up(sizeof(float) * 8 * (size_t)N * (p->Ppad / NN_BOX), 256);
    p->o_pd = off; off += reart_align_up(sizeof(float) * (size_t)N * P1 * K, 256);
    p->o_pi = off; off += reart_align_up(sizeof(int) * (size_t)N * P1 * K, 256);
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
    SearchArgs sr = {};
    sr.G = N * jb.nqg; sr.S1 = sr.S3 = p.S; sr.sparse = 40; sr.share = 2;
    if (K == 1) { sr.k1[0] = jb; sr.n1 = 1; } else { sr.k3 = jb; sr.n3 = 1; }
    rc = reart_search_launch(sr, st);
    if (rc != REART_OK) return rc;
    const dim3 fg(reart_div_up(P1, 256), N);
    if (K == 1) hipLaunchKernelGGL((knn_warm_finish_kernel<1>), fg, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((knn_warm_finish_kernel<3>), fg, dim3(256), 0, st, a);
    REART_CHECK_LAUNCH();
    return REART_OK;
}
