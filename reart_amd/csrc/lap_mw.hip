// reart_amd/csrc/lap_mw.hip -- the sequential part of a points-form re-solve (lap.hip: lap_jv_kernel<., true, 1>) with the
// ROW REDUCTION RUN ONE CHAIN PER WAVE (reference: the assignment refresh of run_robot.py:164-187 -- scipy's
// linear_sum_assignment on (T-1) matrices cdist(pc_src, pc_tgt), utils/model_utils.py:85-103 -- every assign_gap iterations).
//
// What the one-row-at-a-time form measured (tools/exp_assign_phase.py, 9 x 1024^2 of the base recipe): a refresh frees
// ~380 rows, the augmenting row reduction settles all but 55-90 of them in ~2 000-2 500 steps, those take one shortest
// augmenting path each, 3 000-6 000 Dijkstra steps per problem -- every step a workgroup-wide reduction + barrier.
// Row-reduction chains from different free rows are independent of each other, so here every wave follows its OWN chain:
//   * a wave holds all n <= 64 * CPL columns (CPL target points per lane) in registers; a step is CPL distances per lane
//     and ONE wave reduction -- no barrier;
//   * the shared state (prices, owners, assignment) lives in LDS and is written only under a lock by the wave that
//     COMMITS a step.  Nothing else synchronises the waves: a wave reads the prices whenever it likes.
// Why unsynchronised reads are safe: prices only ever RISE (a step raises its column by v2 - v1 >= 0) and a column never
// loses its owner.  A step that saw a stale (lower) price of a column it did NOT choose underestimated that column --
// which lost the comparison anyway -- and pays at most the gap it would be allowed to pay now: every dual constraint
// still holds and the new pair is tight.  What must not have changed is what the step writes: under the lock the wave
// compares the chosen column's (price, owner) with the values its decision used; a difference and the row is scanned again.
// The remaining rows' path searches then run with the whole workgroup on one search (they are NOT independent: measured,
// see below).  The optimum is the optimum in whatever order the rows are settled; the potentials depend on the
// interleaving, so -- like the raced solves -- they are valid duals that are not reproducible from run to run.  The exact
// certificate of the re-solve (lap_jv_pass_kernel + lap_jv_kernel<., ., 2>) runs afterwards unchanged.
#include "common.h"
#include "internal.h"
#include "lap_dev.h"

#define MW_NW 8              // waves per workgroup = searches in flight per problem (two per SIMD: 256 VGPRs each)
// row-reduction steps allowed per free row before the rest goes to the path searches.  A chain step here costs a third of a
// search step (0.4 us against 1.3-1.5 us), so the budget is four times the one-row-at-a-time solver's: measured per refresh of
// the base recipe (9 x 1024^2, raced), budget 8 / 16 / 32 / 64: 10.5 / 10.1 / 9.7 / 10.7 ms -- rows left 50-100 / 15-55 / 5-25 /
// 2-13, of which the last few need searches of hundreds of steps whatever the budget.
#define MW_ARR_BUDGET 32
#define MW_CHECK 16          // a search looks at the race flag and at its labelled columns every MW_CHECK steps
// A search that has not met a sink after MW_BUCKET_AFTER one-column steps goes on in BUCKETS (see the search loop): all
// unlabelled columns within `width` of the closest one are settled together by label-correcting rounds.
#define MW_BUCKET_AFTER 8    // (replayed solves, columns in caller order, 16 / 24 / 32 / 48: 250 / 252 / 257 / 258 ms over the recipe's slowest 24; with the
                             // columns along a Z-order curve -- see below -- and 4-24 buckets, 4 / 8 / 12 / 24 / 40: 189 / 190 / 192 / 197 / 206)
#define MW_BUCKET_W0 1e-8    // first bucket width of a problem, as a fraction of the cost scale
#define MW_BK 192            // columns relaxed from per round (the rest of a bucket waits for the next round); 12 KB of lists: two 1024-column workgroups per compute unit
// (the three below were first set by whole-loop runs, whose trajectories are chaotic: 8 / 48 / 4.0.  On DUMPED solves replayed with
// every variant, tools/replay_tail.py, same box -- recipe slowest 24 / recipe sample 40 / projection slowest 24 / projection
// sample 40, ms: 8, 48: 359 / 127 / 796 / 207 | 4, 24: 311 / 118 / 657 / 186 | 2, 12: 272 / 117 / 575 / 177 | 2, 8: 276 / 117 /
// 578 / 178 | 1, 6: 288 / 146 / 602 / 205 | 16, 96: 424 / 145 / 996 / 255; widening twofold: 259 / 116 / 555 / 178.  A wide bucket
// relaxes from its members again and again: 2.6 entries per settled column at 8 / 48, 1.4 at 2 / 12.  Since the loops number
// their columns along a Z-order curve (lap.spatial_order: the exact path of a relaxation runs in one or two waves, an entry is
// cheaper) the balance sits a little wider -- 2, 12: 198 / 101 / 398 / 143 | 3, 16: 193 / 97 / 398 / 139 | 4, 24: 195 / 95 / 402 / 139.)
#define MW_BUCKET_LO 4       // a bucket that closes with fewer columns than this widens the next one ...
#define MW_BUCKET_HI 24      // ... with more than this, halves it
#define MW_BUCKET_UP 2.0
#define MW_FOREST_LO 8       // the same thresholds for the backward growth's buckets (lap_mc_forest_kernel; fourfold)
#define MW_FOREST_HI 48

#ifdef REART_PRUNE_PHASE   // diagnostic build only (make -C reart_amd/csrc phase; tools/exp_mw.py)
// per workgroup (first 64): 0 set-up | 1 row-reduction phase | 2 path-search phase (wall ticks of wave 0) | 3, 4 ticks the waves
// spent inside those phases' work loops (summed over waves) | 5 ticks waiting for the lock | 6 path-search steps thrown away |
// 7 path-search steps in all | 8 searches started | 9 longest search (steps)
__device__ unsigned long long g_mw_phase[64 * 10];
extern "C" int reart_debug_mw_phase(unsigned long long *out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mw_phase), sizeof(g_mw_phase)) != hipSuccess) return REART_ERR_LAUNCH;
    if (reset) { static unsigned long long z[64 * 10]; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_mw_phase), z, sizeof(z)); }
    return REART_OK;
}
#define MWP_ADD(k, v) do { if (lane == 0 && blockIdx.y == 0 && blockIdx.x < 64) atomicAdd(&g_mw_phase[blockIdx.x * 10 + (k)], (unsigned long long)(v)); } while (0)
#define MWP_MAX(k, v) do { if (lane == 0 && blockIdx.y == 0 && blockIdx.x < 64) atomicMax(&g_mw_phase[blockIdx.x * 10 + (k)], (unsigned long long)(v)); } while (0)
#define MWP_NOW() wall_clock64()
// s_memtime ticks of the search step's sections, wave 0 of workgroup (0, 0): arg-min | barrier | merge | reads | relaxation
__device__ unsigned long long g_mw_step[8];
extern "C" int reart_debug_mw_step(unsigned long long *out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mw_step), sizeof(g_mw_step)) != hipSuccess) return REART_ERR_LAUNCH;
    if (reset) { static unsigned long long z[8]; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_mw_step), z, sizeof(z)); }
    return REART_OK;
}
#define MWS_DECL unsigned long long mws_t = __builtin_amdgcn_s_memtime(), mws[6] = {0, 0, 0, 0, 0, 0}
#define MWS(k) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); mws[k] += n_ - mws_t; mws_t = n_; } while (0)
#define MWS_FLUSH() do { if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) for (int k_ = 0; k_ < 6; ++k_) g_mw_step[k_] += mws[k_]; } while (0)
#else
#define MWS_DECL do { } while (0)
#define MWS(k) do { } while (0)
#define MWS_FLUSH() do { } while (0)
#define MWP_ADD(k, v) do { } while (0)
#define MWP_MAX(k, v) do { } while (0)
#define MWP_NOW() 0ull
#endif

struct MwShared {
    int lock, qhead, nnext, budget, abort_, flag, unsolved;
    int steps, arr, conflicts;
};

__device__ __forceinline__ void mw_lock(int *lock) {
    if ((threadIdx.x & 63) == 0) {
        for (;;) {
            int expected = 0;
            if (__hip_atomic_compare_exchange_strong(lock, &expected, 1, __ATOMIC_ACQUIRE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");      // every lane's reads of the state come after the lock
}
__device__ __forceinline__ void mw_unlock(int *lock) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // every lane's writes are in LDS before the lock opens
    if ((threadIdx.x & 63) == 0) __hip_atomic_store(lock, 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ int mw_flag(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ int mw_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// sqrtf(x) for x >= 0, bit for bit: v_sqrt_f32 (within one unit in the last place) and the library's own correction -- the
// neighbours s -/+ 1 ulp judged by the sign of their residuals -- without its range scaling for inputs below 2^-96 and its
// class test (zero needs neither: the residuals of 0 are NaN and 0, nothing is selected); such inputs take sqrtf itself.
// Five instructions fewer per square root on a step that is bound by the instructions it issues.
__device__ __forceinline__ float mw_sqrt(float x) {
    if (__builtin_expect(x < 0x1p-96f && x > 0.f, 0)) return sqrtf(x);
    const float s = __builtin_amdgcn_sqrtf(x);
    const float sm = __int_as_float(__float_as_int(s) - 1), sp = __int_as_float(__float_as_int(s) + 1);
    const float rm = fmaf(-sm, s, x), rp = fmaf(-sp, s, x);
    float r = rm <= 0.f ? sm : s;
    r = rp > 0.f ? sp : r;
    return r;
}

// (smallest value, its lowest key) over the first 2^STEPS lanes of the wave (STEPS = 6: all of it), every lane gets the result.
// Two butterflies on 32-bit INTEGERS instead of one on doubles: a double's bit pattern, sign-folded, orders like the number, so
// the minimum of the high words, then of the low words among the lanes that attain it, is the minimum value -- v_min_u32 with
// the lane permutation folded in, where the double form takes two moves and a v_min_f64 per step.  Exact ties go by the key.
template <int STEPS>
__device__ __forceinline__ unsigned mw_bmin(unsigned x) {
    x = min(x, (unsigned)reart_bfly<0>((int)x));
    if (STEPS > 1) x = min(x, (unsigned)reart_bfly<1>((int)x));
    if (STEPS > 2) x = min(x, (unsigned)reart_bfly<2>((int)x));
    if (STEPS > 3) x = min(x, (unsigned)reart_bfly<3>((int)x));
    if (STEPS > 4) x = min(x, (unsigned)reart_bfly<4>((int)x));
    if (STEPS > 5) x = min(x, (unsigned)reart_bfly<5>((int)x));
    return STEPS < 6 ? (unsigned)__builtin_amdgcn_readfirstlane((int)x) : x;
}
template <int STEPS>
__device__ __forceinline__ void mw_argmin_key(double &v, int &j) {
    const bool in = STEPS == 6 || (int)(threadIdx.x & 63) < (1 << STEPS);
    const int hi = __double2hiint(v), lo = __double2loint(v);
    const unsigned sg = (unsigned)(hi >> 31);
    const unsigned kh = in ? (unsigned)hi ^ (sg | 0x80000000u) : 0xffffffffu, kl = (unsigned)lo ^ sg;
    const unsigned mh = mw_bmin<STEPS>(kh);
    const unsigned ml = mw_bmin<STEPS>(kh == mh ? kl : 0xffffffffu);
    const bool at = in && kh == mh && kl == ml;
    unsigned long long m = __ballot(at);
    if (__builtin_popcountll(m) > 1) {
        const unsigned mj = mw_bmin<STEPS>(at ? (unsigned)j : 0xffffffffu);
        m = __ballot(at && (unsigned)j == mj);
    }
    const int wl = m ? __ffsll((long long)m) - 1 : 0;
    j = __builtin_amdgcn_readlane(j, wl);
    v = __hiloint2double(__builtin_amdgcn_readlane(hi, wl), __builtin_amdgcn_readlane(lo, wl));
}

typedef float mw_f4 __attribute__((ext_vector_type(4)));
// x rounded to fp32 so that the result is not below it (finite x, +inf): (float)(x (1 +- 2^-23)) >= x whatever the conversion's rounding
__device__ __forceinline__ float mw_f32_up(double x) { return (float)(x + fabs(x) * 1.1920928955078125e-07); }

#define MW_FILTER_K 1.00000095367431640625       // 1 + 2^-20: see the bucket rounds' filter
// the filter's squared distance: one product and two fused multiply-adds per pair of columns (an estimate within 2^-22 of the
// expression of the costs, which is never contracted)
__device__ __forceinline__ jv_f2 mw_sq_estimate(jv_f2 dx, jv_f2 dy, jv_f2 dz) {
#pragma clang fp contract(fast)
    return dx * dx + dy * dy + dz * dz;
}
// this lane's CPL costs of row (ax, ay, az): reart_cdist's expression, two columns per packed-fp32 operand
template <int CPL>
__device__ __forceinline__ void mw_row_costs(float ax, float ay, float az, const jv_f2 (&tcx)[CPL / 2], const jv_f2 (&tcy)[CPL / 2],
                                             const jv_f2 (&tcz)[CPL / 2], float (&rc)[CPL]) {
    const jv_f2 ax2 = {ax, ax}, ay2 = {ay, ay}, az2 = {az, az};
#pragma unroll
    for (int k = 0; k < CPL / 2; ++k) {
        const jv_f2 dx = ax2 - tcx[k], dy = ay2 - tcy[k], dz = az2 - tcz[k];
        const jv_f2 sq = (dx * dx + dy * dy) + dz * dz;
        rc[2 * k] = mw_sqrt(sq.x); rc[2 * k + 1] = mw_sqrt(sq.y);
    }
}

// Column j = 64 k + lane is slot k of lane `lane` in every wave.
// STAGE 0: the whole sequential part in one launch.  STAGE 1: set-up only -- the state after the release / greedy steps goes to
// a.mc_* for lap_mc_arr_kernel.  STAGE 2: the path searches only, from the state the row reduction left in a.mc_*.
template <int CPL, int STAGE, int NW = MW_NW>
__global__ __launch_bounds__(64 * NW) void lap_jvmw_kernel(JvArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lsm[];
    constexpr int BS = 64 * NW;
    const int n = a.n, b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    double *price = (double *)lsm;                                      // [n]
    int *owner = (int *)(price + n);                                    // [n] column -> row
    int *assigned = owner + n;                                          // [n] row -> column
    int *flist = assigned + n;                                          // [n] free rows
    int *next = flist + n;                                              // [n] rows left for the path search
    float *psx = (float *)(next + n), *psy = psx + n, *psz = psy + n;   // source points
    float *ptx = psz + n, *pty = ptx + n, *ptz = pty + n;               // target points (wave-uniform reads of one column)
    double *hcol = (double *)(ptz + n);                                 // [n] searches: potential of the row that owns the column
    int *tof = (int *)(hcol + n), *tpr = tof + n;                       // [n] each: the column's tree (lap_mc_trees_kernel) | its parent there
    __shared__ MwShared sh;
    __shared__ double s_red[NW];
    __shared__ int s_cw[NW];
    const bool race = STAGE != 1 && a.done != nullptr;
    const int racer = race ? (int)blockIdx.y : 0;
    [[maybe_unused]] const unsigned long long tp0_ = MWP_NOW();
    auto lost = [&]() -> int { return __hip_atomic_load(a.done + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };

    // ---- the problem into LDS / registers
    double mx;
    {
        float lo = INFINITY, hi = -INFINITY;
        const float *S_ = a.src + (size_t)b * n * 3, *T_ = a.tgt + (size_t)b * n * 3;
        for (int e = tid; e < 3 * n; e += BS) {
            const float sv = S_[e], tv = T_[e];
            (e % 3 == 0 ? psx : (e % 3 == 1 ? psy : psz))[e / 3] = sv;
            (e % 3 == 0 ? ptx : (e % 3 == 1 ? pty : ptz))[e / 3] = tv;
            lo = fminf(lo, fminf(sv, tv)); hi = fmaxf(hi, fmaxf(sv, tv));
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { lo = fminf(lo, __shfl_xor(lo, o, 64)); hi = fmaxf(hi, __shfl_xor(hi, o, 64)); }
        mx = 1.7320508 * (double)(hi - lo);      // the scale of the costs (the tolerances are fractions of it), as lap_jv_kernel's
    }
    if (lane == 0) s_red[wv] = mx;
    for (int j = tid; j < n; j += BS) {
        if (STAGE == 2) {
            price[j] = a.mc_price[(size_t)b * n + j]; owner[j] = a.mc_owner[(size_t)b * n + j];
            assigned[j] = a.mc_assigned[(size_t)b * n + j]; next[j] = a.mc_next[(size_t)b * n + j];
            tof[j] = a.mc_tree[(size_t)b * n + j]; tpr[j] = a.mc_tpar[(size_t)b * n + j];
            continue;
        }
        price[j] = (race && STAGE == 0) ? a.price_start[(size_t)b * n + j] : (a.price_in ? a.price_in[(size_t)b * n + j] : 0.0);
        owner[j] = -1; tof[j] = -1; tpr[j] = -1;
        const int c = ((race && STAGE == 0) ? a.col_start : a.col4row)[(size_t)b * n + j];
        assigned[j] = (c >= 0 && c < n) ? c : -1;
        next[j] = 0x7fffffff;
    }
    if (tid == 0) { sh.lock = 0; sh.qhead = 0; sh.nnext = 0; sh.abort_ = 0; sh.flag = 0; sh.unsolved = 0; sh.steps = 0; sh.arr = 0; sh.conflicts = 0; }
    __syncthreads();
    mx = 0.0;
    for (int w = 0; w < NW; ++w) mx = fmax(mx, s_red[w]);
    if (!(mx > 0.0)) mx = 1.0;
    if (STAGE == 2) mx = a.scale[b];
    const double keep_tol = mx * a.keep_tol;
    jv_f2 tcx[CPL / 2], tcy[CPL / 2], tcz[CPL / 2];
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
        const int j = 64 * k + lane < n ? 64 * k + lane : 0;
        tcx[k >> 1][k & 1] = ptx[j]; tcy[k >> 1][k & 1] = pty[j]; tcz[k >> 1][k & 1] = ptz[j];
    }
    int st_freed = 0, nleft = 0;
    int my_steps = 0, my_arr = 0, my_conf = 0;
    [[maybe_unused]] unsigned long long tp_ = MWP_NOW();
    if (STAGE != 2) {
    // previous pairs: a repeated column keeps its lowest row
    for (int i = tid; i < n; i += BS)
        if (assigned[i] >= 0) atomicMin((unsigned int *)&owner[assigned[i]], (unsigned int)i);   // -1 = 0xffffffff: empty
    __syncthreads();
    for (int i = tid; i < n; i += BS)
        if (assigned[i] >= 0 && owner[assigned[i]] != i) assigned[i] = -1;
    __syncthreads();
    // pairs that lost their arg-min under the old prices are released (row minima from lap_jv_pass_kernel); a kept pair counts
    // as tight: a row's potential is never stored here, it IS c_i,s(i) + p_s(i)
    for (int i = tid; i < n; i += BS) {
        const int j = assigned[i];
        flist[i] = a.pre_j1[(size_t)b * n + i];
        if (j >= 0 && a.pre_cur[(size_t)b * n + i] - a.pre_v1[(size_t)b * n + i] > keep_tol) { assigned[i] = -1; owner[j] = -1; }
    }
    __syncthreads();
    // greedy: a free row takes its arg-min column when nobody owns it (lowest row wins)
    for (int i = tid; i < n; i += BS)
        if (assigned[i] < 0) { atomicAdd(&sh.flag, 1); if (owner[flist[i]] < 0) atomicMin(&next[flist[i]], i); }
    __syncthreads();
    st_freed = sh.flag;
    for (int i = tid; i < n; i += BS)
        if (assigned[i] < 0 && owner[flist[i]] < 0 && next[flist[i]] == i) assigned[i] = flist[i];
    __syncthreads();
    for (int i = tid; i < n; i += BS)
        if (assigned[i] >= 0) owner[assigned[i]] = i;
    if (tid == 0) sh.flag = 0;
    __syncthreads();
    // the rows still free, ascending
    for (int i0 = 0; i0 < n; i0 += BS) {
        const int i = i0 + tid;
        const bool fr = i < n && assigned[i] < 0;
        const unsigned long long m = __ballot(fr);
        if (lane == 0) s_cw[wv] = __builtin_popcountll(m);
        __syncthreads();
        int off = sh.flag;
        for (int w = 0; w < wv; ++w) off += s_cw[w];
        if (fr) flist[off + __builtin_popcountll(m & ((1ull << lane) - 1ull))] = i;
        __syncthreads();
        if (tid == 0) { int t = 0; for (int w = 0; w < NW; ++w) t += s_cw[w]; sh.flag += t; }
        __syncthreads();
    }
    const int nfree = sh.flag;
    if (tid == 0) sh.budget = MW_ARR_BUDGET * nfree + 64;
    __syncthreads();

    if (STAGE == 1) {                                   // hand the state to lap_mc_arr_kernel
        for (int j = tid; j < n; j += BS) {
            a.mc_price[(size_t)b * n + j] = price[j]; a.mc_owner[(size_t)b * n + j] = owner[j];
            a.mc_assigned[(size_t)b * n + j] = assigned[j]; a.mc_list[(size_t)b * n + j] = flist[j];
            a.mc_tree[(size_t)b * n + j] = -1; a.mc_tpar[(size_t)b * n + j] = -1;
        }
        // the unowned columns, for lap_mc_tighten_kernel (the list lives where the row reduction will leave its rows: spent by then)
        if (tid == 0) sh.flag = 0;
        __syncthreads();
        for (int j = tid; j < n; j += BS)
            if (owner[j] < 0) a.mc_next[(size_t)b * n + atomicAdd(&sh.flag, 1)] = j;
        __syncthreads();
        if (tid == 0) {
            int *c = a.mc_cnt + 8 * b;
            c[0] = nfree; c[1] = 0; c[2] = 0; c[3] = 0; c[4] = 0; c[5] = st_freed; c[6] = 0; c[7] = sh.flag;
            if (b == 0) a.mc_cnt[8 * a.B] = 0;                          // launch-wide: row-reduction teams that have run out of rows
            a.scale[b] = mx;
            // the outputs and the racers' meeting point start defined HERE, not by fills in front of the solve (every launch of a
            // refresh is a 5 us slot of its own, however little it does: tools/solve_gaps.py)
            a.certified[b] = 0;
            if (a.done_clear) a.done_clear[b] = 0;
            if (a.stats) { int *o = a.stats + 4 * b; o[0] = 0; o[1] = 0; o[2] = 0; o[3] = 0; }
        }
        return;
    }
    tp_ = MWP_NOW();
    if (wv == 0) MWP_ADD(0, tp_ - tp0_);
    // ---- augmenting row reduction, one chain per wave: the row takes its cheapest column and pays the gap to its second
    // cheapest (the pair is tight, every other constraint still holds), the row it displaces goes on in the same wave
    for (bool go = true; go;) {
        int q = 0;
        if (lane == 0) q = atomicAdd(&sh.qhead, 1);
        q = mw_uniform(q);
        if (q >= nfree) break;
        int i = flist[jv_order(q, nfree, racer)];
        for (;;) {
            if (mw_flag(&sh.abort_)) { go = false; break; }
            const float ax = psx[i], ay = psy[i], az = psz[i];
            float rc[CPL];
            mw_row_costs<CPL>(ax, ay, az, tcx, tcy, tcz, rc);
            double v1 = INFINITY, v2 = INFINITY;
            int j1 = 0x7fffffff, pay = 0;
#pragma unroll
            for (int k = 0; k < CPL; ++k) {
                const int j = 64 * k + lane;
                lap_top2_push((double)rc[k] + (j < n ? price[j] : INFINITY), j, v1, j1, v2);
            }
            lap_wave_top2_fast(v1, j1, v2, pay);
            if (!(v1 < INFINITY)) { if (lane == 0) sh.unsolved = 1; go = false; break; }       // non-finite costs
            // what the decision rests on: the arg-min column's price and owner (wave-uniform reads)
            const double pj1 = price[j1];
            const int own = owner[j1];
            const bool consistent = (double)mw_sqrt(reart_sqdist3(ax, ay, az, ptx[j1], pty[j1], ptz[j1])) + pj1 == v1;
            const bool tie = !(v1 < v2);
            int bud = 0;
            if (lane == 0) bud = atomicSub(&sh.budget, 1);
            bud = mw_uniform(bud);
            if (bud <= 0 || (tie && own >= 0)) {                // out of budget / an exact tie on an owned column: path search
                if (lane == 0) next[atomicAdd(&sh.nnext, 1)] = i;
                break;
            }
            { [[maybe_unused]] const unsigned long long tl_ = MWP_NOW(); mw_lock(&sh.lock); MWP_ADD(5, MWP_NOW() - tl_); }
            const bool ok = consistent && price[j1] == pj1 && owner[j1] == own;
            if (ok && lane == 0) {
                if (!tie) price[j1] = pj1 + (v2 - v1);
                owner[j1] = i; assigned[i] = j1;
                if (own >= 0) assigned[own] = -1;
            }
            mw_unlock(&sh.lock);
            if (!ok) { ++my_conf; continue; }
            ++my_arr;
            if (race && (my_arr & (MW_CHECK - 1)) == 0 && lane == 0 && lost()) sh.abort_ = 1;
            if (own < 0) break;
            i = own;
        }
    }
    MWP_ADD(3, MWP_NOW() - tp_);
    __syncthreads();
    if (wv == 0) MWP_ADD(1, MWP_NOW() - tp_);
    tp_ = MWP_NOW();
    nleft = sh.nnext;
    if (tid == 0) sh.qhead = 0;
    __syncthreads();
    } else {                                            // STAGE 2: what the row reduction on the other compute units left
        const int *c = a.mc_cnt + 8 * b;
        nleft = c[2]; st_freed = c[5] | ((c[7] & 0x3ff) << 21);     // (c[7]: rounds of lap_mc_forest_kernel, reported next to the released rows)
        if (tid == 0) { sh.arr = c[3]; sh.conflicts = c[4]; sh.unsolved = c[6]; }
        __syncthreads();
    }

    // ---- one shortest augmenting path per remaining row, the whole workgroup on one search (thread t owns the columns
    // t, t + BS, ...: labels, predecessors and prices in registers; a step is one workgroup arg-min + ONE barrier).  Searches
    // per wave were built and measured first (profiles/r04_lap_per_wave_search_variant.hip.txt): the row reduction's chains
    // are independent, the searches are not -- 5 % of them label > 500 of the 1024 columns and hold 45 % of all steps, any
    // commit elsewhere invalidates them, and a wave alone takes 1.7 us per step.
    constexpr int CPT = CPL >= NW ? CPL / NW : 1;   // n <= 64 CPL <= BS * CPT
    static_assert(CPT >= 1, "a thread owns at least one column");
    float qx[CPT], qy[CPT], qz[CPT];
    double pj[CPT];
    unsigned deadq = 0u;
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
        const int j = tid + k * BS;
        const int jj = j < n ? j : 0;
        qx[k] = ptx[jj]; qy[k] = pty[jj]; qz[k] = ptz[jj];
        pj[k] = j < n ? price[j] : INFINITY;
        if (j >= n) deadq |= 1u << k;
    }
    // a matched row's potential is the cost of its own pair, c_i,s(i) + p_s(i) (tight by construction): kept per COLUMN and
    // rewritten, with the same expression the relaxations use, whenever a search has moved the column's price or owner
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
        const int j = tid + k * BS;
        const int i = j < n ? owner[j] : -1;
        if (i >= 0) hcol[j] = (double)mw_sqrt(reart_sqdist3(psx[i], psy[i], psz[i], qx[k], qy[k], qz[k])) + pj[k];
    }
    int *cpred = flist;                                   // the free-row list is spent: column -> row it was reached from
    // ---- the unowned columns' prices.  A column a released row left keeps the price it had: tight for a pair that no longer
    // exists.  Nothing constrains an unowned column's price from below except the matched rows' potentials (row i must not
    // prefer it to its own column: c_ij + p_j >= u_i), so it is lowered until the first matched row is indifferent -- p_j -=
    // min_i (c_ij + p_j - u_i), the expression of the relaxations.  Every dual constraint still holds, no pair changes, and a
    // search now meets the column as soon as it has labelled that row instead of after every column cheaper than the gap.
    if (nleft > 0 && !mw_flag(&sh.abort_) && !mw_flag(&sh.unsolved)) {
        if (tid == 0) sh.flag = 0;
        __syncthreads();                                  // hcol is complete
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            const int j = tid + k * BS;
            if (j < n && owner[j] < 0) cpred[atomicAdd(&sh.flag, 1)] = j;
        }
        __syncthreads();
        const int nh = sh.flag;
        for (int hI = 0; hI < nh; ++hI) {
            const int jh = cpred[hI];
            const float hx = ptx[jh], hy = pty[jh], hz = ptz[jh];
            const double ph = price[jh];
            double m = INFINITY;
#pragma unroll
            for (int k = 0; k < CPT; ++k) {
                const int j = tid + k * BS;
                const int i = j < n ? owner[j] : -1;
                if (i >= 0) m = fmin(m, (((double)mw_sqrt(reart_sqdist3(psx[i], psy[i], psz[i], hx, hy, hz)) + ph) - hcol[j]));
            }
            m = lap_wave_min_d(m);
            if (lane == 0) s_red[wv] = m;
            __syncthreads();
            m = s_red[0];
#pragma unroll
            for (int w = 1; w < NW; ++w) m = fmin(m, s_red[w]);
            __syncthreads();                              // s_red is rewritten by the next column
            if (tid == 0 && m > 0.0 && m < INFINITY) price[jh] = ph - m;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            const int j = tid + k * BS;
            if (j < n) pj[k] = price[j];
        }
    }
    // (Also measured: rounds that settle several columns -- every wave's closest column a candidate, candidates relaxed from
    // ahead of their turn, the sorted ready prefix settled together, profiles/r04_lap_speculative_rounds_variant.hip.txt.
    // Exact, 1.6 columns per round, but a round cost 2.6 us against 1.27 us per step: the step is bound by the instructions
    // its waves issue -- 35 per column and relaxation, the correctly rounded square root among them --, not by latencies
    // that extra relaxations could hide.)
    __shared__ double s_rv[2][NW];
    __shared__ int s_rj[2][NW], s_lostp[2];
    // bucket rounds: the columns relaxed FROM in a round (their rows' points and potentials, their labels), two buffers; the
    // list lengths rotate through three slots (slot r + 1 is cleared before the barrier of round r: its last reader passed
    // the barrier of round r - 1)
    // (an entry's hot part is ONE 16-byte broadcast read: the row's point and g = (its potential - its label), rounded UP to fp32;
    // the label and the potential themselves are read by the few pairs that pass the filter)
    __shared__ __attribute__((aligned(16))) float s_e4[2][MW_BK / 2][8];     // entries in PAIRS: x0 x1 y0 y1 z0 z1 g0 g1 -- packed operands as they are read
    __shared__ __attribute__((aligned(16))) double s_eg[2][MW_BK][2];
    __shared__ int s_ei[2][MW_BK], s_ecnt[3];
    __shared__ double s_bv[2][NW], s_bs[2][NW];
    __shared__ int s_bj[2][NW], s_bsj[2][NW], s_bn[2][NW], s_blost[2];
    if (tid == 0) { s_lostp[0] = 0; s_lostp[1] = 0; s_ecnt[0] = 0; s_ecnt[1] = 0; s_ecnt[2] = 0; }
    double bwidth = mx * MW_BUCKET_W0;                    // bucket width, carried from search to search
#ifdef REART_PRUNE_PHASE
    int my_relax = 0, my_rounds = 0, my_buckets = 0;     // diagnostic build: columns relaxed from / rounds / buckets, reported in place of the commit conflicts and reduction steps
#endif
    int brot = 0, bpar = 0;                               // rotating slot of the list length / parity of the closing reductions
    bool aborted = mw_flag(&sh.abort_) != 0, unsolved = mw_flag(&sh.unsolved) != 0;     // uniform: read after the barrier
    MWS_DECL;
    for (int f = 0; f < nleft && !aborted && !unsolved; ++f) {
        const int i0 = next[jv_order(f, nleft, racer)];
        double d[CPT];
        unsigned scanned = deadq, freecol = 0u;
        {
            const float ax = psx[i0], ay = psy[i0], az = psz[i0];
#pragma unroll
            for (int k = 0; k < CPT; ++k) {
                const int j = tid + k * BS;
                d[k] = (double)mw_sqrt(reart_sqdist3(ax, ay, az, qx[k], qy[k], qz[k])) + pj[k];     // labels up to the row's potential
                if (j < n) { cpred[j] = i0; if (owner[j] < 0) freecol |= 1u << k; }
            }
        }
        double mu = 0.0;
        int sink = -1;
        bool buckets = false;
        MWS(5);
        for (int it = 0; ; ++it) {
            if (it >= MW_BUCKET_AFTER) { buckets = true; break; }
            double bv = INFINITY;
            int bj = 0x7fffffff;
#pragma unroll
            for (int k = 0; k < CPT; ++k)
                if (!((scanned >> k) & 1u)) {
                    const int key = (tid + k * BS) | (((freecol >> k) & 1u) ? 0 : JV_OWNED);       // unowned columns first among ties
                    if (d[k] < bv || (d[k] == bv && key < bj)) { bv = d[k]; bj = key; }
                }
            mw_argmin_key<6>(bv, bj);           // (the double-precision butterfly lap_wave_argmin_fast: 2 800-3 030 ticks per step against 2 710-2 770)
            const int par = it & 1;
            if (lane == 0) { s_rv[par][wv] = bv; s_rj[par][wv] = bj; }
            if (race && tid == 0) s_lostp[par] = (it & (MW_CHECK - 1)) == 0 ? lost() : s_lostp[par ^ 1];
            MWS(0);
            __syncthreads();
            MWS(1);
            if (race && s_lostp[par]) { aborted = true; break; }            // uniform: everybody reads the step's slot
            bv = lane < NW ? s_rv[par][lane] : INFINITY; bj = lane < NW ? s_rj[par][lane] : 0x7fffffff;
            mw_argmin_key<(NW <= 2 ? 1 : (NW <= 4 ? 2 : (NW <= 8 ? 3 : 4)))>(bv, bj);
            ++my_steps;
            mu = bv;
            MWS(2);
            if (bj == 0x7fffffff || !(bv < INFINITY)) { unsolved = true; break; }          // non-finite costs only
            const int jstar = bj & ~JV_OWNED;
            if ((jstar & (BS - 1)) == tid) scanned |= 1u << (jstar / BS);
            const int i = owner[jstar];
            if (i < 0 || tof[jstar] >= 0) { sink = jstar; break; }     // unowned, or a node of a tree that leads to an unowned column at no cost
            const float ax = psx[i], ay = psy[i], az = psz[i];
            const double h = hcol[jstar];                                  // row i's potential
#ifdef REART_PRUNE_PHASE
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
            MWS(3);
#pragma unroll
            for (int k = 0; k < CPT; ++k) {
                const double nd = mu + (((double)mw_sqrt(reart_sqdist3(ax, ay, az, qx[k], qy[k], qz[k])) + pj[k]) - h);
                const bool better = !((scanned >> k) & 1u) && nd < d[k];
                d[k] = better ? nd : d[k];
                if (better) cpred[tid + k * BS] = i;
            }
            MWS(4);
        }
        if (buckets && !aborted && !unsolved) {
            // ---- the search goes on in BUCKETS.  All unlabelled columns with a label below hi = (closest label) + width are
            // settled TOGETHER: rounds relax from every bucket member whose label is new or has improved (up to MW_BK per round,
            // one barrier per round) until no label inside the bucket moves -- label-correcting inside a bucket, label-setting
            // from bucket to bucket (delta-stepping), so the labels below the first sink's are the shortest distances Dijkstra
            // finds, one column per 1.2 us step.  A long search labels hundreds of columns of a nearly tight graph whose
            // shortest-path tree is 40-100 nodes deep (tools/sim_tail.py on dumped solves): 900 steps become ~170 rounds.
            // Reduced costs are clamped at zero inside a bucket (they are >= 0 up to rounding; a cycle of rounding-negative
            // edges must not improve labels for ever).  A sink (unowned column, or a node of a live tree) is never relaxed
            // from; the search ends with the closest sink of the first bucket that holds one.
            unsigned sinkb = freecol;
#pragma unroll
            for (int k = 0; k < CPT; ++k) {
                const int j = tid + k * BS;
                if (j < n && tof[j] >= 0) sinkb |= 1u << k;
            }
            unsigned pend = 0u;
            double lo = INFINITY;
            bool first = true;
            const double bdelta = mx * 1e-12;
            float rk[CPT];                                    // (d - p)(1 + 2^-20) of the unsettled columns rounded UP to fp32 (settled: -inf, never a candidate)
#pragma unroll
            for (int k = 0; k < CPT; ++k) rk[k] = ((scanned >> k) & 1u) ? -INFINITY : mw_f32_up((d[k] - pj[k]) * MW_FILTER_K);
            for (;;) {
                // closing reductions of the previous bucket double as the opening of this one: the closest unlabelled column
                double bv = INFINITY;
                int bj = 0x7fffffff;
#pragma unroll
                for (int k = 0; k < CPT; ++k)
                    if (!((scanned >> k) & 1u) && d[k] < bv) { bv = d[k]; bj = tid + k * BS; }
                if (first) {
                    mw_argmin_key<6>(bv, bj);
                    if (lane == 0) { s_bv[bpar][wv] = bv; s_bj[bpar][wv] = bj; }
                    if (race && tid == 0) s_blost[bpar] = lost();
                    __syncthreads();
                    if (race && s_blost[bpar]) { aborted = true; break; }
                    bv = lane < NW ? s_bv[bpar][lane] : INFINITY; bj = lane < NW ? s_bj[bpar][lane] : 0x7fffffff;
                    mw_argmin_key<(NW <= 2 ? 1 : (NW <= 4 ? 2 : (NW <= 8 ? 3 : 4)))>(bv, bj);
                    bpar ^= 1;
                    ++my_steps;
                    lo = bv;
                    first = false;
                }
                if (!(lo < INFINITY)) { unsolved = true; break; }
                const double hi = lo + bwidth;
#pragma unroll
                for (int k = 0; k < CPT; ++k)
                    if (!((scanned >> k) & 1u) && (d[k] < hi || d[k] == lo)) pend |= 1u << k;
                pend &= ~sinkb;
                // ---- rounds
                for (;;) {
                    const int buf = brot & 1, slot = brot % 3;
#pragma unroll
                    for (int k = 0; k < CPT; ++k) {
                        const bool want = (pend >> k) & 1u;
                        const unsigned long long m = __ballot(want);
                        if (m) {
                            int base = 0;
                            if (lane == 0) base = atomicAdd(&s_ecnt[slot], __builtin_popcountll(m));
                            base = mw_uniform(base);
                            const int at = base + __builtin_popcountll(m & ((1ull << lane) - 1ull));
                            if (want && at < MW_BK) {
                                const int j = tid + k * BS, i = owner[j];
                                const double h = hcol[j];
                                float *rec = &s_e4[buf][at >> 1][at & 1];
                                rec[0] = psx[i]; rec[2] = psy[i]; rec[4] = psz[i];
                                rec[6] = mw_f32_up(((h - d[k]) + bdelta) * MW_FILTER_K);
                                s_eg[buf][at][0] = d[k]; s_eg[buf][at][1] = h; s_ei[buf][at] = i;
                                pend &= ~(1u << k);
                            }
                        }
                    }
                    if (tid == 0) s_ecnt[(brot + 1) % 3] = 0;
                    __syncthreads();
                    int ne = s_ecnt[slot];
                    ++brot;
                    if (ne == 0) break;                    // uniform: nobody had anything left to relax from
                    ne = ne < MW_BK ? ne : MW_BK;
                    ++my_steps;
#ifdef REART_PRUNE_PHASE
                    my_relax += ne; ++my_rounds;
#endif
                    // The label of column k improves through entry e iff c < (d - p) + (h - df) =: T (in exact arithmetic).  Nearly no
                    // (entry, column) pair does, and the square root and the double-precision sums are most of the work: a pair whose
                    // squared distance exceeds T^2 by more than every rounding on the way is skipped.  The test is all fp32, two
                    // columns per packed operand, two entries in flight: T from two UPPER bounds (column: d - p, entry: h - df +
                    // bdelta -- 1e-12 of the cost scale, which covers the <= 1e-15 of the double sums), both times 1 + 2^-20 (which
                    // covers the fp32 sum, the square, the estimate's own roundings and the 2^-24 of the fp32 root); the squared
                    // distance as an ESTIMATE with fused multiply-adds (the cost itself never is: common.h); T |T| instead of
                    // T > 0 && T^2.  Two ENTRIES per packed operand (the lists hold them interleaved).  The pairs that pass take
                    // the expression itself, in the entries' order.
                    for (int e = 0; e < ne; e += 2) {
                        const mw_f4 Exy = *(const mw_f4 *)&s_e4[buf][e >> 1][0];
                        mw_f4 Ezg = *(const mw_f4 *)&s_e4[buf][e >> 1][4];
                        if (e + 1 >= ne) Ezg.w = -INFINITY;            // (odd count: the record's second half holds an older round's entry)
                        const jv_f2 ex = {Exy.x, Exy.y}, ey = {Exy.z, Exy.w}, ez = {Ezg.x, Ezg.y}, eg = {Ezg.z, Ezg.w};
                        float s0[CPT], s1[CPT], Q0[CPT], Q1[CPT];
#pragma unroll
                        for (int k = 0; k < CPT; ++k) {                // the two ENTRIES are the packed pair: one form for any columns per thread
                            const jv_f2 a = mw_sq_estimate(ex - qx[k], ey - qy[k], ez - qz[k]);
                            const jv_f2 T = eg + rk[k];
                            s0[k] = a.x; s1[k] = a.y;
                            Q0[k] = T.x * fabsf(T.x); Q1[k] = T.y * fabsf(T.y);
                        }
                        bool any = false;
#pragma unroll
                        for (int k = 0; k < CPT; ++k) any |= (s0[k] <= Q0[k]) | (s1[k] <= Q1[k]);
                        if (any) {
#pragma unroll
                            for (int u = 0; u < 2; ++u) {
                                const float Ex = u ? ex.y : ex.x, Ey = u ? ey.y : ey.x, Ez = u ? ez.y : ez.x;
#pragma unroll
                                for (int k = 0; k < CPT; ++k)
                                    if (u ? s1[k] <= Q1[k] : s0[k] <= Q0[k]) {
                                        const double df = s_eg[buf][e + u][0], h = s_eg[buf][e + u][1];
                                        double w = ((double)mw_sqrt(reart_sqdist3(Ex, Ey, Ez, qx[k], qy[k], qz[k])) + pj[k]) - h;
                                        w = w > 0.0 ? w : 0.0;
                                        const double nd = df + w;
                                        if (nd < d[k]) {
                                            d[k] = nd; rk[k] = mw_f32_up((nd - pj[k]) * MW_FILTER_K);
                                            cpred[tid + k * BS] = s_ei[buf][e + u];
                                            if ((nd < hi || nd == lo) && !((sinkb >> k) & 1u)) pend |= 1u << k;
                                        }
                                    }
                            }
                        }
                    }
                }
                // ---- the bucket is stable: its columns are settled; the closest sink among them ends the search, otherwise the
                // closest column outside opens the next bucket (one pair of reductions for both)
                double sv = INFINITY;
                int sj = 0x7fffffff, nnew = 0;
                bv = INFINITY; bj = 0x7fffffff;
#pragma unroll
                for (int k = 0; k < CPT; ++k) {
                    if ((scanned >> k) & 1u) continue;
                    if (d[k] < hi || d[k] == lo) {
                        scanned |= 1u << k;
                        rk[k] = -INFINITY;
                        ++nnew;
                        if ((sinkb >> k) & 1u) {
                            const int key = (tid + k * BS) | (((freecol >> k) & 1u) ? 0 : JV_OWNED);
                            if (d[k] < sv || (d[k] == sv && key < sj)) { sv = d[k]; sj = key; }
                        }
                    } else if (d[k] < bv) { bv = d[k]; bj = tid + k * BS; }
                }
                mw_argmin_key<6>(sv, sj);
                mw_argmin_key<6>(bv, bj);
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) nnew += __shfl_xor(nnew, o, 64);
                if (lane == 0) { s_bs[bpar][wv] = sv; s_bsj[bpar][wv] = sj; s_bv[bpar][wv] = bv; s_bj[bpar][wv] = bj; s_bn[bpar][wv] = nnew; }
                if (race && tid == 0) s_blost[bpar] = lost();
                __syncthreads();
                if (race && s_blost[bpar]) { aborted = true; break; }
                sv = lane < NW ? s_bs[bpar][lane] : INFINITY; sj = lane < NW ? s_bsj[bpar][lane] : 0x7fffffff;
                bv = lane < NW ? s_bv[bpar][lane] : INFINITY; bj = lane < NW ? s_bj[bpar][lane] : 0x7fffffff;
                nnew = lane < NW ? s_bn[bpar][lane] : 0;
                mw_argmin_key<(NW <= 2 ? 1 : (NW <= 4 ? 2 : (NW <= 8 ? 3 : 4)))>(sv, sj);
                mw_argmin_key<(NW <= 2 ? 1 : (NW <= 4 ? 2 : (NW <= 8 ? 3 : 4)))>(bv, bj);
#pragma unroll
                for (int o = 8; o >= 1; o >>= 1) nnew += __shfl_xor(nnew, o, 64);
                nnew = mw_uniform(nnew);
                bpar ^= 1;
                ++my_steps;
#ifdef REART_PRUNE_PHASE
                ++my_buckets;
#endif
                if (sj != 0x7fffffff && sv < INFINITY) { mu = sv; sink = sj & ~JV_OWNED; break; }
                if (nnew < MW_BUCKET_LO) bwidth *= MW_BUCKET_UP;
                else if (nnew > MW_BUCKET_HI) bwidth = fmax(bwidth * 0.5, bdelta * 1e-3);      // (never down to zero: it could not grow again)
                lo = bv;
                if (bj == 0x7fffffff) { unsolved = true; break; }
            }
        }
        if (aborted || unsolved) break;
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            const int j = tid + k * BS;
            // (a bucket settles columns beyond the sink's label too: their prices stay)
            if ((((scanned & ~deadq) >> k) & 1u) && j != sink && d[k] < mu) { pj[k] += mu - d[k]; price[j] = pj[k]; }
        }
        __syncthreads();
        const int tree_hit = tof[sink];                    // the tree this search uses up (its root, when the search met the unowned column itself)
        if (tid == 0) {                                    // flip the path
            if (owner[sink] >= 0) {
                // down the tree first: every row on the way moves to its parent column (tight pairs: prices stay), the last one
                // takes the unowned root -- from the root upwards, so that no owner is overwritten before it has moved.  The
                // way can be hundreds of columns long (the forest grows a node per round): the parent links are reversed on the
                // way down and followed back (the tree is spent after this search anyway).
                int prev = -1, c = sink, guard = 0;
                for (; c >= 0 && guard <= n; ++guard) { const int up = tpr[c]; tpr[c] = prev; prev = c; c = up; }
                if (guard > n || owner[prev] >= 0) sh.unsolved = 1;           // (a loop or a root that is owned: never observed)
                else
                    for (c = prev; tpr[c] >= 0; c = tpr[c]) {
                        const int ch = tpr[c], r = owner[ch];
                        assigned[r] = c; owner[c] = r;
                        hcol[c] = (double)mw_sqrt(reart_sqdist3(psx[r], psy[r], psz[r], ptx[c], pty[c], ptz[c])) + price[c];
                    }
            }
            int j = sink, guard = 0;
            for (;; ++guard) {
                const int i = cpred[j];
                const int jn = assigned[i];
                assigned[i] = j; owner[j] = i;
                if (i == i0) break;
                if (guard > n || jn < 0) { sh.unsolved = 1; break; }      // a broken chain must not spin: the host solves this problem
                j = jn;
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < CPT; ++k) {                    // the tree is spent (its unowned column is taken)
            const int j = tid + k * BS;
            if (tree_hit >= 0 && j < n && tof[j] == tree_hit) tof[j] = -1;
        }
#pragma unroll
        for (int k = 0; k < CPT; ++k) {                    // the labelled columns have new prices, those on the path new owners
            const int j = tid + k * BS;
            if (((scanned & ~deadq) >> k) & 1u) {
                const int i = owner[j];          // (a bucket may have settled unowned columns besides the sink)
                if (i >= 0) hcol[j] = (double)mw_sqrt(reart_sqdist3(psx[i], psy[i], psz[i], qx[k], qy[k], qz[k])) + pj[k];
            }
        }
    }
    MWS_FLUSH();
    if (tid == 0 && aborted) sh.abort_ = 1;
    if (tid == 0 && unsolved) sh.unsolved = 1;
    if (wv != 0) my_steps = 0;                             // every wave counted the same steps
    MWP_ADD(4, MWP_NOW() - tp_);
#ifdef REART_PRUNE_PHASE
    if (tid == 0) { sh.conflicts = min(my_relax >> 4, 0xffff); sh.arr = min(my_rounds, 0xfff) | (min(my_buckets, 0xfff) << 12); }
    __syncthreads();
    my_arr = 0; my_conf = 0;
#endif
    if (lane == 0) { atomicAdd(&sh.steps, my_steps); atomicAdd(&sh.arr, my_arr); atomicAdd(&sh.conflicts, my_conf); }
    __syncthreads();
    if (wv == 0) MWP_ADD(2, MWP_NOW() - tp_);
    if (mw_flag(&sh.abort_)) return;                    // uniform after the barrier: another racer has published
    if (race) {
        if (tid == 0) sh.flag = atomicCAS(a.done + b, 0, racer + 1) == 0;
        __syncthreads();
        if (!sh.flag) return;
    }
    const bool solved = !sh.unsolved;
    for (int i = tid; i < n; i += BS) a.col4row[(size_t)b * n + i] = assigned[i];
    if (a.price_out)
        for (int j = tid; j < n; j += BS) a.price_out[(size_t)b * n + j] = price[j];
    if (tid == 0) {
        a.certified[b] = solved ? 2 : 0;                // 2 = pending: the certificate launches follow
        a.scale[b] = mx; a.cert_bad[b] = 0;
        if (a.stats) {
            int *o = a.stats + 4 * b;
            o[0] = st_freed + (racer << 16); o[1] = nleft | (sh.conflicts << 16); o[2] = sh.steps; o[3] = 1 + (sh.arr << 8);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The row reduction of ONE problem on MANY compute units (lap_mc_arr_kernel, between the set-up and the search launches of
// reart_internal_jvmc_launch).  Chains only meet in the column they commit on, so nothing ties them to one workgroup's LDS:
// the state lives in memory, every wave of gridDim.x workgroups per problem follows a chain, and a commit is lock-free --
//   1. owner[j] : seen -> LOCKED by compare-and-swap (fails if somebody else has committed on j since the wave looked);
//   2. price[j] still the value the decision used?  (otherwise owner[j] <- seen again, and the row is scanned again);
//   3. price[j] += v2 - v1, assigned[i] <- j, assigned[displaced row] <- -1;  4. owner[j] <- i (release).
// All reads and writes of the shared state are device-scope atomics (no cached copy can go stale, and the workgroups of a
// problem may sit on different XCDs); that a wave works with prices which other waves are raising meanwhile is harmless for
// the reason given at the top of this file.  A chain gives up after MW_MC_CHAIN steps or at an exact tie on an owned column and
// leaves its row for the path searches.
// A chain is sequential (5 us per step at n = 1024, 10 us at 2048: a round of memory-side atomics and two dependent ones), so
// the launch lasts as long as its longest chain is allowed to.  Measured, chains of 256 / 128 / 64 steps: README recipe
// (9 x 1024^2) 12.1 / 11.2 s / - for the whole run, kinematic projection (19 x 2048^2) 62.5 / 64.9 / 64.4 iterations/s; on 8
// workgroups per problem (13, 16 or 28: no better -- more commits collide).  JvArgs.mc_chain overrides.
#define MW_MC_CHAIN 128
#define MW_MC_CHAIN_FULL 64     // ... with the chip full of problems (see reart_internal_jvmc_launch)
// ... and that longest chain is nearly always a HOPELESS one: a row the searches end up with burns its whole budget first, so
// with one to four such rows per problem (the typical re-solve of the projection) the launch lasts 128 steps x 3 us = 0.4 ms
// while the other thousand chains are done after 0.1 (tools/replay_kernels.py: arr 418 us of a 1 310 us solve with <= 8 rows
// left).  Cutting every chain at 32 or 64 steps loses (rows that would have settled go to the searches: recipe +6 %).  So the
// cut depends on who is still running: a chain that has used MW_ARR_TAIL_STEPS steps gives up once all but 1 / MW_ARR_TAIL_DIV of
// the launch's teams have run out of rows -- while the chip is busy a long chain delays nobody, at the end it delays everybody.
// ... and on what the row's problem already has in store for the searches: where many rows are left anyway the searches are the
// long part of the solve and every further row costs a search of its own (hold-out sets, tools/holdout.sh: 11 x 896^2 with 17
// rows left lost 6 % to an unconditional cut), so the cut applies only while fewer than MW_ARR_TAIL_LEFT rows of the problem
// have been given up (4 / 8 / 16 / no limit on the hold-out sets and nao, replayed dumps: 16 is never the worst, the others each are
// somewhere; at LOOP level -- tools/ab_loop.sh, tools/holdout_loop.sh: --deterministic loops solve the same problems with every
// variant, each from the potentials its own solves left -- 32 together with MW_FOREST_FEW 8 is never behind: nao projection 442 ->
// 458 it/s, hold-out projections +6 / 0 / +3 %, kinematic leg +1 %, recipes +-0; 64 and steps 16 lose).
#define MW_ARR_TAIL_STEPS 32
#define MW_ARR_TAIL_DIV 16
#define MW_ARR_TAIL_LEFT 32

// Between the set-up and the row reduction: every unowned column's price is lowered until the first MATCHED row is indifferent
// between it and its own column (the step the searches' part of lap_jvmw_kernel explains; here for all the columns the
// released rows left, so that a chain ends in such a column as soon as it displaces that row).  A wave per unowned column:
// the matched rows' points and potentials (one per owned column) sit in its registers, a column is CPL distances per lane and
// a wave minimum.  Unowned columns do not enter any row's potential, so they are independent of each other; the launch is
// complete before the chains start, whose argument needs prices that only rise from then on.
template <int CPL>
__global__ __launch_bounds__(64 * MW_NW) void lap_mc_tighten_kernel(JvArgs a) {
    const int n = a.n, b = blockIdx.y, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int nh = a.mc_cnt[8 * b + 7];
    const int w0 = blockIdx.x * MW_NW + wv, nw = gridDim.x * MW_NW;
    if (w0 >= nh) return;
    double *price = a.mc_price + (size_t)b * n;
    const int *owner = a.mc_owner + (size_t)b * n, *holes = a.mc_next + (size_t)b * n;
    const float *S_ = a.src + (size_t)b * n * 3, *T_ = a.tgt + (size_t)b * n * 3;
    float sx[CPL], sy[CPL], sz[CPL];
    double u[CPL];
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
        const int j = 64 * k + lane;
        const int i = j < n ? owner[j] : -1;
        const int ii = i >= 0 ? i : 0, jj = j < n ? j : 0;
        sx[k] = S_[3 * ii]; sy[k] = S_[3 * ii + 1]; sz[k] = S_[3 * ii + 2];
        u[k] = i >= 0 ? (double)mw_sqrt(reart_sqdist3(sx[k], sy[k], sz[k], T_[3 * jj], T_[3 * jj + 1], T_[3 * jj + 2])) + price[jj] : INFINITY;
    }
    for (int h = w0; h < nh; h += nw) {
        const int jh = holes[h];
        const float hx = T_[3 * jh], hy = T_[3 * jh + 1], hz = T_[3 * jh + 2];
        const double ph = price[jh];
        double m = INFINITY;
#pragma unroll
        for (int k = 0; k < CPL; ++k)
            if (u[k] < INFINITY) m = fmin(m, (((double)mw_sqrt(reart_sqdist3(sx[k], sy[k], sz[k], hx, hy, hz)) + ph) - u[k]));
        m = lap_wave_min_d(m);
        if (lane == 0 && m > 0.0 && m < INFINITY) price[jh] = ph - m;
    }
}
// After the row reduction, before the path searches: for every column still unowned, the TREE of matched rows that reach it at
// zero reduced cost, grown MW_TREE_K nodes deep.  The step of lap_mc_tighten_kernel one level up and repeated: a tree T (columns
// with the rows that own them, rooted in the unowned column) may lower all its prices -- and with them its rows' potentials,
// which ARE c + p of their own pairs -- by the least slack any row outside has towards a column inside:
//     D = min over rows i not in T, columns t in T of (c_it + p_t - u_i);
// nothing becomes infeasible (rows inside only gain slack towards the outside, rows outside keep >= 0 towards the inside), all
// pairs stay tight, and the row that attains D now reaches the tree at no cost: it joins with its column.  This is the backward
// Dijkstra search from the unowned column, done for all of them at once -- a wave per tree, one node per round:
//   * per row (one per owned column: CPL slots per lane) the wave keeps M_i = min over the tree's columns of (c_it + q_t), where
//     q_t = p_t + the tree's total shift when t joined, so that a round costs ONE distance per row (to the newest column) and the
//     slack is (M_i - shift) - u_i;
//   * trees compete for rows through a compare-and-swap on a.mc_tree; a tree whose nearest row belongs to another tree stops
//     (it has shifted by that slack, which is all it may do);
//   * a wave reads potentials that other trees are lowering meanwhile: a stale potential is a HIGHER one, the slack comes out
//     smaller, the shift stays safe; prices are written once, when the tree is done (p_t = q_t - shift).
// The searches (lap_jvmw_kernel<., 2>) end at the first labelled column that belongs to a live tree and walk down its parent
// links: the path to an unowned column is known and costs nothing.  Without trees a search labels every column cheaper than the
// last hop to ONE particular row (in a graph whose reduced costs are nearly all within rounding of zero: hundreds).
#define MW_TREE_K 12        // round 4 (one-column searches) measured 6 / 12 / 24 / 40: recipe 3.71 / 3.59 / 3.56 / 3.59 ms per refresh; round 5 (bucket rounds, the
                            // forest in buckets) 3 / 6 / 12 / 24: recipe 2.78 / 2.70 / 2.60-2.69 / 2.79-2.81 ms, projection 174 / 176 / 178 / 176 it/s
template <int CPL>
__global__ __launch_bounds__(256) void lap_mc_trees_kernel(JvArgs a) {
    static_assert(MW_TREE_K < 63, "a tree's nodes live one per lane");
    const int n = a.n, b = blockIdx.y, lane = threadIdx.x & 63;
    const int *cnt = a.mc_cnt + 8 * b;
    const int nh = cnt[2];                               // rows left for the searches = columns still unowned
    const int w0 = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), nw = gridDim.x * (blockDim.x >> 6);
    if (w0 >= nh || cnt[6]) return;
    double *price = a.mc_price + (size_t)b * n;
    const int *owner = a.mc_owner + (size_t)b * n;
    int *tree = a.mc_tree + (size_t)b * n, *tpar = a.mc_tpar + (size_t)b * n;
    const float *S_ = a.src + (size_t)b * n * 3, *T_ = a.tgt + (size_t)b * n * 3;
    auto ld_d = [](const double *p) -> double { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto bcast_d = [](double v, int l) -> double {
        return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
    };
    float sx[CPL], sy[CPL], sz[CPL];
    double u[CPL], M[CPL];
    unsigned long long free_mask[CPL];                   // wave-uniform: the unowned columns of slot k
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
        const int j = 64 * k + lane;
        const int i = j < n ? owner[j] : 0;
        free_mask[k] = __ballot(j < n && i < 0);
        const int ii = (j < n && i >= 0) ? i : 0, jj = j < n ? j : 0;
        sx[k] = S_[3 * ii]; sy[k] = S_[3 * ii + 1]; sz[k] = S_[3 * ii + 2];
        u[k] = (j < n && i >= 0) ? (double)mw_sqrt(reart_sqdist3(sx[k], sy[k], sz[k], T_[3 * jj], T_[3 * jj + 1], T_[3 * jj + 2])) + ld_d(price + jj)
                                 : INFINITY;
    }
    for (int h = w0; h < nh; h += nw) {
        int jh = -1;                                     // the h-th unowned column
        {
            int c = 0;
#pragma unroll
            for (int k = 0; k < CPL; ++k) {
                const int pc = __builtin_popcountll(free_mask[k]);
                if (jh < 0 && h < c + pc) {
                    unsigned long long m2 = free_mask[k];
                    for (int t = 0; t < h - c; ++t) m2 &= m2 - 1ull;
                    jh = 64 * k + __ffsll((long long)m2) - 1;
                }
                c += pc;
            }
        }
        if (jh < 0) break;                               // (fewer unowned columns than rows left: non-finite costs upstream)
        if (lane == 0) { tree[jh] = h; tpar[jh] = -1; }
#pragma unroll
        for (int k = 0; k < CPL; ++k) M[k] = INFINITY;
        double off = 0.0, mq = 0.0;
        int mcol = -1, nm = 1;
        unsigned mine = 0u;
        // lane m keeps the tree's m-th column: its index, its q and its POINT (a round is then one round trip to memory -- the
        // claim, with the joining column's price and point fetched alongside it -- instead of four one after the other: the
        // claim, the price, the point for the next scan, the members' points for the parent; 8 -> 5 us per round)
        float mtx = 0.f, mty = 0.f, mtz = 0.f;
        if (lane == 0) { mcol = jh; mq = ld_d(price + jh); mtx = T_[3 * jh]; mty = T_[3 * jh + 1]; mtz = T_[3 * jh + 2]; }
        for (int r = 0; r < MW_TREE_K; ++r) {
            const double tq = bcast_d(mq, nm - 1);
            const float tx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mtx), nm - 1));
            const float ty = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mty), nm - 1));
            const float tz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mtz), nm - 1));
            double best = INFINITY;
            int bk = 0x7fffffff;
#pragma unroll
            for (int k = 0; k < CPL; ++k)
                if (u[k] < INFINITY && !((mine >> k) & 1u)) {
                    M[k] = fmin(M[k], (double)mw_sqrt(reart_sqdist3(sx[k], sy[k], sz[k], tx, ty, tz)) + tq);
                    const double sl = (M[k] - off) - u[k];
                    if (sl < best) { best = sl; bk = 64 * k + lane; }
                }
            lap_wave_argmin_fast(best, bk);
            if (!(best < INFINITY)) break;
            off += best > 0.0 ? best : 0.0;
            bk = mw_uniform(bk);
            // (issued before the claim's answer is waited for: an unclaimed column's price is written by nobody -- a tree writes its
            // own columns only, when it is done -- so the value is the one a read behind a successful claim would return)
            const double pnew = ld_d(price + bk);
            const float nbx = T_[3 * bk], nby = T_[3 * bk + 1], nbz = T_[3 * bk + 2];
            int ok = 0;
            if (lane == 0) {
                int seen = -1;
                ok = __hip_atomic_compare_exchange_strong(tree + bk, &seen, h, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 1 : 0;
            }
            if (!mw_uniform(ok)) break;                  // the nearest row is another tree's: this one is as low as it may go
            // the row's parent: the tree column that attains its M (the same expression, so the same value)
            float rx = 0.f, ry = 0.f, rz = 0.f;
#pragma unroll
            for (int k = 0; k < CPL; ++k)
                if ((bk >> 6) == k) { rx = sx[k]; ry = sy[k]; rz = sz[k]; }
            rx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rx), bk & 63));      // (the builtin moves integers: bits, not values)
            ry = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ry), bk & 63));
            rz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rz), bk & 63));
            double pv = INFINITY;
            int pm = 0x7fffffff;
            if (lane < nm) {
                pv = (double)mw_sqrt(reart_sqdist3(rx, ry, rz, mtx, mty, mtz)) + mq;
                pm = mcol;
            }
            lap_wave_argmin_fast(pv, pm);
            if (lane == 0) tpar[bk] = pm;
            if (lane == nm) { mcol = bk; mq = pnew + off; mtx = nbx; mty = nby; mtz = nbz; }
            if (lane == (bk & 63)) mine |= 1u << (bk >> 6);
            ++nm;
        }
        if (lane < nm) __hip_atomic_store(price + mcol, mq - off, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// The trees of lap_mc_trees_kernel stop where they meet (a tree's nearest row belongs to its neighbour) -- and the unowned
// columns of a re-solve cluster, so they meet at once.  Here the forest goes on growing AS ONE SET: all its columns (and their
// rows' potentials) shift together by the least slack of an outside row, that row joins under the column it is now tight to
// and inherits that column's ROOT (a search that reaches it walks to that unowned column; the root is what a search uses up).
// One workgroup per problem: a thread owns the rows of its columns (M_i = min over the forest of c_it + q_t and the column
// that attains it in registers), a round is one workgroup arg-min + one distance per row.  MW_FOREST_R rounds.
// With one to four rows left for the searches (the typical re-solve of the nao projection) a forest of 512 rows costs 0.44 ms for
// searches that take 0.16 (tools/replay_kernels.py): such a problem grows MW_FOREST_PER rows per row left.  NOT in proportion
// for every problem: with 10-20 rows left (the hold-out sequences of tools/holdout.sh, the recipe) a smaller forest loses 6-9 %.
// Up to how many rows left: 4 / 6 / 8 / 12 / 16 at loop level (tools/ab_loop.sh, nao projection, same trajectory): 442 / 447 / 453 /
// 457 / 451 it/s; 8 with MW_ARR_TAIL_LEFT 32: 458, and the one setting that is behind on none of the hold-out loops
// (tools/holdout_loop.sh; 12 loses 0.5 % on the recipe).  MW_FOREST_PER 48: 447; MW_FOREST_R 256 / 768: 449 / 425.
#define MW_FOREST_FEW 8
// ... and a problem with MANY rows left grows a larger forest: its searches are the long part of the solve and flood hundreds of
// columns before they meet the forest (tools/sim_regrow.py: a forest over the whole graph saves 10-25 % of the settled columns
// of the hard solves).  At loop level (tools/ab_loop.sh / holdout_loop.sh, same trajectories): more than 24 rows left -> 1 024
// rows: the hold-out projection whose solves are all hard 95.3 -> 98.3 it/s, the other +1.8 %, the kinematic leg 210 -> 215, nao
// +-0; -> 2 048 rows from 24 / 32 rows left: +5 / +4 % on the hard one, -0.7 % on the other; from 16 or 12 rows left: -5 ... -14 %.
#define MW_FOREST_MANY 24
#define MW_FOREST_R_MANY 1024
#define MW_FOREST_PER 32
#define MW_FOREST_W0 1e-8     // first bucket width of the growth, as a fraction of the cost scale
#define MW_FOREST_R 512       // measured 0 / 48 / 128 / 256 / 512 / 1024: recipe 3.57 / 3.58 / 3.51 / 3.37 / 3.33 / 3.70 ms per refresh, projection 78.8 / 77.9 / 80.4 / 82.4 / 86.7 / 86.8 it/s
template <int CPL>
__global__ __launch_bounds__(512) void lap_mc_forest_kernel(JvArgs a) {
    constexpr int BS = 512, NW = 8, CPT = CPL / NW >= 1 ? CPL / NW : 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char fsm[];
    const int n = a.n, b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int *cnt = a.mc_cnt + 8 * b;
    if (cnt[2] <= 0 || cnt[6]) { if (tid == 0) a.mc_cnt[8 * b + 7] = 0; return; }
    double *gprice = a.mc_price + (size_t)b * n;
    const int *owner = a.mc_owner + (size_t)b * n;
    int *gtree = a.mc_tree + (size_t)b * n, *gtpar = a.mc_tpar + (size_t)b * n;
    const float *S_ = a.src + (size_t)b * n * 3, *T_ = a.tgt + (size_t)b * n * 3;
    double *q = (double *)fsm;                    // [n] every column's price as the launch found it
    double *oj = q + n;                           // [n] forest columns: the shift when they joined
    int *flist = (int *)(oj + n);                 // [n] the forest's columns
    int *troot = flist + n;                       // [n] column -> root (tree id) or -1
    int *jp = troot + n;                          // [n] rows that joined: jump pointer towards the root (starts as the parent)
    float *ltx = (float *)(jp + n), *lty = ltx + n, *ltz = lty + n;       // [n] target points
    __shared__ int s_n;
    if (tid == 0) s_n = 0;
    __syncthreads();
    float sx[CPT], sy[CPT], sz[CPT];
    double u[CPT], M[CPT];
    int Mt[CPT];
    unsigned out = 0u;                            // bit k: a matched row outside the forest
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
        const int j = tid + k * BS;
        const int jj = j < n ? j : 0;
        const int i = j < n ? owner[j] : -1, tr = j < n ? gtree[j] : -1;
        const double pk = gprice[jj];
        if (j < n) {
            troot[j] = tr; q[j] = pk; oj[j] = 0.0;
            ltx[j] = T_[3 * j]; lty[j] = T_[3 * j + 1]; ltz[j] = T_[3 * j + 2];
            if (tr >= 0) flist[atomicAdd(&s_n, 1)] = j;
        }
        const int ii = i >= 0 ? i : 0;
        sx[k] = S_[3 * ii]; sy[k] = S_[3 * ii + 1]; sz[k] = S_[3 * ii + 2];
        const bool o = j < n && i >= 0 && tr < 0;
        if (o) out |= 1u << k;
        u[k] = o ? (double)mw_sqrt(reart_sqdist3(sx[k], sy[k], sz[k], T_[3 * jj], T_[3 * jj + 1], T_[3 * jj + 2])) + pk : INFINITY;
        M[k] = INFINITY; Mt[k] = -1;
    }
    __syncthreads();
    const int nf0 = s_n;
    for (int m = 0; m < nf0; ++m) {               // M over the forest the trees left
        const int t = flist[m];
        const float tx = ltx[t], ty = lty[t], tz = ltz[t];
        const double tq = q[t];
#pragma unroll
        for (int k = 0; k < CPT; ++k)
            if ((out >> k) & 1u) {
                const double v = (double)mw_sqrt(reart_sqdist3(sx[k], sy[k], sz[k], tx, ty, tz)) + tq;
                if (v < M[k]) { M[k] = v; Mt[k] = t; }
            }
    }
    double off = 0.0;
    int nf = nf0;
    const int rounds = cnt[2] <= MW_FOREST_FEW ? min(MW_FOREST_R, MW_FOREST_PER * cnt[2]) : (cnt[2] > MW_FOREST_MANY ? MW_FOREST_R_MANY : MW_FOREST_R);     // rows the growth stops after
    // The growth is a shortest-path computation like the searches' (the label of an outside row: L_i = M_i - u_i, the shift
    // at which it becomes tight to the forest; a row that joins at shift o offers its column at q + o to everybody else) and
    // runs in BUCKETS like them: all outside rows with a label below (closest label) + width join together, label-correcting
    // rounds inside the bucket (a member whose label improves offers its column again), one barrier per round.  One row per
    // round and arg-min, the first form, took 512 x 1.15 us of every re-solve.
    __shared__ __attribute__((aligned(16))) float s_f4[2][MW_BK / 2][8];      // entries in PAIRS: x0 x1 y0 y1 z0 z1 g0 g1 (the searches' layout and filter)
    __shared__ double s_fq[2][MW_BK], s_cmax[2][NW], s_cmin[2][NW];
    __shared__ int s_fj[2][MW_BK], s_fcnt[3], s_cn[2][NW];
    if (tid == 0) { s_fcnt[0] = 0; s_fcnt[1] = 0; s_fcnt[2] = 0; }
    const double fdelta = a.scale[b] * 1e-12, pref = gprice[0];
    double bw = a.scale[b] * MW_FOREST_W0, L[CPT];
    float rk[CPT];
    int brot = 0, bpar = 0, seq = 0, joined = 0;
    unsigned pend = 0u;
    double lo = INFINITY;
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
        L[k] = ((out >> k) & 1u) ? M[k] - u[k] : INFINITY;
        if (!((out >> k) & 1u)) M[k] = -INFINITY;            // never a candidate of the relaxations' filter
        // (M - pref)(1 + 2^-20) rounded UP to fp32, rewritten whenever M moves; potentials RELATIVE to one of the problem's own: they
        // drift over a run's refreshes to many times the cost scale, and an fp32 bound of the raw value would let every pair pass
        rk[k] = ((out >> k) & 1u) ? mw_f32_up((M[k] - pref) * MW_FILTER_K) : -INFINITY;
        lo = fmin(lo, L[k]);
    }
    lo = lap_wave_min_d(lo);
    if (lane == 0) s_cmin[bpar][wv] = lo;
    __syncthreads();                                         // (also: s_fcnt is cleared)
    lo = s_cmin[bpar][0];
#pragma unroll
    for (int w = 1; w < NW; ++w) lo = fmin(lo, s_cmin[bpar][w]);
    bpar ^= 1;
    ++seq;
    while (joined < rounds && lo < INFINITY) {
        const double base = fmax(lo, off), hi = base + bw;
#pragma unroll
        for (int k = 0; k < CPT; ++k)
            if (((out >> k) & 1u) && (L[k] < hi || L[k] == lo)) pend |= 1u << k;
        for (;;) {
            const int buf = brot & 1, slot = brot % 3;
#pragma unroll
            for (int k = 0; k < CPT; ++k) {
                const bool want = (pend >> k) & 1u;
                const unsigned long long m = __ballot(want);
                if (m) {
                    int at0 = 0;
                    if (lane == 0) at0 = atomicAdd(&s_fcnt[slot], __builtin_popcountll(m));
                    at0 = mw_uniform(at0);
                    const int at = at0 + __builtin_popcountll(m & ((1ull << lane) - 1ull));
                    if (want && at < MW_BK) {
                        const int j = tid + k * BS;
                        const double tq = q[j] + fmax(L[k], off);
                        float *rec = &s_f4[buf][at >> 1][at & 1];
                        rec[0] = ltx[j]; rec[2] = lty[j]; rec[4] = ltz[j];
                        rec[6] = mw_f32_up((fdelta - (tq - pref)) * MW_FILTER_K);
                        s_fq[buf][at] = tq; s_fj[buf][at] = j;
                        pend &= ~(1u << k);
                    }
                }
            }
            if (tid == 0) s_fcnt[(brot + 1) % 3] = 0;
            __syncthreads();
            int ne = s_fcnt[slot];
            ++brot;
            if (ne == 0) break;
            ne = ne < MW_BK ? ne : MW_BK;
            ++seq;
            // v = c + tq improves M iff c < M - tq =: T.  The searches' filter (lap_jvmw_kernel, bucket rounds) mirrored: all fp32, T from
            // two UPPER bounds (the row's M, the entry's delta - tq, both times 1 + 2^-20 and rounded up), the squared distance as an
            // estimate with fused multiply-adds, T |T| for T > 0 && T^2, two ENTRIES per packed operand; the pairs that pass -- a
            // superset of those that improve -- take the exact expression in the entries' order, so the result is bit for bit the
            // unfiltered one.  (The first form -- per pair a double-precision T, a conversion and two double products -- was ~16
            // instructions per pair against ~6: the kernel is bound by the instructions a compute unit issues.)
            for (int e = 0; e < ne; e += 2) {
                const mw_f4 Exy = *(const mw_f4 *)&s_f4[buf][e >> 1][0];
                mw_f4 Ezg = *(const mw_f4 *)&s_f4[buf][e >> 1][4];
                if (e + 1 >= ne) Ezg.w = -INFINITY;            // (odd count: the record's second half holds an older round's entry)
                const jv_f2 ex = {Exy.x, Exy.y}, ey = {Exy.z, Exy.w}, ez = {Ezg.x, Ezg.y}, eg = {Ezg.z, Ezg.w};
                float s0[CPT], s1[CPT], Q0[CPT], Q1[CPT];
#pragma unroll
                for (int k = 0; k < CPT; ++k) {
                    const jv_f2 d2 = mw_sq_estimate(ex - sx[k], ey - sy[k], ez - sz[k]);
                    const jv_f2 T = eg + rk[k];
                    s0[k] = d2.x; s1[k] = d2.y;
                    Q0[k] = T.x * fabsf(T.x); Q1[k] = T.y * fabsf(T.y);
                }
                bool any = false;
#pragma unroll
                for (int k = 0; k < CPT; ++k) any |= (s0[k] <= Q0[k]) | (s1[k] <= Q1[k]);
                if (any) {
#pragma unroll
                    for (int w = 0; w < 2; ++w) {
                        const float tx = w ? ex.y : ex.x, ty = w ? ey.y : ey.x, tz = w ? ez.y : ez.x;
#pragma unroll
                        for (int k = 0; k < CPT; ++k)
                            if (w ? s1[k] <= Q1[k] : s0[k] <= Q0[k]) {
                                const double tq = s_fq[buf][e + w];
                                const double v = (double)mw_sqrt(reart_sqdist3(sx[k], sy[k], sz[k], tx, ty, tz)) + tq;
                                if (v < M[k]) {
                                    const double ln = v - u[k];
                                    // a member (or a row this brings into the bucket) whose join shift got smaller offers its column again
                                    if ((ln < hi || ln == lo) && fmax(ln, off) < fmax(L[k], off)) pend |= 1u << k;
                                    M[k] = v; Mt[k] = s_fj[buf][e + w]; L[k] = ln;
                                    rk[k] = mw_f32_up((v - pref) * MW_FILTER_K);
                                }
                            }
                    }
                }
            }
        }
        // ---- the bucket is stable: its rows join (parent = the column that attains M, shift = their label, at least the
        // forest's shift so far); one pair of reductions: how many joined, the largest shift among them, the closest row left
        int nnew = 0;
        double mxo = -INFINITY, mn = INFINITY;
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            if (!((out >> k) & 1u)) continue;
            if (L[k] < hi || L[k] == lo) {
                const int j = tid + k * BS;
                const double o = fmax(L[k], off);
                out &= ~(1u << k);
                gtpar[j] = Mt[k]; jp[j] = Mt[k]; oj[j] = o;
                flist[atomicAdd(&s_n, 1)] = j;
                M[k] = -INFINITY; L[k] = INFINITY; rk[k] = -INFINITY;
                mxo = fmax(mxo, o);
                ++nnew;
            } else mn = fmin(mn, L[k]);
        }
        mxo = -lap_wave_min_d(-mxo); mn = lap_wave_min_d(mn);
#pragma unroll
        for (int o_ = 32; o_ >= 1; o_ >>= 1) nnew += __shfl_xor(nnew, o_, 64);
        if (lane == 0) { s_cmax[bpar][wv] = mxo; s_cmin[bpar][wv] = mn; s_cn[bpar][wv] = nnew; }
        __syncthreads();
        mxo = s_cmax[bpar][0]; mn = s_cmin[bpar][0]; nnew = s_cn[bpar][0];
#pragma unroll
        for (int w = 1; w < NW; ++w) { mxo = fmax(mxo, s_cmax[bpar][w]); mn = fmin(mn, s_cmin[bpar][w]); nnew += s_cn[bpar][w]; }
        bpar ^= 1;
        ++seq;
        joined += nnew;
        off = fmax(off, mxo);
        lo = mn;
        if (nnew < MW_FOREST_LO) bw *= 4.0;
        else if (nnew > MW_FOREST_HI) bw = fmax(bw * 0.5, fdelta * 1e-3);                      // (never down to zero: it could not grow again)
    }
    nf = nf0 + joined;
    // the roots of the rows that joined: their parents' (pointer jumping over the parent links; a tree's first columns carry
    // their root from lap_mc_trees_kernel)
    for (int it = 0; it < 16; ++it) {
        int moved = 0;
        for (int m = nf0 + tid; m < nf; m += BS) {
            const int c = flist[m];
            if (troot[c] >= 0) continue;
            const int p = jp[c], r = troot[p];
            if (r >= 0) troot[c] = r;
            else { jp[c] = jp[p]; moved = 1; }
        }
        if (!__syncthreads_or(moved)) break;
    }
    if (tid == 0) a.mc_cnt[8 * b + 7] = seq < 1023 ? seq : 1023;     // sequential workgroup-wide steps of the growth (rounds + bucket reductions), reported like the searches'
    __syncthreads();
    for (int m = tid; m < nf; m += BS) {          // the forest's prices and roots
        const int t = flist[m];
        gprice[t] = (q[t] + oj[t]) - off;
        gtree[t] = troot[t];
    }
}

#define MW_LOCKED (-2)
// The commit below orders its stores with s_waitcnt vmcnt(0) instead of release semantics: an argument about gfx9 hardware
// (stores count in vmcnt; device-scope atomics are served at the memory side), not something the HIP memory model promises.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "lap_mc_arr_kernel's commit protocol is written for gfx942 / gfx950"
#endif
template <int CPL>
__global__ __launch_bounds__(64 * MW_NW) void lap_mc_arr_kernel(JvArgs a) {
    const int n = a.n, b = blockIdx.y, lane = threadIdx.x & 63;
    double *price = a.mc_price + (size_t)b * n;
    int *owner = a.mc_owner + (size_t)b * n, *assigned = a.mc_assigned + (size_t)b * n, *next = a.mc_next + (size_t)b * n;
    const int *flist = a.mc_list + (size_t)b * n;
    int *cnt = a.mc_cnt + 8 * b;
    const float *S_ = a.src + (size_t)b * n * 3, *T_ = a.tgt + (size_t)b * n * 3;
    const int nfree = cnt[0];
    jv_f2 tcx[CPL / 2], tcy[CPL / 2], tcz[CPL / 2];
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
        const int j = 64 * k + lane < n ? 64 * k + lane : 0;
        tcx[k >> 1][k & 1] = T_[3 * j]; tcy[k >> 1][k & 1] = T_[3 * j + 1]; tcz[k >> 1][k & 1] = T_[3 * j + 2];
    }
    auto ld_i = [](const int *p) -> int { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto ld_d = [](const double *p) -> double { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    int my_arr = 0, my_conf = 0;
    for (;;) {
        int q = 0;
        if (lane == 0) q = __hip_atomic_fetch_add(&cnt[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        q = mw_uniform(q);
        if (q >= nfree) break;
        int i = flist[q];
        float ax = S_[3 * i], ay = S_[3 * i + 1], az = S_[3 * i + 2];
        int spins = 0;
        for (int budget = a.mc_chain; ; ) {
            // ONE round of reads per step: every column's price and owner as they are now (stale by the time they are used:
            // harmless, see the top of the file; what the step writes is validated under the column's lock)
            constexpr bool OWN_TOO = CPL <= 16;          // (32 columns per lane: the owners would not fit the registers -- 332 B of scratch)
            double pr[CPL];
            int ow[OWN_TOO ? CPL : 1];
#pragma unroll
            for (int k = 0; k < CPL; ++k) {
                const int j = 64 * k + lane;
                pr[k] = j < n ? ld_d(price + j) : INFINITY;
                if (OWN_TOO) ow[k] = j < n ? ld_i(owner + j) : -1;
            }
            float rc[CPL];
            mw_row_costs<CPL>(ax, ay, az, tcx, tcy, tcz, rc);
            double v1 = INFINITY, v2 = INFINITY;
            int j1 = 0x7fffffff, pay = 0;
#pragma unroll
            for (int k = 0; k < CPL; ++k) lap_top2_push((double)rc[k] + pr[k], 64 * k + lane, v1, j1, v2);
            lap_wave_top2_fast(v1, j1, v2, pay);
            if (!(v1 < INFINITY)) { if (lane == 0) __hip_atomic_store(&cnt[6], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            // the price and owner the decision used, from the lane that holds the arg-min column (no second round of reads)
            j1 = mw_uniform(j1);
            double pl = 0.0;
            int ol = -1;
#pragma unroll
            for (int k = 0; k < CPL; ++k)
                if ((j1 >> 6) == k) { pl = pr[k]; if (OWN_TOO) ol = ow[k]; }
            const double pj1 = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(pl), j1 & 63),
                                                __builtin_amdgcn_readlane(__double2loint(pl), j1 & 63));
            const int own = OWN_TOO ? __builtin_amdgcn_readlane(ol, j1 & 63) : ld_i(owner + j1);
            if (own == MW_LOCKED && ++spins < (1 << 14)) { ++my_conf; continue; }     // somebody is committing on it: look again
            if (own == MW_LOCKED) {                                            // (never observed; a wait must not be unbounded)
                if (lane == 0) next[__hip_atomic_fetch_add(&cnt[2], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)] = i;
                break;
            }
            const bool tie = !(v1 < v2);
            if (--budget < 0 || (tie && own >= 0)) {                           // out of budget / an exact tie on an owned column
                if (lane == 0) next[__hip_atomic_fetch_add(&cnt[2], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)] = i;
                break;
            }
            // the displaced row's point: in flight while the commit makes its round trips
            const int in_ = own >= 0 ? own : i;
            const float nx = S_[3 * in_], ny = S_[3 * in_ + 1], nz = S_[3 * in_ + 2];
            int ok = 0;
            if (lane == 0) {
                int seen = own;
                // Orders, not cache maintenance: every access to the shared state is a device-scope atomic (served where all
                // compute units meet), so the lock needs no L2 write-back / invalidate (what ACQUIRE / RELEASE at device scope
                // cost here: buffer_inv sc1 + buffer_wbl2 sc1 per commit).  The re-read of the price issues after the swap has
                // returned (it is control-dependent on its result); the owner is released after the wave's earlier stores have
                // completed (s_waitcnt vmcnt(0): on gfx9 stores count there).
                if (__hip_atomic_compare_exchange_strong(owner + j1, &seen, MW_LOCKED, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    if (ld_d(price + j1) == pj1) {
                        if (!tie) __hip_atomic_store(price + j1, pj1 + (v2 - v1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(assigned + i, j1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (own >= 0) __hip_atomic_store(assigned + own, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        __hip_atomic_store(owner + j1, i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ok = 1;
                    } else __hip_atomic_store(owner + j1, own, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            ok = mw_uniform(ok);
            if (!ok) { ++my_conf; continue; }
            ++my_arr;
            if (own < 0) break;
            i = own; ax = nx; ay = ny; az = nz;
        }
    }
    if (lane == 0) {
        __hip_atomic_fetch_add(&cnt[3], my_arr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&cnt[4], my_conf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}


// The same chains with a TEAM of TW waves per chain (one workgroup = one team).  At 32 columns per lane a wave alone spends
// 9-10 us per step on its ~2 000 instructions (32 square roots, the fp64 top-2) -- and the launch lasts as long as the longest
// chain is allowed to (128 steps: 1.2 ms of every re-solve of the projection's 2048 x 2048 problems).  In a team every wave
// scans CPL / TW columns per lane, the waves' (min, arg-min, second min) meet in LDS after ONE barrier, every wave merges the
// same TW triples (so every decision is uniform across the team without a broadcast), thread 0 commits exactly as a lone
// wave's lane 0 does and the outcome comes back after a second barrier.  Column 64 * (TW * k + wave) + lane is slot k of a lane.
template <int CPL, int TW>
__global__ __launch_bounds__(64 * TW) void lap_mc_arr_team_kernel(JvArgs a) {
    constexpr int CW = CPL / TW, LG = TW <= 2 ? 1 : (TW <= 4 ? 2 : 3);
    static_assert(CPL % TW == 0 && CW >= 2 && CW % 2 == 0 && TW <= 8, "columns per lane of a team wave: an even number");
    const int n = a.n, b = blockIdx.y, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double *price = a.mc_price + (size_t)b * n;
    int *owner = a.mc_owner + (size_t)b * n, *assigned = a.mc_assigned + (size_t)b * n, *next = a.mc_next + (size_t)b * n;
    const int *flist = a.mc_list + (size_t)b * n;
    int *cnt = a.mc_cnt + 8 * b;
    const float *S_ = a.src + (size_t)b * n * 3, *T_ = a.tgt + (size_t)b * n * 3;
    const int nfree = cnt[0];
    __shared__ double s_v1[2][TW], s_v2[2][TW], s_p[2][TW];
    __shared__ int s_j1[2][TW], s_o[2][TW], s_q[2], s_ok[2], s_fin[2];
    jv_f2 tcx[CW / 2], tcy[CW / 2], tcz[CW / 2];
    int col[CW];
#pragma unroll
    for (int k = 0; k < CW; ++k) {
        col[k] = 64 * (TW * k + wv) + lane;
        const int j = col[k] < n ? col[k] : 0;
        tcx[k >> 1][k & 1] = T_[3 * j]; tcy[k >> 1][k & 1] = T_[3 * j + 1]; tcz[k >> 1][k & 1] = T_[3 * j + 2];
    }
    auto ld_i = [](const int *p) -> int { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto ld_d = [](const double *p) -> double { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    int my_arr = 0, my_conf = 0, par = 0, qpar = 0, opar = 0;
    int *fin = a.mc_cnt + 8 * a.B;                        // launch-wide: teams that have run out of rows
    const int teams = gridDim.x * gridDim.y, tail_at = teams - teams / MW_ARR_TAIL_DIV;
    for (;;) {
        if (threadIdx.x == 0) s_q[qpar] = __hip_atomic_fetch_add(&cnt[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const int q = s_q[qpar];
        qpar ^= 1;
        if (q >= nfree) break;
        int i = flist[q];
        float ax = S_[3 * i], ay = S_[3 * i + 1], az = S_[3 * i + 2];
        int spins = 0;
        for (int budget = a.mc_chain; ; ) {
            double pr[CW];
            int ow[CW];
            // (in the same round of reads as the prices -- but only by chains long enough for the cut to apply: read by every
            // team at every step, the launch-wide counter was one address hammered from all eight XCDs, and every step of every
            // chain waited for it: +8 % on the recipe's solves)
            int fin_now = 0;
            if (threadIdx.x == 0 && a.mc_chain - budget >= MW_ARR_TAIL_STEPS) {
                fin_now = ld_i(fin);
                if (ld_i(&cnt[2]) >= MW_ARR_TAIL_LEFT) fin_now = 0;      // the problem has rows for the searches anyway: no cut
            }
#pragma unroll
            for (int k = 0; k < CW; ++k) {
                pr[k] = col[k] < n ? ld_d(price + col[k]) : INFINITY;
                ow[k] = col[k] < n ? ld_i(owner + col[k]) : -1;
            }
            float rc[CW];
            mw_row_costs<CW>(ax, ay, az, tcx, tcy, tcz, rc);
            double v1 = INFINITY, v2 = INFINITY;
            int j1 = 0x7fffffff, slot = 0;
#pragma unroll
            for (int k = 0; k < CW; ++k) lap_top2_push((double)rc[k] + pr[k], col[k], k, v1, j1, v2, slot);
            lap_wave_top2_fast(v1, j1, v2, slot);
            // the price and owner the wave's candidate was judged by, from the lane that holds that column
            j1 = mw_uniform(j1); slot = mw_uniform(slot);
            double pl = 0.0;
            int ol = -1;
#pragma unroll
            for (int k = 0; k < CW; ++k)
                if (slot == k) { pl = pr[k]; ol = ow[k]; }
            const int hl = j1 == 0x7fffffff ? 0 : (j1 & 63);
            const double pw = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(pl), hl), __builtin_amdgcn_readlane(__double2loint(pl), hl));
            const int owv = __builtin_amdgcn_readlane(ol, hl);
            if (lane == 0) { s_v1[par][wv] = v1; s_v2[par][wv] = v2; s_j1[par][wv] = j1; s_p[par][wv] = pw; s_o[par][wv] = owv; }
            if (threadIdx.x == 0) s_fin[par] = fin_now;
            __syncthreads();
            const bool tail = s_fin[par] >= tail_at && a.mc_chain - budget >= MW_ARR_TAIL_STEPS;      // uniform across the team
            v1 = lane < TW ? s_v1[par][lane] : INFINITY; v2 = lane < TW ? s_v2[par][lane] : INFINITY;
            j1 = lane < TW ? s_j1[par][lane] : 0x7fffffff;
            int win = lane;
            lap_lanes_top2<LG>(v1, j1, v2, win);
            win = win < TW ? win : 0;
            const double pj1 = s_p[par][win];
            const int own = s_o[par][win];
            par ^= 1;
            if (!(v1 < INFINITY)) { if (threadIdx.x == 0) __hip_atomic_store(&cnt[6], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            if (own == MW_LOCKED && ++spins < (1 << 14)) { ++my_conf; continue; }     // somebody is committing on it: look again
            const bool tie = !(v1 < v2);
            if (own == MW_LOCKED || --budget < 0 || (tie && own >= 0) || (tail && own >= 0)) {   // (a lock that never opens: never observed) / out of budget / an exact tie on an owned column / the launch's tail
                if (threadIdx.x == 0) next[__hip_atomic_fetch_add(&cnt[2], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)] = i;
                break;
            }
            const int in_ = own >= 0 ? own : i;
            const float nx = S_[3 * in_], ny = S_[3 * in_ + 1], nz = S_[3 * in_ + 2];
            if (threadIdx.x == 0) {
                int ok = 0, seen = own;
                // (the commit of lap_mc_arr_kernel, word for word: see there for why these orders suffice)
                if (__hip_atomic_compare_exchange_strong(owner + j1, &seen, MW_LOCKED, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    if (ld_d(price + j1) == pj1) {
                        if (!tie) __hip_atomic_store(price + j1, pj1 + (v2 - v1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(assigned + i, j1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (own >= 0) __hip_atomic_store(assigned + own, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        __hip_atomic_store(owner + j1, i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ok = 1;
                    } else __hip_atomic_store(owner + j1, own, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                s_ok[opar] = ok;
            }
            __syncthreads();
            const int ok = s_ok[opar];
            opar ^= 1;
            if (!ok) { ++my_conf; continue; }
            ++my_arr;
            if (own < 0) break;
            i = own; ax = nx; ay = ny; az = nz;
        }
    }
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(fin, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&cnt[3], my_arr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&cnt[4], my_conf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

int reart_internal_jvmw_nmax() { return 64 * 32; }

// (dynamic LDS is raised to what the launch needs, not to a flat 152 KB: the kernel also holds ~18 KB of static LDS -- the
// bucket rounds' lists -- and the two together must stay within the compute unit's 160 KB)
template <int CPL>
static int mw_launch(const JvArgs &a, int racers, hipStream_t stream) {
    const size_t lds = (size_t)a.n * (8 + 4 * 4 + 6 * 4 + 8 + 2 * 4);
    if (lds > REART_LDS_DEFAULT_CAP &&
        hipFuncSetAttribute((const void *)lap_jvmw_kernel<CPL, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return REART_ERR_LAUNCH;
    hipLaunchKernelGGL((lap_jvmw_kernel<CPL, 0>), dim3(a.B, racers), dim3(64 * MW_NW), lds, stream, a);
    REART_CHECK_LAUNCH();
    return REART_OK;
}

int reart_internal_jvmw_launch(const JvArgs &a, int racers, hipStream_t stream) {
    if (a.n < 1 || a.n > reart_internal_jvmw_nmax() || !a.src || !a.tgt || !a.pre_v1) return REART_ERR_UNSUPPORTED;
    if (a.n <= 512) return mw_launch<8>(a, racers, stream);
    if (a.n <= 1024) return mw_launch<16>(a, racers, stream);
    return mw_launch<32>(a, racers, stream);
}

template <int CPL>
static int mc_launch(const JvArgs &a, int racers, int arr_wgs, hipStream_t stream) {
    const size_t lds = (size_t)a.n * (8 + 4 * 4 + 6 * 4 + 8 + 2 * 4);
    if (lds > REART_LDS_DEFAULT_CAP &&
        (hipFuncSetAttribute((const void *)lap_jvmw_kernel<CPL, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
         hipFuncSetAttribute((const void *)lap_jvmw_kernel<CPL, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess))
        return REART_ERR_LAUNCH;
    JvArgs s1 = a;
    s1.done_clear = a.done;
    s1.done = nullptr;                                   // the set-up reads the caller's assignment and potentials themselves
    JvArgs s2 = a;
    // (a caller that deals a problem no more than two workgroups does so because the chip is full of problems -- the recipe as a
    // sweep: 4 x 95 in flight --, and there a hopeless chain is not a wait at the end of a launch but compute units taken from the
    // other groups' work: 20 x 15 000 with energies, deterministic, caps 128 / 64 / 48 / 32: 24.1 / 22.6 / 22.9 / 22.4 s)
    if (s2.mc_chain <= 0) s2.mc_chain = arr_wgs <= 2 ? MW_MC_CHAIN_FULL : MW_MC_CHAIN;
    hipLaunchKernelGGL((lap_jvmw_kernel<CPL, 1>), dim3(a.B), dim3(64 * MW_NW), lds, stream, s1);
    REART_CHECK_LAUNCH();
    hipLaunchKernelGGL((lap_mc_tighten_kernel<CPL>), dim3(arr_wgs, a.B), dim3(64 * MW_NW), 0, stream, s2);
    REART_CHECK_LAUNCH();
    // a chain step is ~1 000 instructions of ONE wave (CPL square roots, the fp64 top-2, the wave merge) and no waiting worth the
    // name: two waves on a SIMD halve each other.  Where the chip has room the same chains run as twice the workgroups of half
    // the waves -- a SIMD each.
    const int split = (2 * arr_wgs * a.B <= 256) ? 2 : 1;
    if (CPL >= 16) {
        // from 16 columns per lane on a TEAM of four waves per chain (measured 2 / 4 / 8: the projection's median re-solve 2.29 /
        // 2.12 / 2.14 ms against 2.73 with a wave per chain); arr_wgs x 8 chains in flight per problem, each a workgroup of its own
        constexpr int TW = 4;
        hipLaunchKernelGGL((lap_mc_arr_team_kernel<(CPL >= 16 ? CPL : 16), TW>), dim3(arr_wgs * MW_NW, a.B), dim3(64 * TW), 0, stream, s2);
    } else
        hipLaunchKernelGGL((lap_mc_arr_kernel<CPL>), dim3(arr_wgs * split, a.B), dim3(64 * MW_NW / split), 0, stream, s2);
    REART_CHECK_LAUNCH();
    hipLaunchKernelGGL((lap_mc_trees_kernel<CPL>), dim3(a.B * 8 <= 256 ? 8 : 4, a.B), dim3(256), 0, stream, s2);
    REART_CHECK_LAUNCH();
    {
        const size_t flds = (size_t)a.n * (8 + 8 + 4 + 4 + 4 + 12);
        if (flds > REART_LDS_DEFAULT_CAP &&
            hipFuncSetAttribute((const void *)lap_mc_forest_kernel<CPL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)flds) != hipSuccess)
            return REART_ERR_LAUNCH;
        hipLaunchKernelGGL((lap_mc_forest_kernel<CPL>), dim3(a.B), dim3(512), flds, stream, s2);
        REART_CHECK_LAUNCH();
    }
#define MW_SNW 16            // waves of a search workgroup from 16 columns per lane on (n > 512): two columns per thread at n = 2048 instead of four
                             // (projection 76.7 -> 79.0 it/s, round 5), one at n = 1024 instead of two (replayed recipe solves: slowest 24 235 ->
                             // 220 ms, sample 111 -> 108; four waves: 355) -- the rounds are bound by the instructions a SIMD issues
    constexpr int SNW = CPL >= 16 ? MW_SNW : MW_NW;
    if (SNW != MW_NW && lds > REART_LDS_DEFAULT_CAP &&
        hipFuncSetAttribute((const void *)lap_jvmw_kernel<CPL, 2, SNW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return REART_ERR_LAUNCH;
    hipLaunchKernelGGL((lap_jvmw_kernel<CPL, 2, SNW>), dim3(a.B, racers), dim3(64 * SNW), lds, stream, a);
    REART_CHECK_LAUNCH();
    return REART_OK;
}

int reart_internal_jvmc_launch(const JvArgs &a, int racers, int arr_wgs, hipStream_t stream) {
    if (a.n < 1 || a.n > reart_internal_jvmw_nmax() || !a.src || !a.tgt || !a.pre_v1 || !a.mc_price || arr_wgs < 1) return REART_ERR_UNSUPPORTED;
    if (a.n <= 512) return mc_launch<8>(a, racers, arr_wgs, stream);
    if (a.n <= 1024) return mc_launch<16>(a, racers, arr_wgs, stream);
    return mc_launch<32>(a, racers, arr_wgs, stream);
}
