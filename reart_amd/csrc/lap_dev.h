// reart_amd/csrc/lap_dev.h -- device helpers and argument blocks shared by the assignment solvers
// (lap.hip: auction, certificate, one-search-at-a-time re-solve; lap_mw.hip: the re-solve with one search per wave).
#pragma once
#include "common.h"
#include <math.h>

#define LAP_NMAX 4096

__device__ __forceinline__ unsigned long long lap_key(double v) { return (unsigned long long)__double_as_longlong(v); }  // v >= 0

// one more candidate for a lane's running (min, arg-min, second min): seven instructions (v_min / v_max on fp64 pairs) where
// the compare-and-select form takes eleven -- the row scans are bound by exactly these.  A tie keeps the earlier column.
__device__ __forceinline__ void lap_top2_push(double v, int j, double &v1, int &j1, double &v2) {
    v2 = fmin(v2, fmax(v1, v));
    j1 = v < v1 ? j : j1;
    v1 = fmin(v1, v);
}
__device__ __forceinline__ void lap_top2_push(double v, int j, int pay, double &v1, int &j1, double &v2, int &p1) {
    v2 = fmin(v2, fmax(v1, v));
    const bool lt = v < v1;
    j1 = lt ? j : j1; p1 = lt ? pay : p1;
    v1 = fmin(v1, v);
}

// the waves' merge of per-lane (min, arg-min, second min [, payload of the arg-min]) triples: LDS-free butterfly
template <int STEP>
__device__ __forceinline__ void lap_top2_step(double &v1, int &j1, double &v2, int &pay) {
    const double ov1 = reart_bfly_d<STEP>(v1), ov2 = reart_bfly_d<STEP>(v2);
    const int oj1 = reart_bfly<STEP>(j1), op = reart_bfly<STEP>(pay);
    const bool take = (ov1 < v1) || (ov1 == v1 && oj1 < j1);
    const double lose = take ? v1 : ov1;           // the larger of the two minima
    v2 = fmin(fmin(v2, ov2), lose);
    v1 = take ? ov1 : v1; j1 = take ? oj1 : j1; pay = take ? op : pay;
}
__device__ __forceinline__ void lap_wave_top2(double &v1, int &j1, double &v2, int &pay) {
    lap_top2_step<0>(v1, j1, v2, pay); lap_top2_step<1>(v1, j1, v2, pay); lap_top2_step<2>(v1, j1, v2, pay);
    lap_top2_step<3>(v1, j1, v2, pay); lap_top2_step<4>(v1, j1, v2, pay); lap_top2_step<5>(v1, j1, v2, pay);
}
__device__ __forceinline__ void lap_wave_top2(double &v1, int &j1, double &v2) {
    int pay = 0;
    lap_wave_top2(v1, j1, v2, pay);
}
// The same results with a third of the instructions when the minimum is attained by ONE lane (the normal case in fp64):
// the minimum alone by a butterfly, its column / payload read from the lane that holds it, the second minimum by another
// butterfly.  An exact tie (or a wave without candidates) takes the full butterfly: the lowest column wins either way.
__device__ __forceinline__ double lap_wave_min_d(double v) {
    v = fmin(v, reart_bfly_d<0>(v)); v = fmin(v, reart_bfly_d<1>(v)); v = fmin(v, reart_bfly_d<2>(v));
    v = fmin(v, reart_bfly_d<3>(v)); v = fmin(v, reart_bfly_d<4>(v)); v = fmin(v, reart_bfly_d<5>(v));
    return v;
}
__device__ __forceinline__ void lap_wave_top2_fast(double &v1, int &j1, double &v2, int &pay) {
    const double m = lap_wave_min_d(v1);
    const unsigned long long at = __ballot(v1 == m);
    if (__builtin_popcountll(at) == 1) {
        const int wl = __ffsll((long long)at) - 1;
        const double c = lap_wave_min_d((int)(threadIdx.x & 63) == wl ? v2 : v1);
        j1 = __builtin_amdgcn_readlane(j1, wl); pay = __builtin_amdgcn_readlane(pay, wl);
        v1 = m; v2 = c;
    } else lap_wave_top2(v1, j1, v2, pay);
}
// the same for values that live in the first 2^LG lanes only (the waves' results meeting after a barrier; the other lanes
// hold +inf): LG butterfly steps instead of six; every lane gets the result
template <int LG>
__device__ __forceinline__ double lap_lanes_min_d(double v) {
    v = fmin(v, reart_bfly_d<0>(v));
    if (LG > 1) v = fmin(v, reart_bfly_d<1>(v));
    if (LG > 2) v = fmin(v, reart_bfly_d<2>(v));
    if (LG > 3) v = fmin(v, reart_bfly_d<3>(v));
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
template <int LG>
__device__ __forceinline__ void lap_lanes_top2(double &v1, int &j1, double &v2, int &pay) {
    const int lane = threadIdx.x & 63;
    const double m = lap_lanes_min_d<LG>(v1);
    const unsigned long long at = __ballot(lane < (1 << LG) && v1 == m);
    if (__builtin_popcountll(at) == 1) {
        const int wl = __ffsll((long long)at) - 1;
        const double c = lap_lanes_min_d<LG>(lane == wl ? v2 : v1);
        j1 = __builtin_amdgcn_readlane(j1, wl); pay = __builtin_amdgcn_readlane(pay, wl);
        v1 = m; v2 = c;
    } else {
        lap_top2_step<0>(v1, j1, v2, pay);
        if (LG > 1) lap_top2_step<1>(v1, j1, v2, pay);
        if (LG > 2) lap_top2_step<2>(v1, j1, v2, pay);
        if (LG > 3) lap_top2_step<3>(v1, j1, v2, pay);
        v1 = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v1)), __builtin_amdgcn_readfirstlane(__double2loint(v1)));
        v2 = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v2)), __builtin_amdgcn_readfirstlane(__double2loint(v2)));
        j1 = __builtin_amdgcn_readfirstlane(j1); pay = __builtin_amdgcn_readfirstlane(pay);
    }
}
template <int LG>
__device__ __forceinline__ void lap_lanes_argmin(double &v, int &j) {
    const int lane = threadIdx.x & 63;
    const double m = lap_lanes_min_d<LG>(v);
    const unsigned long long at = __ballot(lane < (1 << LG) && v == m);
    if (__builtin_popcountll(at) == 1) { j = __builtin_amdgcn_readlane(j, __ffsll((long long)at) - 1); v = m; }
    else {
        reart_argmin_step<0>(v, j);
        if (LG > 1) reart_argmin_step<1>(v, j);
        if (LG > 2) reart_argmin_step<2>(v, j);
        if (LG > 3) reart_argmin_step<3>(v, j);
        v = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
        j = __builtin_amdgcn_readfirstlane(j);
    }
}
__device__ __forceinline__ void lap_wave_argmin_fast(double &v, int &j) {
    const double m = lap_wave_min_d(v);
    const unsigned long long at = __ballot(v == m);
    if (__builtin_popcountll(at) == 1) { j = __builtin_amdgcn_readlane(j, __ffsll((long long)at) - 1); v = m; }
    else reart_wave_argmin_d(v, j);
}

#define JV_RACE_MAX 28
static __device__ const int jv_race_prime[JV_RACE_MAX] = {0, 0, 4099, 4111, 4127, 4129, 4133, 4139, 4153, 4157, 4159, 4177, 4201, 4211,
                                                          4217, 4219, 4229, 4231, 4241, 4243, 4253, 4259, 4261, 4271, 4273, 4283, 4289, 4297};
__device__ __forceinline__ int jv_order(int k, int cnt, int racer) {
    if (racer == 0) return k;
    if (racer == 1) return cnt - 1 - k;
    return (int)(((unsigned)k * (unsigned)(jv_race_prime[racer] % cnt)) % (unsigned)cnt);   // the primes exceed every count: a permutation
}
typedef float jv_f2 __attribute__((ext_vector_type(2)));
#define JV_OWNED (1 << 30)  // tie key of the path search's arg-min: owned columns after unowned ones
#define JV_PTS_NMAX 2048   // points form: both point sets + the solver state must fit in LDS
#ifndef JV_PTS_BS
#define JV_PTS_BS 512
#endif
#define JV_SPLIT_NMIN 512   // from here on the two full passes of a re-solve run as their own whole-chip launches
#ifndef JV_ARR_BUDGET
#define JV_ARR_BUDGET 8    // row-reduction steps allowed per free row before the rest goes to the path search
#endif
struct JvArgs {
    const float *cost; int B, n;
    int *col4row;              // in: previous assignment (or -1), out: the optimum
    int *certified;
    const double *price_in;    // previous potentials (prices, the auction's sign convention)
    double *price_out;
    int max_rounds_cert;
    int *stats;                // nullable [B][4]: released rows, rows left for the path search, Dijkstra steps, certificate rounds
                               // + 256 * row-reduction steps
    double keep_tol;           // fraction of the largest cost
    // PTS form: no cost matrix; c_ij = sqrt(((dx*dx)+(dy*dy))+(dz*dz)) of src point i and tgt point j, the expression of
    // reart_cdist, evaluated where it is needed from copies of both point sets in LDS
    const float *src, *tgt;    // [B][n][3]
    // three-launch form (jv_launch): the two full passes over the costs run on the whole chip, the sequential part in between
    double *pre_v1, *pre_cur;  // [B][n] per row: min_k (c_ik + p_k) and c_i,s(i) + p_s(i) under the incoming prices / assignment
    int *pre_j1;               // [B][n] the arg-min column
    double *scale;             // [B] the cost scale the tolerances are fractions of
    int *cert_bad;             // [B] set by the certificate pass when a row's column is not its arg-min
    int pass_mode;             // lap_jv_pass_kernel: 0 = row potentials of the start, 1 = first certificate round
    // racing form (MODE 1, gridDim.y racers per matrix): the racers read the start from copies (col4row / price_out are
    // written by the winner while others may still be loading) and meet in done[b] (0 = nobody has finished)
    int *done;
    int *done_clear;           // set-up launch of the lap_mw.hip form (its own `done` is null: it does not race): cleared there, no memset launch
    // lap_mw.hip, form with the row reduction on many compute units: the state between its three launches, in the workspace
    double *mc_price;          // [B][n]
    int *mc_owner, *mc_assigned, *mc_list, *mc_next;       // [B][n] each: column -> row | row -> column | free rows | rows left
    int *mc_tree, *mc_tpar;    // [B][n] each: column -> tree (the unowned column it leads to at zero reduced cost; -1 none) | its parent column there
    int *mc_cnt;               // [B][8]: free rows | queue head | rows left | reduction steps | conflicts | released | unsolved | unowned columns
    int mc_chain;              // steps after which a chain leaves its row to the path searches
    const int *col_start;
    const double *price_start;
    // --deterministic (ties.hip): the certificate pass also lists the pairs off the assignment that are tight under the final
    // potentials -- its scan meets exactly those -- for the cycle check that follows the solve; null: not asked for
    int *tie_edges, *tie_n;    // [B][n][tie_cap] every row's tight columns (its first tie_cap) | [B][n] how many it has
    int tie_cap;
};

// ties.hip: the cycle check over pair lists a solve's certificate pass wrote (reart_lap_resolve_points_mc_ties); `stale` [B]
// nullable: problems whose potentials moved after the pass (certificate rounds) get tie = 2
int reart_internal_tie_cycles(int B, int n, const int *col4row, int *tie, const int *cols, const int *cnt, int K, const int *stale,
                              hipStream_t stream);

// lap_mw.hip: the sequential part of a points-form re-solve with one search per WAVE (see there); same inputs and
// outputs as lap_jv_kernel<., true, 1>.  Returns REART_ERR_UNSUPPORTED when n exceeds what its waves hold in registers.
int reart_internal_jvmw_launch(const JvArgs &a, int racers, hipStream_t stream);
// the same with the row reduction spread over `arr_wgs` workgroups per problem (state in a.mc_*): three launches
int reart_internal_jvmc_launch(const JvArgs &a, int racers, int arr_wgs, hipStream_t stream);
int reart_internal_jvmw_nmax();
